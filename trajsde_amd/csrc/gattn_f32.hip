// gattn_f32.hip -- the global interactor's attention (reference models/aggregators/agg_hivt.py:101-117) with its per-edge multiply-adds
// on the fp32 matrix cores (round 4).  Same mathematics as attn.hip k_global_attn<8, false, DROP, true> -- lin_k_edge folded into the
// query once per target, lin_v_edge applied once per target after the aggregation, one pass, online softmax -- but the two things the
// vector form spends 32 of its ~55 instructions per edge on,
//     p_h(e)  = rel_e . U_h          [16 edges x 64] x [64 x 8 heads]
//     S_h    += w_h(e) rel_e         [8 heads x 16 edges] x [16 edges x 64]
// are tile products on v_mfma_f32_16x16x4_f32: exact fp32 products, no operand split, 32 matrix instructions per 16 edges.  What
// belongs to the gathered node rows (q_h . k_node[src]_h and sum_e w_h(e) v_node[src]) stays on the vector pipe, 32 multiply-adds per
// lane and tile, on the rows as they are loaded: whole rows per 16 lanes (lane (kk, n), register r: the source of edge 4 kk + r, columns
// 4n .. 4n+3 -- half a head), like the rel rows.  (Loaded 64 bytes per lane, rows on lanes, every instruction touched 32 cache lines for
// 1 KB and the four of a row re-requested the same lines: 70 % of the kernel was waiting for memory, in-kernel phase clocks.)
// Layouts and the hand-offs between the two products: aggregator_bwd.hip k_edge_rows_bwd_mm, which this kernel shares its structure with
// (persistent workgroup of GMF_WAVES waves per CU, LDS weight image in the lanes' order, one wave per target).
#include "attn_common.hpp"
#include "common.hpp"
#include "dropout.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "stamps.hpp"
#include "tile.hpp"

TSDE_STAMP_TABLE(gmf, 8)

namespace tsde {

#ifndef TSDE_GMF_WAVES
#define TSDE_GMF_WAVES 8                                           // two waves a SIMD: the loads of a tile run a tile ahead (below)
#endif
constexpr int GMF_WAVES = TSDE_GMF_WAVES;
constexpr int GMF_TP = 68;                                          // padded row of a wave's staging tile / of the plain weight copy
constexpr int GMF_WIMG = 4096, GMF_WV = 64 * GMF_TP;               // floats: Wke in the lanes' order, plain Wve rows
constexpr int GMF_PER_WAVE = 16 * GMF_TP + 16 * 8 + 16 * 8 + 64 + 64;   // staging tile (reused for S at the end) | node logits | weights | node sums | query
constexpr int gmf_lds_bytes() { return (GMF_WIMG + GMF_WV + GMF_WAVES * GMF_PER_WAVE) * 4; }

#ifndef TSDE_GMF_ORDER
#define TSDE_GMF_ORDER 0         // 1: key / value rows in two register sets, requested in the order they are needed (measured: slower)
#endif
template <bool DROP>
__global__ __launch_bounds__(64 * GMF_WAVES) void k_global_attn_mf(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                   const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                   const float* __restrict__ q, const float* __restrict__ kn,
                                                                   const float* __restrict__ vn, int64_t N, float* __restrict__ agg,
                                                                   float* __restrict__ stats, DropArg drop) {
  constexpr int HEADS = 8;
  constexpr float INV = INV_SQRT_DH;
  constexpr int TP = GMF_TP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wimg = lds;                                         // [d][kk][v4][j < 8][4]: Wke[8 j + d][16 kk + 4 v4 + e]
  float* const wvp = lds + GMF_WIMG;                               // Wve[d][c], rows padded to TP
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const tile = lds + GMF_WIMG + GMF_WV + wv * GMF_PER_WAVE;
  float* const pnt = tile + 16 * TP;                               // [16 edges][8 heads]: node part of the logits
  float* const wt = pnt + 128;                                     // [16 edges][8 heads]: the tile's softmax weights
  float* const nsum = wt + 128;                                    // [64]: node-row part of the aggregate
  float* const qsm = nsum + 64;                                    // [64]: the target's scaled query row
  const int c16 = lane & 15, q4 = lane >> 4, hd = c16 & 7;
  const bool lo8 = c16 < 8;
  {
    const float* wke = img + GAttnL::WKE;
    const float* wve = img + GAttnL::WVE;
    for (int i = threadIdx.x; i < GMF_WIMG / 4; i += blockDim.x) {
      const int j = i & 7, v4 = (i >> 3) & 3, kk = (i >> 5) & 3, d = i >> 7;
      *reinterpret_cast<f4*>(wimg + 4 * i) = *reinterpret_cast<const f4*>(wke + (8 * j + d) * 64 + 16 * kk + 4 * v4);
    }
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
      const int row = i >> 4, c4 = i & 15;
      *reinterpret_cast<f4*>(wvp + row * TP + 4 * c4) = *reinterpret_cast<const f4*>(wve + row * 64 + 4 * c4);
    }
  }
  __syncthreads();
  const float bve = img[GAttnL::BVE + lane];
  const int64_t stride = int64_t(gridDim.x) * GMF_WAVES;
  PhaseClock<8> clk;                                               // diagnostic builds only (stamps.hpp, tools/phase_stamps.py gmf)
  clk.start();
  unsigned long long tiles_done = 0;
  prioritize_younger_half();
  // xcd_block(): the workgroups of one XCD walk consecutive targets, so a round of them gathers one scene's node rows through ONE L2
  // (dealt by blockIdx, a scene's 256 targets were spread over all eight: every L2 had to hold the node rows of every scene in flight)
  for (int64_t node = xcd_block() * GMF_WAVES + wv; node < N; node += stride) {
    // the logits' 1 / sqrt(dh) is folded into the query; the key bias q_h . bke_h shifts every logit of a (target, head) alike: dropped
    const float ql = q[node * 64 + lane] * INV;
    const int beg = segptr[node], end = segptr[node + 1];
    if (end <= beg) {                                              // (uniform) no edge: the aggregate is zero
      agg[node * 64 + lane] = 0.f;
      if (stats != nullptr && lo8 && q4 == 0) *reinterpret_cast<float2*>(stats + (node * HEADS + hd) * 2) = float2{-INFINITY, 1.0f / 1e-16f};
      continue;
    }
    // Loads run ahead of their use, into the registers a tile has just finished with: the rel rows of tile t+2 are
    // requested once tile t is staged (two register sets, the loop body twice: the HBM rows need more than one tile's time to arrive), the gathered key rows of t+1 once t's node logits are done, its value rows once t's node sums are
    // (sources two tiles ahead).  All of them unconditional -- indices clamped to the segment's last edge -- so that the compiler's counted
    // waits stay counted.
    auto src_at = [&](int (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = src[e < end ? e : end - 1];
      }
    };
    auto rel_load = [&](f4 (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = *reinterpret_cast<const f4*>(rel + int64_t(e < end ? e : end - 1) * 64 + 4 * c16);
      }
    };
    auto row_load = [&](f4 (&dst)[4], const float* __restrict__ base, const int (&sidx)[4]) {    // four whole rows per instruction
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const f4*>(base + int64_t(sidx[r]) * 64 + 4 * c16);
    };
    int s_next[4];
    f4 nxa[4], nxb[4];                                             // rel rows of the even / odd tiles: two tiles in flight per wave
#if TSDE_GMF_ORDER
    // EXPERIMENT (round 6, -DTSDE_GMF_ORDER=1; measured, SLOWER: 168.8 / 169.8 us a layer against 165.1 / 165.6 on one box, 236 registers
    // against 213).  The memory counter is in order: a wait for a request also waits for everything requested before it.  With one set of
    // key / value rows, re-requested where the set falls free (behind the rel rows of two tiles ahead), the waits for them pull the rel
    // look-ahead in.  Here: two sets each, the next tile's rows requested at the TOP of a tile and BEFORE the rel rows of two tiles ahead,
    // so that every wait of a tile leaves the younger requests in flight.  It does not pay: the look-ahead is not what bounds the kernel.
    f4 kra[4], vra[4], krb[4], vrb[4];
#else
    f4 kra[4], vra[4];
    f4 (&krb)[4] = kra, (&vrb)[4] = vra;
#endif
    {
      int s0[4];
      src_at(s0, beg);
      src_at(s_next, beg + 16);
      rel_load(nxa, beg);
      row_load(kra, kn, s0);
      row_load(vra, vn, s0);
      rel_load(nxb, beg + 16);
    }
    // B operand of the first product: lane (kk = q4, j = c16 < 8) holds U_j at rows 16 kk + s (columns 8-15 stay zero)
    float uz[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) uz[s] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float qs = __shfl(ql, 8 * hd + d);
      const float qd = lo8 ? qs : 0.f;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const f4 wr = *reinterpret_cast<const f4*>(wimg + (((d * 4 + q4) * 4 + v4) * 8 + hd) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) uz[4 * v4 + e] = fmaf(wr[e], qd, uz[4 * v4 + e]);
      }
    }
    __builtin_amdgcn_wave_barrier();                               // (the previous target's readers of the wave's LDS are done)
    qsm[lane] = ql;                                                // read back per tile in the rows-on-lanes layout (node part of the logits)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f4 qn = *reinterpret_cast<const f4*>(qsm + 4 * c16);      // the query's columns 4n .. 4n+3 (half of head n >> 1)
    f4 R[4], accn = f4{0.f, 0.f, 0.f, 0.f};                         // accn: columns 4n .. 4n+3 of the node sums, this lane row's edges
#pragma unroll
    for (int b = 0; b < 4; ++b) R[b] = f4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, s = 0.f, sk = 0.f;                        // of head hd; s, sk: this lane's edges 4 q4 + i only
    clk.mark(7);                                                   // per-target prologue (and the previous target's epilogue)
    auto tile_step = [&](f4 (&nx)[4], f4 (&kr)[4], f4 (&vr)[4], f4 (&krn)[4], f4 (&vrn)[4], int e0) {
      ++tiles_done;
      int s_after[4];
      src_at(s_after, e0 + 32);
      __builtin_amdgcn_wave_barrier();                             // the previous readers of the wave's LDS are done (same wave, in order)
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(tile + (4 * q4 + r) * TP + 4 * c16) = nx[r];
#if TSDE_GMF_ORDER
      row_load(krn, kn, s_next);                                   // the next tile's key and value rows, into the other set ...
      row_load(vrn, vn, s_next);
#endif
      rel_load(nx, e0 + 32);                                       // ... then two tiles ahead, into the registers just staged
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      clk.mark(0);                                                 // wait for the tile's rel rows, stage them
      f4 P0 = f4{0.f, 0.f, 0.f, 0.f}, P1 = P0, P2 = P0, P3 = P0;     // four chains: a matrix instruction waits for its accumulator
      {
        f4 a[4];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) a[v4] = *reinterpret_cast<const f4*>(tile + c16 * TP + 16 * q4 + 4 * v4);
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][0], uz[4 * v4 + 0], P0, 0, 0, 0);
          P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][1], uz[4 * v4 + 1], P1, 0, 0, 0);
          P2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][2], uz[4 * v4 + 2], P2, 0, 0, 0);
          P3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][3], uz[4 * v4 + 3], P3, 0, 0, 0);
        }
      }
      P0 = (P0 + P1) + (P2 + P3);
      clk.mark(1);                                                 // first product
      // node part of the logits: edge 4 q4 + r, head n >> 1 -- this lane's four columns and its neighbour's (lane ^ 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = qn[0] * kr[r][0];
#pragma unroll
        for (int e = 1; e < 4; ++e) p = fmaf(qn[e], kr[r][e], p);
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        if ((c16 & 1) == 0) pnt[(4 * q4 + r) * 8 + (c16 >> 1)] = p;
      }
#if !TSDE_GMF_ORDER
      row_load(kr, kn, s_next);
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      clk.mark(2);                                                 // wait for the key rows, node logits
      uint32_t mine = 0u;
      if (DROP) {                                                  // lane (x, r) draws the block of edge x and keeps word r (heads 2r, 2r+1)
        uint32_t w[4];
        philox_words(drop.seed, drop_stream(drop, DK_ATTN), uint32_t(e0 - beg) + uint32_t(c16), uint32_t(node), 0u, w);
        mine = q4 == 0 ? w[0] : (q4 == 1 ? w[1] : (q4 == 2 ? w[2] : w[3]));
      }
      // lane (j = c16, q4): logits of head j for the edges 4 q4 + i (columns 8-15 mirror 0-7: their results are never used)
      f4 lg;
      float cm = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + 4 * q4 + i;
        const float p = P0[i] + pnt[(4 * q4 + i) * 8 + hd];
        lg[i] = e < end ? p : -INFINITY;
        cm = fmaxf(cm, lg[i]);
      }
      cm = row_max(cm);                                            // the head's maximum over the tile: the other three lane rows
      const float mn = fmaxf(m, cm);
      const float sc = fast_exp(m - mn);                           // m = -inf on the first tile -> 0
      m = mn;
      s *= sc;
      sk *= sc;
      f4 W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float ex = fast_exp(lg[i] - m);                            // beyond the segment: exp(-inf) = 0
        s += ex;
        if (DROP) {                                                // attention dropout (AGG:116): the softmax sum keeps every edge
          const uint32_t word = uint32_t(__shfl(int(mine), 16 * (hd >> 1) + 4 * q4 + i));
          ex *= drop_pick(word, hd & 1, drop);
        }
        sk += ex;
        W[i] = lo8 ? ex : 0.f;
        if (lo8) wt[(4 * q4 + i) * 8 + hd] = ex;
      }
      clk.mark(3);                                                 // softmax scalars
      // The sums so far shrink by the factor of their row's head: rows 4 q4 + i' of S (lane row q4), head n >> 1 of the node sums.  Only in
      // a tile that raised some head's maximum (the first ones of a target, as a rule): the factors are exactly 1 otherwise, and the five
      // cross-lane reads + 20 multiplies sit on the wave's dependent chain between the two products.
      if (__builtin_amdgcn_ballot_w64(sc != 1.0f) != 0ull) {        // (uniform)
        float sr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = __shfl(sc, (4 * q4 + i) & 7);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) R[b][i] *= sr[i];
        accn *= __shfl(sc, c16 >> 1);
      }
      clk.mark(4);                                                 // rescale
      {
        f4 rw[4];                                                  // the tile's rows again, whole rows per 16 lanes: B operand under the column order 4n + b
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r] = *reinterpret_cast<const f4*>(tile + (4 * q4 + r) * TP + 4 * c16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int b = 0; b < 4; ++b) R[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i], rw[i][b], R[b], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      clk.mark(5);                                                 // second product
#pragma unroll
      for (int r = 0; r < 4; ++r) accn += vr[r] * wt[(4 * q4 + r) * 8 + (c16 >> 1)];      // node part of the aggregate
#if !TSDE_GMF_ORDER
      row_load(vr, vn, s_next);
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) s_next[r] = s_after[r];
      clk.mark(6);                                                 // wait for the value rows, node sums
    };
    for (int e0 = beg; e0 < end; e0 += 32) {
      tile_step(nxa, kra, vra, krb, vrb, e0);
      if (e0 + 16 < end) tile_step(nxb, krb, vrb, kra, vra, e0 + 16);
    }
    // ---- per target: S_h / (s_h + 1e-16) to LDS (the staging tile is free), node sums over the 16 edge lanes, lin_v_edge
    s = xor32_sum(xor16_sum(s));
    sk = xor32_sum(xor16_sum(sk));
    const float inv = 1.0f / (s + 1e-16f);                         // PyG softmax denominator
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * q4 + i;
      const float ri = __shfl(inv, row & 7);
      if (row < 8) *reinterpret_cast<f4*>(tile + row * TP + 4 * c16) = f4{R[0][i] * ri, R[1][i] * ri, R[2][i] * ri, R[3][i] * ri};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) accn[e] = xor32_sum(xor16_sum(accn[e]));      // the four lane rows' edges
    if (q4 == 0) *reinterpret_cast<f4*>(nsum + 4 * c16) = accn;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane = channel d from here on (head h = d >> 3)
    const int h = lane >> 3;
    const float invh = __shfl(inv, h), skh = __shfl(sk, h);
    float out = fmaf(bve, skh * invh, nsum[lane] * invh);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const f4 wr = *reinterpret_cast<const f4*>(wvp + lane * TP + 4 * k4);
      const f4 sv = *reinterpret_cast<const f4*>(tile + h * TP + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
    }
    agg[node * 64 + lane] = end > beg ? out : 0.f;
    if (stats != nullptr && lo8 && q4 == 0) *reinterpret_cast<float2*>(stats + (node * HEADS + hd) * 2) = float2{m, inv};
  }
#ifdef TSDE_STAMPS
  if (lane == 0 && (wv == 0 || wv == 5)) clk.flush(g_stamps_gmf, tiles_done);
#endif
}

#ifndef TSDE_PRODUCT        // alternative form: trajsde_amd/variants/libtrajsde_alt.so only (TRAJSDE_GMF_TILES=2)
// ---- the same kernel over 32 edges a step (round 5).  The phase clocks of the kernel above say 70 % of a 16-edge tile is waiting at the
// three places its dependent chain touches memory, and the ablations say the waits are the chain, not the bytes: stage -> product ->
// scalars -> product -> node sums, ~5 300 cycles with nothing of its own to overlap at two waves per SIMD.  A third wave per SIMD does
// not fit (213 registers; through an LDS ring instead of registers the tiles in flight are 18 KB a wave).  So the chain is made to carry
// twice the edges: two consecutive tiles of the SAME target go through every phase together -- they share the target's state (m, s, R,
// node sums, U), only the tile-local values double -- and each wait is paid once per 32 edges.  Per edge and head the arithmetic is
// that of the one-tile form except for the order in which a head's maximum is raised (one rescale per 32 edges instead of two) and
// the grouping of the sums: parity against the oracle at the same tolerance, not bit-identical to the one-tile form.
// MEASURED (32 x 256 agents, one box, alternating): 0.541-0.547 ms for the three layers against 0.525-0.530 for the one-tile form --
// 3 % SLOWER at 256 registers.  Twice the edges per dependent chain buy nothing, so the chain is not what bounds the kernel either:
// what remains is pipe time -- 32 fp32 matrix instructions of 32 cycles (1 024 cycles of the SIMD's matrix pipe) and ~290 vector
// instructions (~1 300 cycles) per 16 edges and wave, served to two waves with little co-execution (HISTORY.md section 5 "Round 5").
constexpr int GMF2_PER_WAVE = 2 * 16 * GMF_TP + 32 * 8 + 32 * 8 + 64 + 64;
constexpr int gmf2_lds_bytes() { return (GMF_WIMG + GMF_WV + GMF_WAVES * GMF2_PER_WAVE) * 4; }
template <bool DROP>
__global__ __launch_bounds__(64 * GMF_WAVES) void k_global_attn_mf2(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                    const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                    const float* __restrict__ q, const float* __restrict__ kn,
                                                                    const float* __restrict__ vn, int64_t N, float* __restrict__ agg,
                                                                    float* __restrict__ stats, DropArg drop) {
  constexpr int HEADS = 8;
  constexpr float INV = INV_SQRT_DH;
  constexpr int TP = GMF_TP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wimg = lds;
  float* const wvp = lds + GMF_WIMG;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const tile = lds + GMF_WIMG + GMF_WV + wv * GMF2_PER_WAVE;   // [2][16][TP]
  float* const pnt = tile + 32 * TP;                               // [32 edges][8 heads]: node part of the logits
  float* const wt = pnt + 256;                                     // [32 edges][8 heads]: the step's softmax weights
  float* const nsum = wt + 256;
  float* const qsm = nsum + 64;
  const int c16 = lane & 15, q4 = lane >> 4, hd = c16 & 7;
  const bool lo8 = c16 < 8;
  {
    const float* wke = img + GAttnL::WKE;
    const float* wve = img + GAttnL::WVE;
    for (int i = threadIdx.x; i < GMF_WIMG / 4; i += blockDim.x) {
      const int j = i & 7, v4 = (i >> 3) & 3, kk = (i >> 5) & 3, d = i >> 7;
      *reinterpret_cast<f4*>(wimg + 4 * i) = *reinterpret_cast<const f4*>(wke + (8 * j + d) * 64 + 16 * kk + 4 * v4);
    }
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
      const int row = i >> 4, c4 = i & 15;
      *reinterpret_cast<f4*>(wvp + row * TP + 4 * c4) = *reinterpret_cast<const f4*>(wve + row * 64 + 4 * c4);
    }
  }
  __syncthreads();
  const float bve = img[GAttnL::BVE + lane];
  const int64_t stride = int64_t(gridDim.x) * GMF_WAVES;
  for (int64_t node = xcd_block() * GMF_WAVES + wv; node < N; node += stride) {
    const float ql = q[node * 64 + lane] * INV;
    const int beg = segptr[node], end = segptr[node + 1];
    if (end <= beg) {                                              // (uniform) no edge: the aggregate is zero
      agg[node * 64 + lane] = 0.f;
      if (stats != nullptr && lo8 && q4 == 0) *reinterpret_cast<float2*>(stats + (node * HEADS + hd) * 2) = float2{-INFINITY, 1.0f / 1e-16f};
      continue;
    }
    auto src_at = [&](int (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = src[e < end ? e : end - 1];
      }
    };
    auto rel_load = [&](f4 (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = *reinterpret_cast<const f4*>(rel + int64_t(e < end ? e : end - 1) * 64 + 4 * c16);
      }
    };
    auto row_load = [&](f4 (&dst)[4], const float* __restrict__ base, const int (&sidx)[4]) {    // four whole rows per instruction
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const f4*>(base + int64_t(sidx[r]) * 64 + 4 * c16);
    };
    // in flight per wave: the rel rows of the NEXT step's two tiles (nx), the key and value rows of THIS step's two tiles (kr, vr),
    // the sources of the next step's tiles (sn)
    int sn[2][4];
    f4 nx[2][4], kr[2][4], vr[2][4];
    {
      int s0[2][4];
      src_at(s0[0], beg);
      src_at(s0[1], beg + 16);
      src_at(sn[0], beg + 32);
      src_at(sn[1], beg + 48);
      rel_load(nx[0], beg);
      rel_load(nx[1], beg + 16);
      row_load(kr[0], kn, s0[0]);
      row_load(kr[1], kn, s0[1]);
      row_load(vr[0], vn, s0[0]);
      row_load(vr[1], vn, s0[1]);
    }
    float uz[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) uz[s] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float qs = __shfl(ql, 8 * hd + d);
      const float qd = lo8 ? qs : 0.f;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const f4 wr = *reinterpret_cast<const f4*>(wimg + (((d * 4 + q4) * 4 + v4) * 8 + hd) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) uz[4 * v4 + e] = fmaf(wr[e], qd, uz[4 * v4 + e]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    qsm[lane] = ql;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f4 qn = *reinterpret_cast<const f4*>(qsm + 4 * c16);
    f4 R[4], accn = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) R[b] = f4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, s = 0.f, sk = 0.f;
    for (int e0 = beg; e0 < end; e0 += 32) {
      __builtin_amdgcn_wave_barrier();                             // the previous readers of the wave's LDS are done (same wave, in order)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(tile + (16 * u + 4 * q4 + r) * TP + 4 * c16) = nx[u][r];
      rel_load(nx[0], e0 + 32);                                    // the next step's tiles, into the registers just staged
      rel_load(nx[1], e0 + 48);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // first product, both tiles: eight independent accumulator chains
      f4 P[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f4 P0 = f4{0.f, 0.f, 0.f, 0.f}, P1 = P0, P2 = P0, P3 = P0;
        f4 a[4];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) a[v4] = *reinterpret_cast<const f4*>(tile + (16 * u + c16) * TP + 16 * q4 + 4 * v4);
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][0], uz[4 * v4 + 0], P0, 0, 0, 0);
          P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][1], uz[4 * v4 + 1], P1, 0, 0, 0);
          P2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][2], uz[4 * v4 + 2], P2, 0, 0, 0);
          P3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][3], uz[4 * v4 + 3], P3, 0, 0, 0);
        }
        P[u] = (P0 + P1) + (P2 + P3);
      }
      // node part of the logits, both tiles
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = qn[0] * kr[u][r][0];
#pragma unroll
          for (int e = 1; e < 4; ++e) p = fmaf(qn[e], kr[u][r][e], p);
          p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
          if ((c16 & 1) == 0) pnt[(16 * u + 4 * q4 + r) * 8 + (c16 >> 1)] = p;
        }
      row_load(kr[0], kn, sn[0]);
      row_load(kr[1], kn, sn[1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint32_t mine[2] = {0u, 0u};
      if (DROP) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          uint32_t w[4];
          philox_words(drop.seed, drop_stream(drop, DK_ATTN), uint32_t(e0 + 16 * u - beg) + uint32_t(c16), uint32_t(node), 0u, w);
          mine[u] = q4 == 0 ? w[0] : (q4 == 1 ? w[1] : (q4 == 2 ? w[2] : w[3]));
        }
      }
      // lane (j = c16, q4): logits of head j for the edges 16 u + 4 q4 + i of the step
      f4 lg[2];
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = e0 + 16 * u + 4 * q4 + i;
          const float p = P[u][i] + pnt[(16 * u + 4 * q4 + i) * 8 + hd];
          lg[u][i] = e < end ? p : -INFINITY;
          cm = fmaxf(cm, lg[u][i]);
        }
      cm = row_max(cm);
      const float mn = fmaxf(m, cm);
      const float sc = fast_exp(m - mn);
      m = mn;
      s *= sc;
      sk *= sc;
      f4 W[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float ex = fast_exp(lg[u][i] - m);
          s += ex;
          if (DROP) {
            const uint32_t word = uint32_t(__shfl(int(mine[u]), 16 * (hd >> 1) + 4 * q4 + i));
            ex *= drop_pick(word, hd & 1, drop);
          }
          sk += ex;
          W[u][i] = lo8 ? ex : 0.f;
          if (lo8) wt[(16 * u + 4 * q4 + i) * 8 + hd] = ex;
        }
      if (__builtin_amdgcn_ballot_w64(sc != 1.0f) != 0ull) {        // (uniform)
        float sr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = __shfl(sc, (4 * q4 + i) & 7);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) R[b][i] *= sr[i];
        accn *= __shfl(sc, c16 >> 1);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f4 rw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r] = *reinterpret_cast<const f4*>(tile + (16 * u + 4 * q4 + r) * TP + 4 * c16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int b = 0; b < 4; ++b) R[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[u][i], rw[i][b], R[b], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) accn += vr[u][r] * wt[(16 * u + 4 * q4 + r) * 8 + (c16 >> 1)];
      row_load(vr[0], vn, sn[0]);
      row_load(vr[1], vn, sn[1]);
      src_at(sn[0], e0 + 64);
      src_at(sn[1], e0 + 80);
    }
    s = xor32_sum(xor16_sum(s));
    sk = xor32_sum(xor16_sum(sk));
    const float inv = 1.0f / (s + 1e-16f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * q4 + i;
      const float ri = __shfl(inv, row & 7);
      if (row < 8) *reinterpret_cast<f4*>(tile + row * TP + 4 * c16) = f4{R[0][i] * ri, R[1][i] * ri, R[2][i] * ri, R[3][i] * ri};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) accn[e] = xor32_sum(xor16_sum(accn[e]));
    if (q4 == 0) *reinterpret_cast<f4*>(nsum + 4 * c16) = accn;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int h = lane >> 3;
    const float invh = __shfl(inv, h), skh = __shfl(sk, h);
    float out = fmaf(bve, skh * invh, nsum[lane] * invh);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const f4 wr = *reinterpret_cast<const f4*>(wvp + lane * TP + 4 * k4);
      const f4 sv = *reinterpret_cast<const f4*>(tile + h * TP + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
    }
    agg[node * 64 + lane] = out;
    if (stats != nullptr && lo8 && q4 == 0) *reinterpret_cast<float2*>(stats + (node * HEADS + hd) * 2) = float2{m, inv};
  }
}

#endif  // TSDE_PRODUCT

// TRAJSDE_GATTN_F32MM=0: the vector form (attn.hip k_global_attn) for A/B runs and cross-checks
bool gattn_f32mm_enabled() {
  static const bool on = []() { const char* e = getenv("TRAJSDE_GATTN_F32MM"); return !(e && e[0] == '0'); }();
  return on;
}
int launch_global_attn_mf(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, float* stats, const DropArg& drop, hipStream_t st) {
  if (N <= 0) return TRAJSDE_OK;
  const int64_t wgs = (N + GMF_WAVES - 1) / GMF_WAVES;
  const int grid = xcd_grid(wgs < 256 ? wgs : 256);                // one workgroup per CU; fewer when the targets do not fill them (a multiple of 8: xcd_block)
  // TRAJSDE_GMF_TILES=2: 32 edges a step (k_global_attn_mf2)
  static const bool two = []() { const char* e = getenv("TRAJSDE_GMF_TILES"); return e && atoi(e) == 2; }();
#ifdef TSDE_PRODUCT
  TS_REQUIRE(!two, "TRAJSDE_GMF_TILES selects an alternative kernel form that lives in trajsde_amd/variants/libtrajsde_alt.so: point TRAJSDE_LIB at it");
#else
  if (two) {
    if (drop.p > 0.f) TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn_mf2<true>), grid, 64 * GMF_WAVES, gmf2_lds_bytes(), st, img, segptr, src, rel, q, kn, vn, N, agg, stats, drop);
    else TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn_mf2<false>), grid, 64 * GMF_WAVES, gmf2_lds_bytes(), st, img, segptr, src, rel, q, kn, vn, N, agg, stats, drop);
    return TRAJSDE_OK;
  }
#endif
  if (drop.p > 0.f) TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn_mf<true>), grid, 64 * GMF_WAVES, gmf_lds_bytes(), st, img, segptr, src, rel, q, kn, vn, N, agg, stats, drop);
  else TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn_mf<false>), grid, 64 * GMF_WAVES, gmf_lds_bytes(), st, img, segptr, src, rel, q, kn, vn, N, agg, stats, drop);
  return TRAJSDE_OK;
}

}  // namespace tsde
