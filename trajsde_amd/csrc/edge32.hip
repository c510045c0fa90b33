// edge32.hip -- the fused edge attention (attn.hip k_edge_attn2: same streams, same records, same arithmetic) on 32x32x16
// matrix tiles.
//
// Why.  A v_mfma_f32_16x16x32_f16 occupies the matrix pipe for 16 cycles and holds the SIMD's vector issue port for 8 of them;
// a v_mfma_f32_32x32x16_f16 does twice the work in 32 cycles and holds the port for the same 8 (MI355X_MICROARCH.md).  The
// kernel is bound by that port (HISTORY.md section 5), and a third of its port time was the 240 matrix instructions per 32
// rows: here they are 120.
//
// Layout.  A wave owns 32 rows (edge streams).  Lane l holds row n = l & 31 and, of every 32-feature block jo of that row, the
// features 32 jo + 8 j + 4 hh + i (hh = l >> 5, j < 4, i < 4): an activation is  f16v a[2]  with a[jo][4 j + i] -- which is exactly
// the D fragment of the 32x32 instruction when the weights are its A operand (32 output features x 16 k) and the rows its B
// operand (16 k x 32 rows), so chains of layers never leave registers, as in the 16-row layout (tile.hpp).  The k slot 8 hh + jj
// of k-step s = 2 jo_in + jp is fed from registers a[jo_in][8 jp + jj], i.e. input feature 32 jo_in + 8 (2 jp + (jj >> 2)) + 4 hh +
// (jj & 3); pack.hip PK_MAT32 writes the weight fragments in that order.  A row lives on 2 lanes instead of 4: LayerNorm and
// per-head reductions are one v_permlane32_swap; a lane holds half of each of the 8 heads (4 heads: a quarter of each).
// (Not in the product library: this file compiles to nothing under -DTSDE_PRODUCT.  The measured-slower alternative kernel forms --
//  this one, k_edge_attn2p, the one-tile instantiations of k_edge_attn2, gattn.hip -- live in trajsde_amd/variants/libtrajsde_alt.so,
//  which the tests that cross-check them load through TRAJSDE_LIB: trajsde_amd/build.py.)
#ifndef TSDE_PRODUCT
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "attn_common.hpp"
#include "dropout.hpp"
#include "tile.hpp"

namespace tsde {

#if TSDE_SPLIT_H3
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f4 grp(const f16v& v, int j) { return f4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}; }
__device__ __forceinline__ void set_grp(f16v& v, int j, const f4 x) {
  v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
}
// per-feature vector stored plainly -> the lane's features of NB blocks
template <int NB>
__device__ __forceinline__ void load_vec32(f16v (&out)[NB], const float* v, int hh) {
#pragma unroll
  for (int jo = 0; jo < NB; ++jo)
#pragma unroll
    for (int j = 0; j < 4; ++j) set_grp(out[jo], j, *reinterpret_cast<const f4*>(v + 32 * jo + 8 * j + 4 * hh));
}
// The B operands of a layer: the lane's 32 input features as four k-steps of 8 slots, split into fp16 pieces (vector work)
struct Split32 {
  u4 h[4], l[4];
};
__device__ __forceinline__ void split32(Split32& x, const f16v (&in)[2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f16v& v = in[s >> 1];
    const int o = 8 * (s & 1);
    split_kstep(f4{v[o], v[o + 1], v[o + 2], v[o + 3]}, f4{v[o + 4], v[o + 5], v[o + 6], v[o + 7]}, x.h[s], x.l[s]);
  }
}
// acc[jo] += W[32 jo .., :] * in for a weight image in PK_MAT32 fragment order [jo][k-step][piece][lane][8] (matrix work only:
// LDS reads and matrix instructions).  The weight fragments of the next PF steps are in flight while a step feeds the matrix cores.
template <int NB>
__device__ __forceinline__ void mm32(f16v (&acc)[NB], const Split32& x, const float* w, int lane) {
  constexpr int STEPS = 4 * NB;
#ifndef TSDE_EDGE32_PF
#define TSDE_EDGE32_PF 2
#endif
  constexpr int PF = TSDE_EDGE32_PF, RING = PF + 1;
  u4 f1[RING], f2[RING];
#pragma unroll
  for (int i = 0; i < PF && i < STEPS; ++i) {
    const float* p = w + ((i % NB) * 4 + i / NB) * 512 + lane * 4;
    f1[i % RING] = *reinterpret_cast<const u4*>(p);
    f2[i % RING] = *reinterpret_cast<const u4*>(p + 256);
  }
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int s = i / NB, jo = i % NB;
    if (i + PF < STEPS) {
      const int k = i + PF;
      const float* p = w + ((k % NB) * 4 + k / NB) * 512 + lane * 4;
      f1[k % RING] = *reinterpret_cast<const u4*>(p);
      f2[k % RING] = *reinterpret_cast<const u4*>(p + 256);
    }
    __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink the reads to just before their use)
    const h8 a1 = __builtin_bit_cast(h8, f1[i % RING]), a2 = __builtin_bit_cast(h8, f2[i % RING]);
    const h8 x1 = __builtin_bit_cast(h8, x.h[s]), x2 = __builtin_bit_cast(h8, x.l[s]);
    acc[jo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, x1, acc[jo], 0, 0, 0);
    acc[jo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, x2, acc[jo], 0, 0, 0);
    acc[jo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, x1, acc[jo], 0, 0, 0);
  }
}
// Ping-pong of the two waves of a SIMD.  Both run the same stream of vector phases (V) and matrix phases (M); left alone they
// drift into step -- both in a V phase halving the issue port, then both in an M phase halving the matrix pipe: measured, the
// kernel's time was the SUM of its vector-only and matrix-only times.  A workgroup barrier after every phase, and one extra
// barrier that waves 4..7 (the second wave of each SIMD) pass before their first phase and waves 0..3 after their last, keeps the
// second wave exactly one phase behind: while one wave of a SIMD issues vector instructions the other one feeds the matrix
// cores.  (Waves that left at the top have ended: the barrier counts the live ones.)
__device__ __forceinline__ void phase_sync() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// ReLU(LayerNorm(Linear(2,64)(x))) in closed form (tile.hpp in2_ln_relu), this lane's 32 features
__device__ __forceinline__ void in2_ln_relu32(f16v (&out)[2], float x0, float x1, const float* c, const float* beta, int hh) {
  const float rstd = in2_rstd(x0, x1, c);
  const float x0r = x0 * rstd, x1r = x1 * rstd;
#pragma unroll
  for (int jo = 0; jo < 2; ++jo)
#pragma unroll
    for (int j = 0; j < 4; ++j) set_grp(out[jo], j, in2_ln_relu4(x0r, x1r, rstd, c, beta, 32 * jo + 8 * j + 4 * hh));
}
// rstd of a feature-centred row (tile.hpp centred_rstd); the row's other half is on lane ^ 32
__device__ __forceinline__ float centred_rstd32(const f16v (&d)[2]) {
  float v = 0.f;
#pragma unroll
  for (int jo = 0; jo < 2; ++jo)
#pragma unroll
    for (int e = 0; e < 16; ++e) v = fmaf(d[jo][e], d[jo][e], v);
  return rsqrt_nr(xor32_sum(v) * (1.0f / 64) + 1e-5f);
}
// MultipleInputEmbedding on the EdgeL6F algebra (attn_common.hpp edge_embed_fused_n): (y - mean) * rstd of the last LayerNorm,
// already split for lin_k | lin_v (`out`); nrm: the same rows in fp32 (SAVE).  Called inside a vector phase, returns inside one;
// PP: phase_sync() between the phases.
template <bool PP>
__device__ __forceinline__ void edge_embed_fused32(Split32& out, f16v (&nrm)[2], const f4 ge, const float* lds, int lane, int hh) {
  using E = EdgeL6F;
  f16v a[2], s[2];
  Split32 x;
  in2_ln_relu32(a, ge[0], ge[1], lds + E::A_C, lds + E::A_E, hh);
  load_vec32<2>(s, lds + E::B3, hh);
  split32(x, a);
  if (PP) phase_sync();
  mm32<2>(s, x, lds + E::WA3, lane);
  if (PP) phase_sync();
  in2_ln_relu32(a, ge[2], ge[3], lds + E::B_C, lds + E::B_E, hh);
  split32(x, a);
  if (PP) phase_sync();
  mm32<2>(s, x, lds + E::WB3, lane);
  if (PP) phase_sync();
  const float r = centred_rstd32(s);
#pragma unroll
  for (int jo = 0; jo < 2; ++jo)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 ga = *reinterpret_cast<const f4*>(lds + E::AG0 + 32 * jo + 8 * j + 4 * hh);
      const f4 be = *reinterpret_cast<const f4*>(lds + E::AE0 + 32 * jo + 8 * j + 4 * hh);
#pragma unroll
      for (int i = 0; i < 4; ++i) s[jo][4 * j + i] = fmaxf(fmaf(s[jo][4 * j + i] * r, ga[i], be[i]), 0.f);
    }
  load_vec32<2>(nrm, lds + E::B2, hh);
  split32(x, s);
  if (PP) phase_sync();
  mm32<2>(nrm, x, lds + E::W2, lane);
  if (PP) phase_sync();
  const float t = centred_rstd32(nrm);
  nrm[0] *= t;
  nrm[1] *= t;
  split32(out, nrm);
}

// softmax state of one row, this lane's half: 8 head slots (slot 4 jo + j covers features 32 jo + 8 j ..: head 4 jo + j of 8, or
// head 2 jo + (j >> 1) of 4 -- then neighbouring slots carry the same numbers)
struct SegState32 {
  f16v acc[2];
  float m[8], s[8];
};
__device__ __forceinline__ void seg_reset32(SegState32& S) {
#pragma unroll
  for (int jo = 0; jo < 2; ++jo)
#pragma unroll
    for (int e = 0; e < 16; ++e) S.acc[jo][e] = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    S.m[t] = -INFINITY;
    S.s[t] = 0.f;
  }
}
// record: acc by feature | m by slot at [64 + slot] | s by slot at [80 + slot]  (k_seg_merge layout 1)
__device__ __forceinline__ void seg_flush32(const SegState32& S, float* __restrict__ rec, int64_t slot, int hh) {
  float* r = rec + slot * SEG_REC;
#pragma unroll
  for (int jo = 0; jo < 2; ++jo)
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f4*>(r + 32 * jo + 8 * j + 4 * hh) = grp(S.acc[jo], j);
  if (hh == 0) {                                            // both lanes of the row hold the same statistics
    *reinterpret_cast<f4*>(r + 64) = f4{S.m[0], S.m[1], S.m[2], S.m[3]};
    *reinterpret_cast<f4*>(r + 68) = f4{S.m[4], S.m[5], S.m[6], S.m[7]};
    *reinterpret_cast<f4*>(r + 80) = f4{S.s[0], S.s[1], S.s[2], S.s[3]};
    *reinterpret_cast<f4*>(r + 84) = f4{S.s[4], S.s[5], S.s[6], S.s[7]};
  }
}

template <bool DROP, bool SAVE, bool PP>
__global__ __launch_bounds__(512) void k_edge_attn3(const float* __restrict__ img_g, const float* __restrict__ geom,
                                                    const int32_t* __restrict__ dst, const float* __restrict__ q, EdgeCount ec, int C_host,
                                                    float* __restrict__ rec, int heads, const int32_t* __restrict__ segptr, DropArg drop,
                                                    float* __restrict__ emb_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL6F;
  const int64_t E = edge_count(ec);
  const int C = stream_len(ec, E, C_host);
  if (E <= 0) return;                                      // (only reachable when the count lives on the device)
  stage_blob(lds, img_g, EL::LDS_SIZE);
  const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const StreamMap smap = stream_map(C);
  const int64_t nstreams = stream_count(E, C);
  const int64_t wid = xcd_block() * waves + wave;          // xcd_grid launch: consecutive streams share an L2
  if (wid * 32 >= nstreams) return;                        // whole wave beyond the list (uniform)
  const int64_t sid = wid * 32 + n, base_e = stream_base(smap, sid);      // this lane's row: its stream and the stream's first edge
  SegState32 S;
  seg_reset32(S);
  int cur = -1, rank0 = 0;                                 // current target; DROP: first edge of that target (mask counter = rank)
  float* qs = lds + EL::LDS_SIZE + wave * 2048;            // this lane's 32 query values of the row's target: [group][lane][4]
  // Loads run ahead of their use: the targets two iterations, the geometry and -- when the row's target is about to change -- the
  // new target's query values (and its segment start) one iteration.  The change itself then costs register moves, LDS
  // writes and the record's stores, no round trip: with the phase barriers a wave that waited on memory held up all eight.
  f4 ng, qn[8];
  int nd, nd2, nrank = 0;
  {
    const int64_t c = base_e < E ? base_e : E - 1, c2 = base_e + 1 < E ? base_e + 1 : E - 1;
    ng = *reinterpret_cast<const f4*>(geom + 4 * c);
    nd = dst[c];
    nd2 = dst[c2];
#pragma unroll
    for (int t = 0; t < 8; ++t) qn[t] = *reinterpret_cast<const f4*>(q + int64_t(nd) * 64 + 32 * (t >> 2) + 8 * (t & 3) + 4 * hh);
    if (DROP) nrank = segptr[nd];
  }
  if (PP && (wave & 4)) phase_sync();                      // the second wave of every SIMD runs one phase behind (phase_sync)
  // every wave walks the LONGER stream length: the phase barriers of the ping-pong form count iterations, and a wave whose streams
  // are the shorter ones (kernels.hpp StreamMap) idles through the rest with its rows masked
  const int Cw = stream_length(smap, wid * 32);
  for (int it = 0; it < smap.Co; ++it) {
    keep_lds_reads_here();
    const int64_t e = base_e + it;
    const bool ok = it < Cw && e < E;
    const f4 ge = ng;
    const int d = nd;
    if (ok && d != cur) {                                  // the row's target changes: its finished segment part leaves
      if (cur >= 0) seg_flush32(S, rec, int64_t(cur) + sid, hh);
      seg_reset32(S);
      cur = d;
      if (DROP) rank0 = nrank;
#pragma unroll
      for (int t = 0; t < 8; ++t) *reinterpret_cast<f4*>(qs + (t * 64 + lane) * 4) = qn[t];
    }
    {
      const int64_t c = e + 1 < E ? e + 1 : E - 1, c2 = e + 2 < E ? e + 2 : E - 1;   // (past a stream's end: loaded, never used)
      ng = *reinterpret_cast<const f4*>(geom + 4 * c);
      nd = nd2;
      nd2 = dst[c2];
      if (e + 1 < E && nd != cur) {                        // next iteration changes target: its query row starts its trip now
#pragma unroll
        for (int t = 0; t < 8; ++t) qn[t] = *reinterpret_cast<const f4*>(q + int64_t(nd) * 64 + 32 * (t >> 2) + 8 * (t & 3) + 4 * hh);
        if (DROP) nrank = segptr[nd];
      }
    }
    f16v emb[2], kv[4];
    Split32 xe;
    edge_embed_fused32<PP>(xe, emb, ge, lds, lane, hh);
    if (SAVE && ok) {                                      // the tape holds the embedding rows proper
#pragma unroll
      for (int jo = 0; jo < 2; ++jo)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f0 = 32 * jo + 8 * j + 4 * hh;
          const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG3 + f0), be = *reinterpret_cast<const f4*>(lds + EL::AE3 + f0);
          *reinterpret_cast<f4*>(emb_out + e * 64 + f0) = grp(emb[jo], j) * ga + be;
        }
    }
    // k (blocks 0, 1) | v (blocks 2, 3) without their constant parts (EdgeL6F); with attention dropout v carries CV here
#pragma unroll
    for (int jo = 0; jo < 4; ++jo)
#pragma unroll
      for (int t = 0; t < 16; ++t) kv[jo][t] = 0.f;
    if (DROP) {
#pragma unroll
      for (int jo = 0; jo < 2; ++jo)
#pragma unroll
        for (int j = 0; j < 4; ++j) set_grp(kv[2 + jo], j, *reinterpret_cast<const f4*>(img_g + EL::CV + 32 * jo + 8 * j + 4 * hh));
    }
    if (PP) phase_sync();
    mm32<4>(kv, xe, lds + EL::WKV, lane);
    if (PP) phase_sync();
    float lg[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f4 qv = *reinterpret_cast<const f4*>(qs + (t * 64 + lane) * 4);
      const f4 kk = grp(kv[t >> 2], t & 3);
      lg[t] = fmaf(qv[3], kk[3], fmaf(qv[2], kk[2], fmaf(qv[1], kk[1], qv[0] * kk[0])));
    }
    if (heads == 4) {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        const float p = xor32_sum(lg[t] + lg[t + 1]) * 0.25f;
        lg[t] = lg[t + 1] = p;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) lg[t] = xor32_sum(lg[t]) * INV_SQRT_DH;
    }
    if (ok) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (DROP) philox_words(drop.seed, drop_stream(drop, DK_ATTN), uint32_t(int(e) - rank0), uint32_t(d), 0u, w);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float mn = fmaxf(S.m[t], lg[t]);
        const float sc = fast_exp(S.m[t] - mn);              // first edge of a segment: exp(-inf) = 0
        const float ex = fast_exp(lg[t] - mn);
        S.m[t] = mn;
        S.s[t] = fmaf(S.s[t], sc, ex);
        float exk = ex;
        if (DROP) {                                          // attention dropout (ENC:592): the kept edges, scaled, in the weighted sum
          const int h = heads == 4 ? (t >> 1) : t;
          exk *= drop_pick(w[h >> 1], h & 1, drop);
        }
        const f4 vv = grp(kv[2 + (t >> 2)], t & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) S.acc[t >> 2][4 * (t & 3) + i] = fmaf(S.acc[t >> 2][4 * (t & 3) + i], sc, exk * vv[i]);
      }
    }
  }
  if (PP && !(wave & 4)) phase_sync();
  if (cur >= 0) seg_flush32(S, rec, int64_t(cur) + sid, hh);
}
template __global__ void k_edge_attn3<false, false, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<false, false, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<true, false, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<true, false, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<false, true, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<false, true, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<true, true, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn3<true, true, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
#endif   // TSDE_SPLIT_H3

}  // namespace tsde

#endif  // TSDE_PRODUCT
