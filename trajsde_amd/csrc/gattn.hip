// gattn.hip -- the fused global attention of a GlobalInteractorLayer (AGG:101-117) on the matrix cores (round 4; inference, 8 heads,
// fp32 rows, no dropout; an ALTERNATIVE, selected by TRAJSDE_GATTN_MM=1 -- the default stays attn.hip k_global_attn, whose algebra
// this kernel shares; parity-tested in test_alternative_kernel_paths_agree):
//   logit_h(e) = [ q_h . k_node[src]_h  +  (Wke_h^T q_h) . rel_e ] / sqrt(8)          out = sum_e alpha_e v_node[src] + Wve (sum_e alpha_{e,h} rel_e) + ..
// One wave per target as before, but the two per-edge contractions are matrix products over tiles of 16 in-edges:
//   P1   logits [16 edges x 8 heads]  = [rel_e | k_node[src_e]] (K = 128)  .  W1      W1 = [U ; Q]: U_h = Wke_h^T q_h, Q = q masked to its head
//   P2   O [8 heads x 128]           += alpha^T [heads x 32 edges]          .  [rel_e | v_node[src_e]]
// both as fp16x3 split products (tile.hpp).  What made the earlier matrix-core forms lose was the load
// shape: with an edge's row on a lane every 16-byte load touched 16 rows.  Here every row is loaded ONCE, 16 bytes per lane, 16
// lanes per row -- four whole rows per load instruction -- in the layout P2 wants as its B operand: lane (n, g) holds features
// 4n .. 4n+3 of the edges 4g .. 4g+3 of two tiles, i.e. column n of the four 16-column blocks c (feature 4n + c) and the k-slots
// (tile, edge) of its lane group.  P1 wants the same rows with an EDGE on a lane: the rel and k_node rows go through a wave-private
// LDS stage on the way, already split into their fp16 planes (one split serves the 16 lanes that read a value), and come back as
// A operands with one ds_read_b128 per plane and k-step (k-slot j of lane group g of step s = feature 32 s + 8 g + j: W1 is built
// per target, so its k order is ours to choose).  P1's result fragment -- lane (head, g): the logits of edges 4g .. 4g+3 -- is,
// exactly the k-slots of P2's A operand: the weights never change lanes.
// Online softmax per head in the lanes of that head's column; the running maximum is moved only when a pair's maximum exceeds it
// by more than 8 (exp <= e^8: inside fp16's range for the split of the weights, exact in the fp32 sums), so the 32 accumulator
// registers are rescaled a few times per target instead of once per pair.
#include "attn_common.hpp"
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "stamps.hpp"
#include "tile.hpp"

TSDE_STAMP_TABLE(gattn, 8)      // diagnostic builds (tools/phase_stamps.py gattn): phases of one 16-edge tile of k_global_attn_mm

namespace tsde {
#ifndef TSDE_STAMPS
static unsigned long long* const g_stamps_gattn = nullptr;
#endif

#if TSDE_SPLIT_H3 && !defined(TSDE_PRODUCT)
constexpr int GA_PITCH = 272;                // bytes per staged row of one plane: 128 halves + 16 B (ds_read_b128 of 16 rows: distinct banks)
constexpr int GA_PLANE = 16 * GA_PITCH;      // one fp16 plane of one 16-edge tile
constexpr float GA_LAZY = 8.0f;              // the running maximum follows a pair's maximum only past this margin

// (timing experiments only: -DTSDE_GA_EXP=1 every gathered node row is row (index & 3), =2 every rel row is one of the segment's first 4)
#if defined(TSDE_GA_EXP) && TSDE_GA_EXP == 1
#define GA_EXP_NODE(i) ((i) & 3)
#else
#define GA_EXP_NODE(i) (i)
#endif
#if defined(TSDE_GA_EXP) && TSDE_GA_EXP == 2
#define GA_EXP_REL(i) ((i) & 3)
#else
#define GA_EXP_REL(i) (i)
#endif
struct GaRows {                              // the rows a lane brings for one tile, one kind: edges 4g .. 4g+3, features 4 nn .. 4 nn + 3
  f4 x[4];
};

#ifndef TSDE_GA_OCC
#define TSDE_GA_OCC 2
#endif
__global__ __launch_bounds__(256, TSDE_GA_OCC) void k_global_attn_mm(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                           const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                           const float* __restrict__ q, const float* __restrict__ kn,
                                                           const float* __restrict__ vn, int64_t N, float* __restrict__ agg) {
  __shared__ __attribute__((aligned(16))) char stage[4][2][GA_PLANE];       // [wave][plane][16 rows]
  __shared__ __attribute__((aligned(16))) float sbuf[4][8][64 + 4];
  __shared__ __attribute__((aligned(16))) float qbuf[4][64];
  __shared__ __attribute__((aligned(16))) float obuf[4][64];
  __shared__ float hbuf[4][16];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nn = lane & 15, g = lane >> 4;
  const int64_t node = xcd_block() * 4 + wv;                 // launched with xcd_grid(): a scene's targets share an L2
  const int64_t nc = node < N ? node : N - 1;
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  // ---- W1 as B operand: lane (head nn, g), step s, slot j = W1[32 s + 8 g + j][nn]; columns 8 .. 15 are zero
  const float ql = q[nc * 64 + lane] * INV_SQRT_DH;          // the logits' 1 / sqrt(dh) rides in the query (as in k_global_attn)
  qbuf[wv][lane] = ql;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u4 b1h[4], b1l[4];
  {
    const int hh = nn & 7;
    f4 qa = *reinterpret_cast<const f4*>(&qbuf[wv][8 * hh]), qb = *reinterpret_cast<const f4*>(&qbuf[wv][8 * hh + 4]);
    if (nn >= 8) qa = qb = f4{0.f, 0.f, 0.f, 0.f};
    f4 w[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) w[s][0] = w[s][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 8; ++d) {                            // U_h[c] = sum over the head's 8 dims of Wke[d][c] q[d]
      const float qd = d < 4 ? qa[d] : qb[d - 4];
      const float* row = wke + (8 * hh + d) * 64 + 8 * g;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w[s][0] += *reinterpret_cast<const f4*>(row + 32 * s) * qd;
        w[s][1] += *reinterpret_cast<const f4*>(row + 32 * s + 4) * qd;
      }
    }
    split_kstep(w[0][0], w[0][1], b1h[0], b1l[0]);
    split_kstep(w[1][0], w[1][1], b1h[1], b1l[1]);
    // k_node part: slot (s, g, j) is node feature d = 32 (s - 2) + 8 g + j, which belongs to head 4 (s - 2) + g
    const f4 z = f4{0.f, 0.f, 0.f, 0.f};
    split_kstep(nn == g ? qa : z, nn == g ? qb : z, b1h[2], b1l[2]);
    split_kstep(nn == 4 + g ? qa : z, nn == 4 + g ? qb : z, b1h[3], b1l[3]);
  }
  const int beg = segptr[nc], end = node < N ? segptr[nc + 1] : beg;
  char* const st = &stage[wv][0][0];
  f4 O[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) O[c] = f4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, spart = 0.f;
  PhaseClock<8> clk;                                         // (diagnostic builds only: stamps.hpp)
  clk.start();
  unsigned long long units = 0;
  (void)units;

  // (buffer loads: descriptor base in SGPRs + a 32-bit byte offset per lane -- no 64-bit address arithmetic per load; the rel
  //  descriptor starts at the target's segment, node rows are addressed from row 0: N < 2^23 is checked on the host)
  const __amdgpu_buffer_rsrc_t rs_rel = row_rsrc(rel + int64_t(beg) * 64), rs_kn = row_rsrc(kn), rs_vn = row_rsrc(vn);
  const __amdgpu_buffer_rsrc_t rs_src = row_rsrc(reinterpret_cast<const float*>(src + beg));
  auto row4 = [&](__amdgpu_buffer_rsrc_t rs, int row) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 256 + 16 * nn, 0, 0));
  };
  // load j of a lane: edge 4g + j of the tile.  The source indices travel TWO tiles ahead of their use and the rows one: a row load
  // whose address waits for an index load issued just before it exposes a memory latency in the middle of a tile (vmcnt is in order)
  // (row offsets relative to the segment, clamped to its last row: one v_min per load; the lane's base 4g is folded into `lim`)
  const int deg = end - beg, lim = deg - 1 - 4 * g;
  auto fetch_idx = [&](int (&sidx)[4], int e0) {
    const int o = e0 - beg;                                  // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) sidx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_src, (min(o + j, lim) + 4 * g) * 4, 0, 0);
  };
  auto fetch_rel = [&](GaRows& R, int e0) {
    const int o = e0 - beg;
#pragma unroll
    for (int j = 0; j < 4; ++j) R.x[j] = row4(rs_rel, GA_EXP_REL(min(o + j, lim) + 4 * g));
  };
  auto fetch_kv = [&](GaRows& K, GaRows& V, const int (&sidx)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) K.x[j] = row4(rs_kn, GA_EXP_NODE(sidx[j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) V.x[j] = row4(rs_vn, GA_EXP_NODE(sidx[j]));
  };
  auto tile_step = [&](const GaRows& RR, const GaRows& KK, const GaRows& VV, int e0) {
    // ---- rel and k_node rows -> the stage, as fp16 planes, an edge per row
    __builtin_amdgcn_wave_barrier();                          // the previous tile's fragment reads are done (same wave, in order)
#ifdef TSDE_STAMPS
    clk.mark(0);                                              // [0] issuing the loads of the tiles ahead
    {
      f4 t0 = RR.x[0], t1 = RR.x[3], t2 = KK.x[0], t3 = KK.x[3];
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    }
    clk.mark(1);                                              // [1] waiting for this tile's rel / k_node rows
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* rowp = st + (4 * g + j) * GA_PITCH + 8 * nn;
      unsigned h0, l0, h1, l1;
      split_pair(RR.x[j][0], RR.x[j][1], h0, l0);
      split_pair(RR.x[j][2], RR.x[j][3], h1, l1);
      *reinterpret_cast<uint2*>(rowp) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(rowp + GA_PLANE) = uint2{l0, l1};
      split_pair(KK.x[j][0], KK.x[j][1], h0, l0);
      split_pair(KK.x[j][2], KK.x[j][3], h1, l1);
      *reinterpret_cast<uint2*>(rowp + 128) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(rowp + 128 + GA_PLANE) = uint2{l0, l1};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    clk.mark(2);                                              // [2] splits + stage writes
    // ---- P1: the tile's logits, lane (head nn, g): edges 4g .. 4g+3
    f4 lg;
    {
      // three chains, one per term of the split product, joined at the end: twelve instructions on ONE accumulator issue one every
      // ~27 cycles, and nothing else of this wave can run beside them (one tile per wave, two waves per SIMD)
      f4 t0 = f4{0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
      const char* frag = st + nn * GA_PITCH + 16 * g;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const h8 ah = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(frag + 64 * s));
        const h8 al = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(frag + 64 * s + GA_PLANE));
        const h8 bh = __builtin_bit_cast(h8, b1h[s]), bl = __builtin_bit_cast(h8, b1l[s]);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, t2, 0, 0, 0);
      }
      lg = t0 + (t1 + t2);
    }
#ifdef TSDE_STAMPS
    asm volatile("" : "+v"(lg));
    clk.mark(3);                                              // [3] P1: fragment reads + 12 matrix instructions
#endif
    // ---- online softmax of this lane's head over the tile's 16 edges
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e0 + 4 * g + r >= end) lg[r] = -INFINITY;
      cm = fmaxf(cm, lg[r]);
    }
    cm = row_max(cm);                                         // over the four lane groups: the head's 16 edges
    if (__builtin_amdgcn_ballot_w64(cm > m + GA_LAZY) != 0ull) {
      const float mn = fmaxf(m, cm);
      const float sc = fast_exp(m - mn);                      // m = -inf on the first tile -> 0
      m = mn;
      spart *= sc;
      // the accumulators hold heads 4g + r in their registers: fetch those heads' factors from the lanes that own them
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sr = __shfl(sc, 4 * g + r);
#pragma unroll
        for (int c = 0; c < 8; ++c) O[c][r] *= sr;
      }
    }
    f4 ex;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ex[r] = fast_exp(lg[r] - m);                            // masked edges: exp(-inf) = 0
      spart += ex[r];
    }
#ifdef TSDE_STAMPS
    asm volatile("" : "+v"(ex));
    clk.mark(4);                                              // [4] softmax
#endif
    // ---- P2 over the tile's 16 edges with K = 32 instructions: the k-slots of a lane group are its 4 edges TWICE -- the B operand
    //      carries the rows' high pieces in slots 0..3 and their low pieces in slots 4..7, the A operand one piece of the weights in
    //      both halves: a_h (b_h + b_l), then a_l (b_h + b_l) -- all four terms of the split product in two instructions
    unsigned eh0, el0, eh1, el1;
    split_pair(ex[0], ex[1], eh0, el0);
    split_pair(ex[2], ex[3], eh1, el1);
    const h8 a2h = __builtin_bit_cast(h8, u4{eh0, eh1, eh0, eh1}), a2l = __builtin_bit_cast(h8, u4{el0, el1, el0, el1});
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned h0, l0, h1, l1;
      split_pair(RR.x[0][c], RR.x[1][c], h0, l0);
      split_pair(RR.x[2][c], RR.x[3][c], h1, l1);
      const h8 br = __builtin_bit_cast(h8, u4{h0, h1, l0, l1});
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h, br, O[c], 0, 0, 0);
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l, br, O[c], 0, 0, 0);
      split_pair(VV.x[0][c], VV.x[1][c], h0, l0);
      split_pair(VV.x[2][c], VV.x[3][c], h1, l1);
      const h8 bv = __builtin_bit_cast(h8, u4{h0, h1, l0, l1});
      O[4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h, bv, O[4 + c], 0, 0, 0);
      O[4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l, bv, O[4 + c], 0, 0, 0);
    }
#ifdef TSDE_STAMPS
    asm volatile("" : "+v"(O[0]), "+v"(O[7]));
    clk.mark(5);                                              // [5] P2: splits + 16 matrix instructions
    ++units;
#endif
  };
  // The pipeline, at the start of tile i:  rel rows of tile i + 2 (the HBM stream: two tiles of lead, three register sets),
  // k_node / v_node rows of tile i + 1 (gathers that mostly hit the L2: one tile of lead, two sets), source indices of tile i + 3
  // (they must be in registers when the gathers of their tile are issued).  Six tiles per trip of the loop so that every set index
  // is a compile-time constant.
  // Every fetch below is UNCONDITIONAL (past the segment's end the offsets clamp to its last row: a few redundant cache hits per
  // target): with loads under `if (tile < end)` the compiler can no longer count what is in flight across the branches and falls
  // back to s_waitcnt vmcnt(0) in front of each group -- the phase clocks showed 77 % of a tile's time in that wait.
  GaRows R[3], K[2], V[2];
  int idx[2][4];
  auto tile_at = [&](int i) { return beg + 16 * i; };
  if (beg < end) {
    fetch_idx(idx[0], tile_at(0));
    fetch_idx(idx[1], tile_at(1));
    fetch_rel(R[0], tile_at(0));
    fetch_rel(R[1], tile_at(1));
    fetch_kv(K[0], V[0], idx[0]);
    fetch_idx(idx[0], tile_at(2));
    for (int i0 = 0; tile_at(i0) < end; i0 += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + u;
        fetch_rel(R[(u + 2) % 3], tile_at(i + 2));
        fetch_kv(K[(u + 1) % 2], V[(u + 1) % 2], idx[(u + 1) % 2]);
        fetch_idx(idx[(u + 1) % 2], tile_at(i + 3));
        if (tile_at(i) < end) tile_step(R[u % 3], K[u % 2], V[u % 2], tile_at(i));
      }
    }
  }
  clk.mark(6);
  // ---- per target: normalise, lin_v_edge on the aggregated rel rows, store
  const float s = row_sum(spart);                             // lanes (head nn, every g): the head's sum
  const float inv = 1.0f / (s + 1e-16f);                      // PyG softmax denominator
  if (g == 0 && nn < 8) {
    hbuf[wv][nn] = inv;
    hbuf[wv][8 + nn] = s * inv;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (g < 2) {                                                // S_h[4 nn + c] of the heads 4g + r -> sbuf, normalised
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float iv = hbuf[wv][4 * g + r];
      *reinterpret_cast<f4*>(&sbuf[wv][4 * g + r][4 * nn]) = f4{O[0][r], O[1][r], O[2][r], O[3][r]} * iv;
    }
  }
  if (g == (nn >> 3)) {                                       // sum_e alpha v_node of node features 4 nn + c: head nn >> 1
    const int r = (nn >> 1) & 3;
    f4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = r == 0 ? O[4 + c][0] : (r == 1 ? O[4 + c][1] : (r == 2 ? O[4 + c][2] : O[4 + c][3]));
    *reinterpret_cast<f4*>(&obuf[wv][4 * nn]) = o * hbuf[wv][nn >> 1];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int h = lane >> 3;
  float out = fmaf(img[GAttnL::BVE + lane], hbuf[wv][8 + h], obuf[wv][lane]);
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
    const f4 wr = *reinterpret_cast<const f4*>(wve + lane * 64 + 4 * k4);
    const f4 sv = *reinterpret_cast<const f4*>(&sbuf[wv][h][4 * k4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
  }
  if (node < N) agg[node * 64 + lane] = out;
#ifdef TSDE_STAMPS
  clk.mark(7);                                                // [7] the per-target epilogue ([6]: loop overhead)
  if (lane == 0) clk.flush(g_stamps_gattn, units);
#endif
}

bool gattn_mm_enabled() {
  // OFF by default: measured at 32 x 256 agents it is on par with the lane-per-feature kernel, not ahead of it (0.177 against 0.167-0.170
  // ms per layer; 0.144 with every rel row served from the cache) -- HISTORY.md section 5, round 4.  TRAJSDE_GATTN_MM=1 selects it.
  static const bool v = []() { const char* e = getenv("TRAJSDE_GATTN_MM"); return e && atoi(e) != 0; }();
  return v;
}
int launch_global_attn_mm(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, hipStream_t st) {
  TS_LAUNCH_TAG("k_global_attn<8>", false, k_global_attn_mm, xcd_grid(cdiv(N, 4)), 256, 0, st, img, segptr, src, rel, q, kn, vn, N, agg);
  return TRAJSDE_OK;
}
#else
bool gattn_mm_enabled() {        // asked for in a build that does not carry it: say so at the launch instead of running something else
  static const bool v = []() { const char* e = getenv("TRAJSDE_GATTN_MM"); return e && atoi(e) != 0; }();
  return v;
}
int launch_global_attn_mm(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*, int64_t, float*,
                          hipStream_t) {
  return fail(TRAJSDE_ERR_UNSUPPORTED, "the matrix-core global attention is an alternative form of the fp16x3 build: load "
                                       "trajsde_amd/variants/libtrajsde_alt.so (TRAJSDE_LIB) for TRAJSDE_GATTN_MM=1");
}
#endif

}  // namespace tsde
