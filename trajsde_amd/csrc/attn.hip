// attn.hip -- the graph-attention family shared by AAEncoder (ENC:498-614), ALEncoder (ENC:693-797) and
// GlobalInteractorLayer (AGG:61-135), decomposed for gfx950 as
//   node kernel   : embeddings / norm1 / q (and k_node, v_node) per node           -> MFMA, 16 nodes per wave
//   edge kernel   : neighbour embedding (3 GEMMs) + k,v + per-head logits per edge -> MFMA, 16 edges per wave
//   segment kernel: per-target softmax + weighted sum over the CSR segment          -> one wave per target, lane = feature
//   update kernel : gate, lin_self, out_proj, residual, norm2                       -> MFMA
//   ffn kernel    : 64 -> 256 -> 64 + residual                                      -> MFMA
// Edges arrive sorted by target (prep.hip), so a target's messages are one contiguous slab: the segment
// softmax needs no atomics and is bitwise deterministic.  q is computed once per target instead of once
// per edge as the reference does (ENC:586, AGG:108); same values, 1/deg of the work.
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "attn_common.hpp"
#include "dropout.hpp"
#include "range.hpp"
#include "tile.hpp"

#include <type_traits>

#ifndef TSDE_SKEW_SLEEPS
#define TSDE_SKEW_SLEEPS 2      // x s_sleep(32) = 2 x 2048 cycles, about half of a tile's VALU + matrix time
#endif

namespace tsde {


// ------------------------------------------------------------------------------------------------ AA node
__global__ __launch_bounds__(512) void k_aa_center(const float* __restrict__ img_g, const float* __restrict__ x,
                                                   const float* __restrict__ x_fake, const float* __restrict__ rot,
                                                   const uint8_t* __restrict__ bos, const int32_t* __restrict__ orig,
                                                   int N, int Nt, int H, float* __restrict__ center,
                                                   float* __restrict__ cn, float* __restrict__ q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, AaCenterL::SIZE);
  using A = AaCenterL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t rows = int64_t(H) * Nt, ntiles = (rows + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < rows ? row : rows - 1;
    const int t = int(r / Nt), i = int(r - int64_t(t) * Nt), o = orig[i];
    const float* xp = i < N ? x + (int64_t(i) * H + t) * 2 : x_fake + (int64_t(i - N) * H + t) * 2;
    const f4 R = *reinterpret_cast<const f4*>(rot + 4 * o);
    const float x0 = xp[0], x1 = xp[1];
    const float r0 = x0 * R[0] + x1 * R[2], r1 = x0 * R[1] + x1 * R[3];          // x_t @ R_n  (ENC:550-552)
    f4 a[4], b[4];
    linear_in2(a, r0, r1, lds + A::W0, lds + A::B0, L.g);
    layer_norm<4>(a, lds + A::G1, lds + A::E1, L.g);
    relu<4>(a);
    linear<4, 4>(b, a, lds + A::W3, lds + A::B3, L);
    layer_norm<4>(b, lds + A::G4, lds + A::E4, L.g);
    relu<4>(b);
    linear<4, 4>(a, b, lds + A::W6, lds + A::B6, L);
    layer_norm<4>(a, lds + A::G7, lds + A::E7, L.g);
    if (bos[int64_t(o) * H + t]) load_vec<4>(a, lds + A::BOS + t * 64, L.g);      // ENC:554-556
    if (row < rows) store_row(a, center, row, L.g);
    layer_norm<4>(a, lds + A::N1G, lds + A::N1B, L.g);                            // norm1 (ENC:563)
    if (row < rows) store_row(a, cn, row, L.g);
    linear<4, 4>(b, a, lds + A::WQ, lds + A::BQ, L);
    if (row < rows) store_row(b, q, row, L.g);
  }
}

// ------------------------------------------------------------------------------------------------ edges
// AA / AL: embedding -> lin_k | lin_v -> logits with q[dst], v.   X6: bf16x6 split-precision matrix products
template <bool X6>
__global__ __launch_bounds__(1024) void k_edge_kv(const float* __restrict__ img_g, const float* __restrict__ geom,
                                                 const int32_t* __restrict__ dst, const float* __restrict__ q, int64_t E,
                                                 float* __restrict__ logits, float* __restrict__ v, int heads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = typename std::conditional<X6, EdgeL6, EdgeL>::type;
  stage_blob(lds, img_g, EL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (E + 15) / 16;
  // The waves of a SIMD (wave, wave+4, ...) run the same VALU-stage / matrix-stage stream; started together they stay in
  // phase and the two pipes take turns.  Every other one of them starts about half a tile late, once: from then on one is in
  // a VALU stage while its neighbour is on the matrix cores.
  if ((wave >> 2) & 1) {
    for (int i = 0; i < TSDE_SKEW_SLEEPS; ++i) __builtin_amdgcn_s_sleep(32);
  }
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t e = tile * 16 + L.n, ec = e < E ? e : E - 1;
    const f4 ge = *reinterpret_cast<const f4*>(geom + 4 * ec);
    const int d = dst[ec];
    f4 emb[4], kv[8], qv[4];
    load_row(qv, q, d, L.g);
    edge_embed<X6>(emb, ge, lds, L);
    if constexpr (X6) linear_x6<8, 4>(kv, emb, lds + EL::WKV, lds + EL::BKV, L);
    else linear<8, 4>(kv, emb, lds + EL::WKV, lds + EL::BKV, L);
    f4 k[4] = {kv[0], kv[1], kv[2], kv[3]};
    f4 vv[4] = {kv[4], kv[5], kv[6], kv[7]};
    store_logits(qv, k, logits, e, e < E, L, heads);
    if (e < E) store_row(vv, v, e, L.g);
  }
}

// The split-precision edge kernel with two 16-edge tiles per wave (32 rows): every LDS weight fragment feeds both tiles'
// matrix-core instructions (linear_acc_x6_2), halving the ds_read_b128 issue slots per tile.  SQ counters say the one-tile
// kernel is instruction-issue bound (per SIMD ~1.0 instruction-issue utilisation summed over its waves: ~1060 VALU, 240
// MFMA and 192 LDS instructions per tile), so the saved slots are the gain.  Per tile the arithmetic and its order are
// those of k_edge_kv<true>: same bits.  (A variant that ran the two tiles half a stage apart with sched_group_barrier
// interleaving was slower: 256 VGPRs with spills, 2 waves per SIMD.)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_edge_kv2(const float* __restrict__ img_g, const float* __restrict__ geom,
                                                      const int32_t* __restrict__ dst, const float* __restrict__ q, int64_t E,
                                                      float* __restrict__ logits, float* __restrict__ v, int heads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL6;
  stage_blob(lds, img_g, EL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t npairs = (E + 31) / 32;
  if ((wave >> 2) & 1) {                                   // see k_edge_kv: start every other wave of a SIMD half a tile late
    for (int i = 0; i < 2 * TSDE_SKEW_SLEEPS; ++i) __builtin_amdgcn_s_sleep(32);
  }
  for (int64_t pair = int64_t(blockIdx.x) * waves + wave; pair < npairs; pair += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t e0 = pair * 32 + L.n, e1 = e0 + 16;
    const int64_t c0 = e0 < E ? e0 : E - 1, c1 = e1 < E ? e1 : E - 1;
    const f4 g0 = *reinterpret_cast<const f4*>(geom + 4 * c0);
    const f4 g1 = *reinterpret_cast<const f4*>(geom + 4 * c1);
    const int d0 = dst[c0], d1 = dst[c1];
    f4 emb0[4], emb1[4], kv0[8], kv1[8], qv[4];
    edge_embed2_x6(emb0, emb1, g0, g1, lds, L);
    load_vec<8>(kv0, lds + EL::BKV, L.g);
    load_vec<8>(kv1, lds + EL::BKV, L.g);
    linear_acc_x6_2<8, 4>(kv0, kv1, emb0, emb1, lds + EL::WKV, L.lane);
    {
      load_row(qv, q, d0, L.g);
      f4 k[4] = {kv0[0], kv0[1], kv0[2], kv0[3]};
      f4 vv[4] = {kv0[4], kv0[5], kv0[6], kv0[7]};
      store_logits(qv, k, logits, e0, e0 < E, L, heads);
      if (e0 < E) store_row(vv, v, e0, L.g);
    }
    {
      load_row(qv, q, d1, L.g);
      f4 k[4] = {kv1[0], kv1[1], kv1[2], kv1[3]};
      f4 vv[4] = {kv1[4], kv1[5], kv1[6], kv1[7]};
      store_logits(qv, k, logits, e1, e1 < E, L, heads);
      if (e1 < E) store_row(vv, v, e1, L.g);
    }
  }
}
template __global__ void k_edge_kv2<512>(const float*, const float*, const int32_t*, const float*, int64_t, float*, float*, int);
template __global__ void k_edge_kv2<768>(const float*, const float*, const int32_t*, const float*, int64_t, float*, float*, int);

// ------------------------------------------------------------------------------------------------ fused edge attention
// k_edge_attn2: the edge kernel above with the segment softmax + aggregation folded in, so that the per-edge v rows and
// logits (288 B per edge, written once and read once: 2 x 2 GB per forward at 32 scenes x 256 agents) never reach HBM.
//
// Layout ("one edge stream per tile row").  The compacted edge list is sorted by target.  It is cut into `nstreams` chunks
// of consecutive edges -- 2C - Cy or Cy of them, by the stream's place in its block of 256 (kernels.hpp StreamMap: the waves a SIMD
// serves first get the longer ones) --; row n of a wave's tile walks chunk `stream(n)` front to back, one
// edge per tile iteration.  A tile iteration is the same 16-row matrix work as before (embedding, lin_k | lin_v), but its 16
// rows are 16 DIFFERENT streams -- so the online softmax of a row is lane-local: lane (n, g) owns features 16jt+4g+c of row n,
// which belong to head 2jt + (g>>1), and keeps the running (m, s) of those heads and the running weighted sum of its 16 value
// features in registers across iterations.  No cross-lane reduction beyond the per-head dot product that the logits always
// needed, no segmented scan, no atomics.  The 32 streams of a wave have the same length, so all its rows run the same number of
// iterations, whatever the in-degree distribution.
//
// When a row's target changes (and at the end of its chunk) the row flushes one RECORD -- acc[64] | m[16] | s[16], 384 B -- to
// slot (target + stream): along the edge list either the target or the stream advances between consecutive records, so the
// slot is unique and the records of a target sit at consecutive slots target + (first stream .. last stream).  A target whose
// segment lies inside one chunk produces one record; one that straddles a chunk boundary produces one per chunk it touches.
// k_seg_merge (one wave per target, lane = feature) combines them in stream order and normalises like PyG's softmax
// (sum + 1e-16).  The chunking depends only on E, so results are bitwise reproducible and independent of the input edge order.
#if TSDE_SPLIT_H3   // the fused kernel exists in the fp16x3 build only: its image + query slots do not fit LDS with three bf16 planes
#ifdef TSDE_EDGE_STAMPS
// Diagnostic build only (tools/edge_phase_stamps.py): cycles (s_memtime) that one wave of every workgroup spends in each phase of
// an iteration of k_edge_attn2, summed over the launch.  The stamp's own s_waitcnt drains the wave's LDS queue at every mark, so
// the total runs a little slower than the shipped kernel; the shares are what it is for.
__device__ unsigned long long g_edge_stamps[16];
__device__ unsigned long long g_edge_wg[4 * 512];
struct PhaseStamps {
  unsigned long long last, real0, acc[12];
  __device__ __forceinline__ void start() {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0;
    last = __builtin_amdgcn_s_memtime();
    real0 = __builtin_amdgcn_s_memrealtime();
  }
#ifdef TSDE_EDGE_STAMPS_LIGHT      // the loop as a whole only (no stamp inside the body: the shipped stream, undisturbed): cycles and clock
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void finish() { acc[0] = __builtin_amdgcn_s_memtime() - last; }
#else
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[i] += t - last;
    last = t;
  }
  __device__ __forceinline__ void finish() {}
#endif
};
#endif
struct SegState {
  f4 acc[4], m, s;
};
__device__ __forceinline__ void seg_reset(SegState& S) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) S.acc[jt] = f4{0.f, 0.f, 0.f, 0.f};
  S.m = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  S.s = f4{0.f, 0.f, 0.f, 0.f};
}
// per-head logits of the 16 rows of a tile: lane (n, g) gets, for jt = 0..3, the logit of the head its features 16jt+4g+c
// belong to (8 heads: head 2jt + (g>>1), completed by the partner lane group g^1; 4 heads: head jt, all four groups)
// Round 6: the running softmax of the fused kernel works in LOG2 units -- the logit scale carries log2(e), so a weight is 2^(l - m)
// with no multiply in front of v_exp_f32 (two per head group and edge before); a record's maxima go out in natural units again
// (seg_flush), so its readers (k_seg_merge, k_node_update, the training tape) are unchanged.
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;
__device__ __forceinline__ float logit_scale(int heads) {
  return (heads == 4 ? 0.25f : INV_SQRT_DH) * (TSDE_R6_SOFTMAX ? LOG2E_F : 1.0f);
}
// the logit of head group jt of a tile's 16 rows
__device__ __forceinline__ float head_logit1(const f4& qv, const f4& k, int heads) {
  float p = qv[0] * k[0];
#pragma unroll
  for (int c = 1; c < 4; ++c) p = fmaf(qv[c], k[c], p);
  p = xor16_sum(p);
  if (heads == 4) p = xor32_sum(p);
  return p * logit_scale(heads);
}
__device__ __forceinline__ f4 head_logits(const f4 (&qv)[4], const f4 (&k)[4], int heads) {
  f4 lg;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) lg[jt] = head_logit1(qv[jt], k[jt], heads);
  return lg;
}
// `keep`: attention dropout (ENC:592): the softmax is normalised over ALL edges (s), the weighted sum takes the kept ones
// scaled by 1/(1-p); f4{1,1,1,1} when dropout is off
#if TSDE_R6_SOFTMAX
// fmaxf() costs three instructions here: the compiler quiets both inputs first (v_max x, x) because it cannot know that neither is a
// signalling NaN.  The plain instruction: both inputs come from ordinary vector instructions (selects, multiplies), its result
// goes to subtractions -- no matrix or transcendental result is involved, so no wait state the compiler would have to see.
__device__ __forceinline__ float max_plain(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// The update in three parts, so that a kernel can run the first two before the value rows exist (k_edge_attn2: inside the lin_v product):
// front: new maximum, scale of the old sums, weight of the edge, running sum;  scale: sums *= sc;  add: sums += w v.
// The sums are updated IN PLACE (multiply, then fma into the same register): written fma(sum, sc, w v) the compiler picked the
// two-address v_fmac with the product as the addend and copied all 16 sums of a tile back at every back-edge (32 v_mov an iteration).
__device__ __forceinline__ void seg_front1(SegState& S, int jt, float lg, float keep, float& sc, float& exk) {
  const float mn = max_plain(S.m[jt], lg);
  sc = __builtin_amdgcn_exp2f(S.m[jt] - mn);               // first edge of a segment: 2^(-inf) = 0
  const float ex = __builtin_amdgcn_exp2f(lg - mn);
  S.m[jt] = mn;
  S.s[jt] = fmaf(S.s[jt], sc, ex);
  exk = ex * keep;
}
__device__ __forceinline__ void seg_scale1(SegState& S, int jt, float sc) {
#pragma unroll
  for (int c = 0; c < 4; ++c) S.acc[jt][c] *= sc;
}
__device__ __forceinline__ void seg_add1(SegState& S, int jt, float exk, const f4& v) {
#pragma unroll
  for (int c = 0; c < 4; ++c) S.acc[jt][c] = fmaf(exk, v[c], S.acc[jt][c]);
}
__device__ __forceinline__ void seg_update(SegState& S, const f4& lg, const f4 (&v)[4], const f4& keep) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float sc, exk;
    seg_front1(S, jt, lg[jt], keep[jt], sc, exk);
    seg_scale1(S, jt, sc);
    seg_add1(S, jt, exk, v[jt]);
  }
}
#else
__device__ __forceinline__ void seg_update(SegState& S, const f4& lg, const f4 (&v)[4], const f4& keep) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const float mn = fmaxf(S.m[jt], lg[jt]);
    const float sc = fast_exp(S.m[jt] - mn);             // first edge of a segment: exp(-inf) = 0
    const float ex = fast_exp(lg[jt] - mn);
    S.m[jt] = mn;
    S.s[jt] = fmaf(S.s[jt], sc, ex);
    const float exk = ex * keep[jt];
    // (The compiler picks v_fmac here -- result in the addend's register -- and copies the 16 sums of a tile back to their loop registers
    //  at every back-edge: 32 v_mov an iteration.  The three-address v_fma_f32 written in assembly removes them but needs an s_nop in
    //  front -- `sc` comes straight from v_exp_f32 and the compiler inserts no wait state for a reader it cannot see into; without it the
    //  sums were wrong, caught by the golden parity test -- and measured 0.795 against 0.782 ms on one box: not kept.)
#pragma unroll
    for (int c = 0; c < 4; ++c) S.acc[jt][c] = fmaf(S.acc[jt][c], sc, exk * v[jt][c]);
  }
}
#endif
__device__ __forceinline__ void seg_flush(const SegState& S, float* __restrict__ rec, int64_t slot, int g) {
  float* r = rec + slot * SEG_REC;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<f4*>(r + 16 * jt + 4 * g) = S.acc[jt];
  *reinterpret_cast<f4*>(r + 64 + 4 * g) = TSDE_R6_SOFTMAX ? S.m * LN2_F : S.m;      // m / s of lane group g at [4g + jt]; maxima in natural units
  *reinterpret_cast<f4*>(r + 80 + 4 * g) = S.s;
}

// SAVE (training path): the embedding rows are also written to emb_out [E,64] -- the tape of the attention backward.
// NT: row tiles per wave (streams per wave = 16 NT); a workgroup walks 256 streams.  Shipped: 8 waves x 2 tiles, two waves per
// SIMD.  (Measured alternatives, 32 x 256 agents: 4 waves x 4 tiles at one wave per SIMD -- every weight fragment feeding 12
// matrix instructions, accumulators in AGPRs -- 0.575 ms per launch against 0.47 ms; 4 waves x 2 tiles 0.60 ms.)
// LIST (0: agent-agent, 1: agent-lane) only names the instantiation, so that a kernel trace lists the two launches of a forward
// -- 6.85 M edges and 0.2 M edges on the metric workload -- separately.
// H8: the head count is the compile-time 8 of the shipped configurations (the launch passes 8): with it a run-time value the logits of
// every 16 features end in a branch on it, eight a tile pair, and the scheduler cannot move anything across them.
template <int NT, bool DROP, bool SAVE, int LIST, bool H8>
__global__ __launch_bounds__(1024 / NT) void k_edge_attn2(const float* __restrict__ img_g, const float* __restrict__ geom,
                                                          const int32_t* __restrict__ dst, const float* __restrict__ q, EdgeCount ec, int C_host,
                                                          float* __restrict__ rec, int heads_arg, const int32_t* __restrict__ segptr, DropArg drop,
                                                          float* __restrict__ emb_out) {
  const int heads = H8 ? 8 : heads_arg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL6F;
  const int64_t E = edge_count(ec);
  const int C = stream_len(ec, E, C_host);
  if (E <= 0) return;                                      // (only reachable when the count lives on the device)
  stage_blob(lds, img_g, EL::LDS_SIZE);
  const Lane L;
#if TSDE_R6_TOP
  // (the wave index as a SCALAR: with it the stream length below is one, and the loop counter and its back-edge are scalar instructions)
  const int waves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
#else
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
#endif
  const StreamMap smap = stream_map(C);
  const int64_t nstreams = stream_count(E, C);
  const int64_t wid = xcd_block() * waves + wave;          // xcd_grid launch: consecutive streams (same snapshot, same scene) share an L2
  if (wid * (16 * NT) >= nstreams) return;                 // whole wave beyond the list (uniform)
  if (NT == 2 && ((wave >> 2) & 1)) {                      // see k_edge_kv: start every other wave of a SIMD half a tile late
    for (int i = 0; i < 2 * TSDE_SKEW_SLEEPS; ++i) __builtin_amdgcn_s_sleep(32);
  }
  SegState S[NT];
  int64_t sid[NT], base_e[NT];                             // the tiles' streams of this lane's rows, and their first edges
  int cur[NT], rank0[NT];                                  // current target; DROP: first edge of that target (mask counter = rank)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    seg_reset(S[t]);
    sid[t] = wid * (16 * NT) + 16 * t + L.n;
    base_e[t] = stream_base(smap, sid[t]);
    cur[t] = -1;
    rank0[t] = 0;
  }
  // The query row of a tile row changes only when the row's target does (every ~deg iterations), but NT x 16 registers to keep
  // it do not fit beside the softmax state, and re-gathering 256 B per row and iteration from L2 was 1.4 GB of fabric reads per
  // launch (FETCH_SIZE).  So each lane parks ITS 16 query values in LDS -- a private slot, [tile][jt][lane][4], conflict-free
  // b128 -- when its row's target changes, and reads them back every iteration.
  float* qs = lds + EL::LDS_SIZE + wave * (NT * 4 * 64 * 4);
  const f4 one4 = f4{1.f, 1.f, 1.f, 1.f};
  // an iteration's geometry and targets are loaded one iteration ahead: the first thing an iteration does is compare its targets
  // with the current ones, and nothing would cover that round trip
  f4 ng[NT];
  int nd[NT];
#if TSDE_R6_TOP
  // Round 6: the edge walk in 32-bit arithmetic.  geometry / targets are read through buffer descriptors whose extent is the list
  // (a read past it returns zeros: no index clamp), at byte offsets that advance by 16 / 4 an iteration: two additions and one
  // compare a tile, where the 64-bit form spent ~12 vector instructions a tile (adds, compares, selects, address pairs).
  // (host side: fused_edge_attention refuses lists of 2^28 edges or more)
  const __amdgpu_buffer_rsrc_t geom_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(geom), 0, int(unsigned(E) * 16u), 0x00020000);
  const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(dst), 0, int(unsigned(E) * 4u), 0x00020000);
  unsigned goff[NT], doff[NT];                             // byte offsets of the row's current edge in the two lists
  const unsigned E16 = unsigned(E) * 16u;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const unsigned e0 = unsigned(base_e[t] < E ? base_e[t] : E);
    goff[t] = e0 * 16u;
    doff[t] = e0 * 4u;
    ng[t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(geom_rs, int(goff[t]), 0, 0));
    nd[t] = int(__builtin_amdgcn_raw_buffer_load_b32(dst_rs, int(doff[t]), 0, 0));
  }
#else
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t c = base_e[t] < E ? base_e[t] : E - 1;
    ng[t] = *reinterpret_cast<const f4*>(geom + 4 * c);
    nd[t] = dst[c];
  }
#endif
#ifdef TSDE_EDGE_STAMPS
  PhaseStamps st;
  st.start();
#else
  NoStamps st;
#endif
  const int Cw = stream_length(smap, wid * (16 * NT));     // a wave's 16 NT streams are equally long
  prioritize_younger_half();
  for (int it = 0; it < Cw; ++it) {
    keep_lds_reads_here();
    f4 ge[NT];
    int d[NT];
    bool ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#if TSDE_R6_TOP
      ok[t] = goff[t] < E16;                               // only the last stream is short; streams >= nstreams are empty
#else
      ok[t] = base_e[t] + it < E;                          // only the last stream is short; streams >= nstreams are empty
#endif
      ge[t] = ng[t];
      d[t] = nd[t];
      // the loads of the previous iteration are consumed HERE, before this iteration issues anything: the compiler's counted
      // waits further down would otherwise also cover the query-row transfers below, which it does not know about
      asm volatile("" : "+v"(ge[t]), "+v"(d[t]));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#if TSDE_R6_TOP
      goff[t] += 16u;                                      // (one past a stream's end: loaded, never used; past the list: zeros)
      doff[t] += 4u;
      ng[t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(geom_rs, int(goff[t]), 0, 0));
      nd[t] = int(__builtin_amdgcn_raw_buffer_load_b32(dst_rs, int(doff[t]), 0, 0));
#else
      const int64_t e = base_e[t] + it;
      const int64_t c = e + 1 < E ? e + 1 : E - 1;         // (one past a stream's end: loaded, never used)
      ng[t] = *reinterpret_cast<const f4*>(geom + 4 * c);
      nd[t] = dst[c];
#endif
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // The row's target changes: its finished segment part leaves.  The state is NOT reset inside the branch -- only its running
      // maxima go to -inf, by a select outside it: the next update then scales s and the sums by exp(-inf) = 0 itself (seg_update),
      // and the 24 state registers of a tile stay out of the branch's merge (24 zero moves + 24 copies a tile and iteration before).
      const bool chg = ok[t] && d[t] != cur[t];
      if (chg) {
        if (cur[t] >= 0) seg_flush(S[t], rec, int64_t(cur[t]) + sid[t], L.g);
        cur[t] = d[t];
        if (DROP) rank0[t] = segptr[d[t]];
        // the new target's query row goes from L2 straight into the lane's LDS slots (global_load_lds_dwordx4: lane l's 16 bytes land
        // at the wave-uniform base + 16 l, which IS the slot layout [tile][jt][lane][4]): no registers, and nothing waits for it
        // until the logits at the end of the iteration (s_waitcnt vmcnt there)
        const float* qrow = q + int64_t(d[t]) * D + 4 * L.g;
        // (through assembly: given the builtin, the compiler orders every later LDS read of the kernel -- the weight fragments --
        //  behind the transfer with s_waitcnt vmcnt(0), which is the very wait this is here to remove)
        const unsigned slot = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) float*)(qs + 4 * t * 256))));
        unsigned m0_keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %5\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %4, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(m0_keep)
                     : "v"(qrow), "v"(qrow + 16), "v"(qrow + 32), "v"(qrow + 48), "s"(slot)
                     : "memory", "scc");
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) S[t].m[jt] = chg ? -INFINITY : S[t].m[jt];
    }
    st.mark(0);                                            // loads, target changes (record flush, query row)
#if TSDE_R6_ATOMS && TSDE_R6_SOFTMAX
    // ---- round 6: the same arithmetic per edge (products, LayerNorms and softmax element for element those of the round-5 stream,
    // edge_embed_fused_n + linear_acc_x6_n + seg_update), re-ordered so that a wave's matrix instructions have vector work beside them:
    //   * both first layers of both tiles are 8 NT matrix instructions back to back (round 5: each instruction followed by the ~8
    //     wait states of its result's first reader, 16 times an iteration);
    //   * the ReLU + operand split of branch B issues between the matrix instructions of the W_A product;
    //   * lin_k is finished before lin_v starts (k-rows first: same order per accumulator), and the logits, the new maxima, the two
    //     exponentials and the rescaling of the running sums of one (tile, head group) issue inside each step of the lin_v product;
    //     behind it only sums += w v is left.
    {
      u4 opA[NT], opB[NT];
      {
        float xa[NT], xb[NT], xc[NT], xd[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          xa[t] = ge[t][0];
          xb[t] = ge[t][1];
          xc[t] = ge[t][2];
          xd[t] = ge[t][3];
        }
        in2_operands_n<NT>(opA, xa, xb, lds + EL::A_C);
        in2_operands_n<NT>(opB, xc, xd, lds + EL::B_C);
      }
      f4 ya[NT][4], yb[NT][4];
      in2_mfma_raw_n<NT>(ya, opA, lds + EL::A_F, L.lane);
      in2_mfma_raw_n<NT>(yb, opB, lds + EL::B_F, L.lane);
      __builtin_amdgcn_sched_barrier(0);
      st.mark(1);
      u4 xh[NT][2], xl[NT][2], bh[NT][2], bl[NT][2];
      f4 sm[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) relu_split_kstep(ya[t][2 * ks], ya[t][2 * ks + 1], xh[t][ks], xl[t][ks]);
        load_vec<4>(sm[t], lds + EL::B3, L.g);
      }
      product_x6_n<NT, 4>(sm, xh, xl, lds + EL::WA3, L.lane, [&](auto I) {
        constexpr int i = decltype(I)::value, per = 8 / (2 * NT);       // 2 NT atoms (tile, k-step) over the 8 steps
        if constexpr (i % per == per - 1) {
          constexpr int a = i / per, t = a >> 1, ks = a & 1;
          relu_split_kstep(yb[t][2 * ks], yb[t][2 * ks + 1], bh[t][ks], bl[t][ks]);
        }
      });
      st.mark(2);
      st.mark(3);
      product_x6_n<NT, 4>(sm, bh, bl, lds + EL::WB3, L.lane, NoAtoms{});
      st.mark(4);
      float r[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) r[t] = centred_rstd(sm[t]);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG0 + 16 * jt + 4 * L.g);
        const f4 be = *reinterpret_cast<const f4*>(lds + EL::AE0 + 16 * jt + 4 * L.g);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) sm[t][jt][c] = fmaxf(fmaf(sm[t][jt][c] * r[t], ga[c], be[c]), 0.f);
      }
      split_rows_n<NT>(xh, xl, sm);
      f4 emb[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) load_vec<4>(emb[t], lds + EL::B2, L.g);
      st.mark(5);
      product_x6_n<NT, 4>(emb, xh, xl, lds + EL::W2, L.lane, NoAtoms{});
      st.mark(6);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float q4 = centred_rstd(emb[t]);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int c = 0; c < 4; ++c) emb[t][jt][c] *= q4;     // (y - mean) * rstd of the last LayerNorm; gamma / beta live downstream
      }
      st.mark(7);
      if (SAVE) {                                            // the tape holds the embedding rows proper
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f4 tr[4];
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) {
            const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG3 + 16 * jt + 4 * L.g);
            const f4 be = *reinterpret_cast<const f4*>(lds + EL::AE3 + 16 * jt + 4 * L.g);
            tr[jt] = emb[t][jt] * ga + be;
          }
          if (ok[t]) store_row(tr, emb_out, base_e[t] + it, L.g);
        }
      }
      split_rows_n<NT>(xh, xl, emb);
      // k / v without their constant parts (EdgeL6F): q . CK shifts all logits of a (target, head) alike, and sum_e alpha_e CV = CV
      // is added by k_seg_merge.  With attention dropout the kept weights do not sum to one, so v carries CV here.
      f4 kk[NT][4], vv[NT][4], keep[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          kk[t][jt] = f4{0.f, 0.f, 0.f, 0.f};
          vv[t][jt] = DROP ? *reinterpret_cast<const f4*>(img_g + EL::CV + 16 * jt + 4 * L.g) : f4{0.f, 0.f, 0.f, 0.f};
        }
        keep[t] = DROP ? drop_attn_row(drop, uint32_t(d[t]), uint32_t(int(base_e[t] + it) - rank0[t]), L.g, heads) : one4;
      }
      product_x6_n<NT, 4>(kk, xh, xl, lds + EL::WKV, L.lane, NoAtoms{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the query rows sent to LDS at the top of the iteration have landed
      f4 sc[NT], exk[NT];
      product_x6_n<NT, 4>(vv, xh, xl, lds + EL::WKV + 4 * 2 * 512, L.lane, [&](auto I) {
        constexpr int i = decltype(I)::value, per = 8 / (4 * NT);       // 4 NT atoms (tile, head group) over the 8 steps
        if constexpr (i % per == per - 1) {
          constexpr int a = i / per, t = a >> 2, jt = a & 3;
          const f4 qv = *reinterpret_cast<const f4*>(qs + ((4 * t + jt) * 64 + L.lane) * 4);
          float lg = head_logit1(qv, kk[t][jt], heads);
          // Branch-free: a row beyond the list (the tail of the last stream; rows of streams >= nstreams) enters with logit -inf --
          // weight 2^(-inf) = 0, running maximum and scale unchanged.  A row that never had an edge turns its state into NaNs
          // (2^(-inf + inf)); it is never flushed (cur < 0).
          if (!ok[t]) lg = -INFINITY;
          float sc1, exk1;
          seg_front1(S[t], jt, lg, keep[t][jt], sc1, exk1);
          seg_scale1(S[t], jt, sc1);
          sc[t][jt] = sc1;
          exk[t][jt] = exk1;
        }
      });
      st.mark(8);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) seg_add1(S[t], jt, exk[t][jt], vv[t][jt]);
      (void)sc;
    }
    st.mark(9);                                            // the rest of the online softmax
#else
    st.mark(0);                                            // loads, target changes (record flush, query row)
    f4 emb[NT][4], kv[NT][8];
    edge_embed_fused_n<NT>(emb, ge, lds, L, st);            // (y - mean) * rstd of the last LayerNorm; gamma / beta live downstream
    if (SAVE) {                                            // the tape holds the embedding rows proper
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f4 tr[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG3 + 16 * jt + 4 * L.g);
          const f4 be = *reinterpret_cast<const f4*>(lds + EL::AE3 + 16 * jt + 4 * L.g);
          tr[jt] = emb[t][jt] * ga + be;
        }
        if (ok[t]) store_row(tr, emb_out, base_e[t] + it, L.g);
      }
    }
    // k / v without their constant parts (EdgeL6F): q . CK shifts all logits of a (target, head) alike, and sum_e alpha_e CV = CV
    // is added by k_seg_merge.  With attention dropout the kept weights do not sum to one, so v carries CV here.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int jo = 0; jo < 8; ++jo) kv[t][jo] = f4{0.f, 0.f, 0.f, 0.f};
      if (DROP) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) kv[t][4 + jt] = *reinterpret_cast<const f4*>(img_g + EL::CV + 16 * jt + 4 * L.g);
      }
    }
    linear_acc_x6_n<NT, 8, 4>(kv, emb, lds + EL::WKV, L.lane);
    st.mark(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the query rows sent to LDS at the top of the iteration have landed
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f4 qv[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) qv[jt] = *reinterpret_cast<const f4*>(qs + ((4 * t + jt) * 64 + L.lane) * 4);
      const f4 k[4] = {kv[t][0], kv[t][1], kv[t][2], kv[t][3]};
      const f4 vv[4] = {kv[t][4], kv[t][5], kv[t][6], kv[t][7]};
      f4 lg = head_logits(qv, k, heads);
      // Branch-free: a row beyond the list (the tail of the last stream; rows of streams >= nstreams) enters with logit -inf --
      // weight exp(-inf) = 0, running maximum and scale unchanged -- instead of skipping the update under an exec mask: one basic
      // block, so the scheduler interleaves the two tiles' dependent chains (dot -> lane swaps -> max -> exp -> fma).  A row that
      // never had an edge turns its state into NaNs (exp(-inf + inf)); it is never flushed (cur < 0).
      if (!ok[t]) lg = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      seg_update(S[t], lg, vv, DROP ? drop_attn_row(drop, uint32_t(d[t]), uint32_t(int(base_e[t] + it) - rank0[t]), L.g, heads) : one4);
    }
    st.mark(9);                                            // logits + online softmax
#endif
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (cur[t] >= 0) seg_flush(S[t], rec, int64_t(cur[t]) + sid[t], L.g);
#ifdef TSDE_EDGE_STAMPS
  st.finish();
  if (LIST == 0 && !DROP && !SAVE && (wave == 0 || wave == 5) && L.lane == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) atomicAdd(&g_edge_stamps[i], st.acc[i]);
    atomicAdd(&g_edge_stamps[10], (unsigned long long)C);
    const unsigned long long real = __builtin_amdgcn_s_memrealtime() - st.real0;
    atomicAdd(&g_edge_stamps[11], real);                   // 100 MHz ticks over the same loop: the clock
    atomicMax(&g_edge_stamps[12], real);                   // slowest / fastest stamped wave of the launches (10 ns units)
    atomicMin(&g_edge_stamps[13], real);
    atomicAdd(&g_edge_stamps[14], 1ull);
    if (wave == 0) {                                       // per workgroup (last launch): loop time, phase-0 cycles, XCC, CU
      unsigned xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      const int b = int(xcd_block()) & 511;
      g_edge_wg[4 * b + 0] = real;
      g_edge_wg[4 * b + 1] = st.acc[0];
      g_edge_wg[4 * b + 2] = xcc;
      g_edge_wg[4 * b + 3] = hwid;
    }
  }
#endif
}
#ifdef TSDE_EDGE_STAMPS
}  // namespace tsde
extern "C" int trajsde_debug_edge_stamps(unsigned long long* host16, int reset) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(tsde::g_edge_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    z[13] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(tsde::g_edge_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int trajsde_debug_edge_wg(unsigned long long* host2048) {
  return hipMemcpyFromSymbol(host2048, HIP_SYMBOL(tsde::g_edge_wg), 4 * 512 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace tsde {
#endif
template __global__ void k_edge_attn2<2, false, false, 0, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, false, 0, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, false, 1, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, false, 1, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
#ifndef TSDE_PRODUCT        // alternative forms: trajsde_amd/variants/libtrajsde_alt.so only (edge32.hip)
template __global__ void k_edge_attn2<1, false, false, 0, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<1, false, false, 1, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
#endif
template __global__ void k_edge_attn2<2, true, false, 0, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, false, 0, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, false, 1, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, false, 1, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, true, 0, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, true, 0, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, true, 1, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, false, true, 1, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, true, 0, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, true, 0, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, true, 1, false>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
template __global__ void k_edge_attn2<2, true, true, 1, true>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int, const int32_t*, DropArg, float*);
// ------------------------------------------------------------------------------------------------ pipelined form
#ifndef TSDE_PRODUCT        // alternative form: trajsde_amd/variants/libtrajsde_alt.so only (edge32.hip)
// k_edge_attn2p: the inference kernel above (same streams, same records, the same arithmetic per edge: bit-identical) with the
// vector work of one tile issued BETWEEN the matrix instructions of the other, by construction.
//
// What the measurements say (tools/edge_phase_stamps.py, tools/microbench/coexec.hip): a wave is an in-order stream whose
// vector stages (LayerNorms, operand splits, softmax) and matrix stages (the five products) alternate and depend on each
// other, so within a wave nothing overlaps; the SIMD then serves the older of its two waves at nearly the speed of a lone
// wave, and a lone wave takes ~10.4 k cycles per iteration for 4.1 k cycles of matrix pipe and 3.6 k of vector issue.  One
// stream that issues a matrix instruction followed by independent vector instructions hides the vector ones in the matrix
// instruction's 16 cycles (microbenchmark: 48 x (mfma + 4 fma) = 21.5 cycles each, against 16 + 16 issued apart).
//
// So the wave's two tiles run ONE STAGE APART: tile B lags tile A by one stage of the chain
//     V1 in2 operands | M1 in2 layers | V2 relu, split | M2 W_A, W_B | V3 LN, relu, split | M3 W_2 | V4 LN, split | M4 lin_k|lin_v | V5 softmax
// and every matrix stage of one tile is a cluster whose steps (3 matrix instructions each) are followed by "atoms" of the
// other tile's vector stage -- small independent pieces (one k-step's split, one head's softmax update, ...), spread evenly
// over the steps and pinned there (sched_barrier).  A.Vk pairs with B.Mk, B.V(k+1) with A.Mk; the one vector stage left
// over (A's softmax) runs alone.  The price: the tiles no longer share weight fragments (each fragment pair feeds 3 matrix
// instructions instead of 6: twice the LDS reads).
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

struct PipeTile {                                          // what a tile carries from stage to stage
  f4 ge;
  int d;
  bool ok;
  u4 opA, opB;                                             // V1: the in2 row operands of the two branches
  f4 a[4], a2[4];                                          // M1: first-layer outputs before the ReLU
  u4 xah[2], xal[2], xbh[2], xbl[2];                       // V2: their split pieces
  f4 s[4];                                                 // M2: W_A h_A + W_B h_B + b (feature-centred)
  u4 uh[2], ul[2];                                         // V3
  f4 nrm[4];                                               // M3: W_2 u + b_2 (feature-centred)
  u4 nh[2], nl[2];                                         // V4
  f4 kv[8];                                                // M4
};

// one matrix stage: acc[jo] += W x over the two k-steps.  A step covers TWO accumulators (jo, jo + 1) of one k-step: six matrix
// instructions that alternate between them, so consecutive ones never depend on each other (a chain on one accumulator makes
// the compiler pad with s_nop whenever it renames the destination); after step i the atoms j with floor(j STEPS / K) == i.
// Per accumulator the order of the products is that of linear_acc_x6_n: same bits.
template <int JT_OUT, int K, bool ZERO, class Atoms>
__device__ __forceinline__ void pipe_cluster(f4 (&acc)[JT_OUT], const u4 (&xh)[2], const u4 (&xl)[2], const float* w, int lane, Atoms&& atoms) {
  static_assert(JT_OUT % 2 == 0, "steps pair up the output tiles");
  constexpr int STEPS = JT_OUT;                            // (k-step s, pair p): s-major like linear_acc_x6_n
  constexpr int PAIRS = JT_OUT / 2;
  u4 f1[2][2], f2[2][2];                                   // [buffer][which of the pair]: hi and lo piece fragments
  auto load = [&](int buf, int s, int p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* ptr = w + ((2 * p + h) * 2 + s) * 512 + lane * 4;
      f1[buf][h] = *reinterpret_cast<const u4*>(ptr);
      f2[buf][h] = *reinterpret_cast<const u4*>(ptr + 256);
    }
  };
  load(0, 0, 0);
  static_for<STEPS>([&](auto I) {
    constexpr int i = decltype(I)::value, s = i / PAIRS, p = i % PAIRS, b = i & 1;
    if constexpr (i + 1 < STEPS) load(b ^ 1, (i + 1) / PAIRS, (i + 1) % PAIRS);
    const h8 xhs = __builtin_bit_cast(h8, xh[s]), xls = __builtin_bit_cast(h8, xl[s]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const h8 a1 = __builtin_bit_cast(h8, f1[b][h]);
      if (ZERO && s == 0) acc[2 * p + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xhs, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else acc[2 * p + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xhs, acc[2 * p + h], 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      acc[2 * p + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f1[b][h]), xls, acc[2 * p + h], 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
      acc[2 * p + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f2[b][h]), xhs, acc[2 * p + h], 0, 0, 0);
    static_for<K>([&](auto J) {
      if constexpr (decltype(J)::value * STEPS / K == i) atoms(J);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}
// the two first layers of a tile (layouts.hpp IN2F): 8 single matrix instructions, atoms as above
template <int K, class Atoms>
__device__ __forceinline__ void pipe_cluster_in2(PipeTile& T, const float* fragA, const float* fragB, int lane, Atoms&& atoms) {
  static_for<8>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const float* fr = (i < 4 ? fragA : fragB) + (i & 3) * 256 + lane * 4;
    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(fr));
    const f4 y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, i < 4 ? T.opA : T.opB), f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    if constexpr (i < 4) T.a[i] = y;
    else T.a2[i - 4] = y;
    static_for<K>([&](auto J) {
      if constexpr (decltype(J)::value * 8 / K == i) atoms(J);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int LIST>
__global__ __launch_bounds__(512) void k_edge_attn2p(const float* __restrict__ img_g, const float* __restrict__ geom,
                                                     const int32_t* __restrict__ dst, const float* __restrict__ q, EdgeCount ec, int C_host,
                                                     float* __restrict__ rec, int heads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL6F;
  constexpr int NT = 2;
  const int64_t E = edge_count(ec);
  const int C = stream_len(ec, E, C_host);
  if (E <= 0) return;
  stage_blob(lds, img_g, EL::LDS_SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const StreamMap smap = stream_map(C);
  const int64_t nstreams = stream_count(E, C);
  const int64_t wid = xcd_block() * waves + wave;
  if (wid * (16 * NT) >= nstreams) return;
  SegState S[NT];
  int64_t sid[NT], base_e[NT];
  int cur[NT];
  f4 ng[NT];
  int nd[NT];
  PipeTile T[NT];
  float* qs = lds + EL::LDS_SIZE + wave * (NT * 4 * 64 * 4);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    seg_reset(S[t]);
    sid[t] = wid * (16 * NT) + 16 * t + L.n;
    base_e[t] = stream_base(smap, sid[t]);
    cur[t] = -1;
    const int64_t c = base_e[t] < E ? base_e[t] : E - 1;
    ng[t] = *reinterpret_cast<const f4*>(geom + 4 * c);
    nd[t] = dst[c];
  }
  // ---- the vector stages of tile t, as atoms -------------------------------------------------------------------------------------
  // adv: edge `nit` of the tile's stream enters: its geometry (loaded one edge ahead), the next edge's loads, and -- when the
  // row's target changes -- the record of the finished segment part and the new target's query row (k_edge_attn2)
  const int Cw = stream_length(smap, wid * (16 * NT));     // a wave's 16 NT streams are equally long
  auto adv = [&](auto TT, int nit) {
    constexpr int t = decltype(TT)::value;
    const int64_t e = base_e[t] + nit;
    T[t].ok = nit < Cw && e < E;
    T[t].ge = ng[t];
    T[t].d = nd[t];
    asm volatile("" : "+v"(T[t].ge), "+v"(T[t].d));         // the previous loads are consumed here (see k_edge_attn2)
    const int64_t c = e + 1 < E ? e + 1 : E - 1;
    ng[t] = *reinterpret_cast<const f4*>(geom + 4 * c);
    nd[t] = dst[c];
    if (T[t].ok && T[t].d != cur[t]) {
      if (cur[t] >= 0) seg_flush(S[t], rec, int64_t(cur[t]) + sid[t], L.g);
      seg_reset(S[t]);
      cur[t] = T[t].d;
      const float* qrow = q + int64_t(T[t].d) * D + 4 * L.g;
      const unsigned slot = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>(
          (__attribute__((address_space(3))) float*)(qs + 4 * t * 256))));
      unsigned m0_keep;
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %5\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %3, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %4, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(m0_keep)
                   : "v"(qrow), "v"(qrow + 16), "v"(qrow + 32), "v"(qrow + 48), "s"(slot)
                   : "memory", "scc");
    }
  };
  auto v1 = [&](auto TT, auto J) {                          // 2 atoms: the row operand of branch J
    constexpr int t = decltype(TT)::value, j = decltype(J)::value;
    const float x0 = j == 0 ? T[t].ge[0] : T[t].ge[2], x1 = j == 0 ? T[t].ge[1] : T[t].ge[3];
    const float rstd = in2_rstd(x0, x1, lds + (j == 0 ? EL::A_C : EL::B_C));
    const u4 op = in2_operand(x0 * rstd, x1 * rstd, rstd);
    if constexpr (j == 0) T[t].opA = op;
    else T[t].opB = op;
  };
  auto v2 = [&](auto TT, auto J) {                          // 5 atoms: ReLU + split of (branch, k-step); the bias of the sum
    constexpr int t = decltype(TT)::value, j = decltype(J)::value;
    if constexpr (j < 4) {
      constexpr int br = j >> 1, ks = j & 1;
      f4(&src)[4] = br == 0 ? T[t].a : T[t].a2;
      f4 r0, r1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r0[k] = __int_as_float(max(__float_as_int(src[2 * ks][k]), 0));
        r1[k] = __int_as_float(max(__float_as_int(src[2 * ks + 1][k]), 0));
      }
      if constexpr (br == 0) split_kstep(r0, r1, T[t].xah[ks], T[t].xal[ks]);
      else split_kstep(r0, r1, T[t].xbh[ks], T[t].xbl[ks]);
    } else {
      load_vec<4>(T[t].s, lds + EL::B3, L.g);
    }
  };
  float r3[NT];
  auto v3 = [&](auto TT, auto J) {                          // 4 atoms: rstd | LN + ReLU + split of k-step 0, 1 | the next bias
    constexpr int t = decltype(TT)::value, j = decltype(J)::value;
    if constexpr (j == 0) {
      r3[t] = centred_rstd(T[t].s);
    } else if constexpr (j < 3) {
      constexpr int ks = j - 1;
      f4 u[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int jt = 2 * ks + h;
        const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG0 + 16 * jt + 4 * L.g);
        const f4 be = *reinterpret_cast<const f4*>(lds + EL::AE0 + 16 * jt + 4 * L.g);
#pragma unroll
        for (int c = 0; c < 4; ++c) u[h][c] = fmaxf(fmaf(T[t].s[jt][c] * r3[t], ga[c], be[c]), 0.f);
      }
      split_kstep(u[0], u[1], T[t].uh[ks], T[t].ul[ks]);
    } else {
      load_vec<4>(T[t].nrm, lds + EL::B2, L.g);
    }
  };
  float r4[NT];
  auto v4 = [&](auto TT, auto J) {                          // 3 atoms: rstd | scale + split of k-step 0, 1
    constexpr int t = decltype(TT)::value, j = decltype(J)::value;
    if constexpr (j == 0) {
      r4[t] = centred_rstd(T[t].nrm);
    } else {
      constexpr int ks = j - 1;
      const f4 n0 = T[t].nrm[2 * ks] * r4[t], n1 = T[t].nrm[2 * ks + 1] * r4[t];
      split_kstep(n0, n1, T[t].nh[ks], T[t].nl[ks]);
    }
  };
  f4 lg5[NT];
  auto v5 = [&](auto TT, auto J) {                          // 5 atoms: the four head slots' logits | the softmax update
    constexpr int t = decltype(TT)::value, j = decltype(J)::value;
    if constexpr (j == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the query rows sent to LDS have landed
    if constexpr (j < 4) {
      const f4 qv = *reinterpret_cast<const f4*>(qs + ((4 * t + j) * 64 + L.lane) * 4);
      float p = qv[0] * T[t].kv[j][0];
#pragma unroll
      for (int c = 1; c < 4; ++c) p = fmaf(qv[c], T[t].kv[j][c], p);
      p = xor16_sum(p);
      if (heads == 4) p = xor32_sum(p);
      lg5[t][j] = p * logit_scale(heads);
    } else {
      f4 lg = lg5[t];
      if (!T[t].ok) lg = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      const f4 vv[4] = {T[t].kv[4], T[t].kv[5], T[t].kv[6], T[t].kv[7]};
      seg_update(S[t], lg, vv, f4{1.f, 1.f, 1.f, 1.f});
    }
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  // ---- prologue: both tiles take their first edge; B also runs its first vector stage (it lags A by one stage)
  adv(T0{}, 0);
  adv(T1{}, 0);
  v1(T1{}, std::integral_constant<int, 0>{});
  v1(T1{}, std::integral_constant<int, 1>{});
  for (int it = 0; it < Cw; ++it) {
    keep_lds_reads_here();
    // A.V1 | B.M1
    pipe_cluster_in2<2>(T[1], lds + EL::A_F, lds + EL::B_F, L.lane, [&](auto J) { v1(T0{}, J); });
    // A.M1 | B.V2
    pipe_cluster_in2<5>(T[0], lds + EL::A_F, lds + EL::B_F, L.lane, [&](auto J) { v2(T1{}, J); });
    // A.V2 | B.M2 (W_A then W_B)
    pipe_cluster<4, 3, false>(T[1].s, T[1].xah, T[1].xal, lds + EL::WA3, L.lane, [&](auto J) { v2(T0{}, J); });
    pipe_cluster<4, 2, false>(T[1].s, T[1].xbh, T[1].xbl, lds + EL::WB3, L.lane,
                              [&](auto J) { v2(T0{}, std::integral_constant<int, decltype(J)::value + 3>{}); });
    // A.M2 | B.V3
    pipe_cluster<4, 2, false>(T[0].s, T[0].xah, T[0].xal, lds + EL::WA3, L.lane, [&](auto J) { v3(T1{}, J); });
    pipe_cluster<4, 2, false>(T[0].s, T[0].xbh, T[0].xbl, lds + EL::WB3, L.lane,
                              [&](auto J) { v3(T1{}, std::integral_constant<int, decltype(J)::value + 2>{}); });
    // A.V3 | B.M3
    pipe_cluster<4, 4, false>(T[1].nrm, T[1].uh, T[1].ul, lds + EL::W2, L.lane, [&](auto J) { v3(T0{}, J); });
    // A.M3 | B.V4
    pipe_cluster<4, 3, false>(T[0].nrm, T[0].uh, T[0].ul, lds + EL::W2, L.lane, [&](auto J) { v4(T1{}, J); });
    // A.V4 | B.M4
    pipe_cluster<8, 3, true>(T[1].kv, T[1].nh, T[1].nl, lds + EL::WKV, L.lane, [&](auto J) { v4(T0{}, J); });
    // A.M4 | B.V5, B takes its next edge, B.V1 of that edge
    pipe_cluster<8, 8, true>(T[0].kv, T[0].nh, T[0].nl, lds + EL::WKV, L.lane, [&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j < 5) v5(T1{}, J);
      else if constexpr (j == 5) adv(T1{}, it + 1);
      else v1(T1{}, std::integral_constant<int, j - 6>{});
    });
    // A.V5, A takes its next edge
    static_for<5>([&](auto J) { v5(T0{}, J); });
    adv(T0{}, it + 1);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (cur[t] >= 0) seg_flush(S[t], rec, int64_t(cur[t]) + sid[t], L.g);
}
template __global__ void k_edge_attn2p<0>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int);
template __global__ void k_edge_attn2p<1>(const float*, const float*, const int32_t*, const float*, EdgeCount, int, float*, int);
#endif  // TSDE_PRODUCT
#endif   // TSDE_SPLIT_H3


// records of one target -> agg row.  One wave per target, lane = feature f (jt = f>>4, g = (f>>2)&3); a target's records sit
// at slots target + (first stream .. last stream) of its segment and are combined in that order.
// stats (training path, or null): the merged (max logit, 1 / (sum + 1e-16)) of every (target, head), [R][heads][2]
// img (EdgeL6F): the constant parts the edge kernel left out -- CV joins the aggregate of every non-empty target (add_cv; not
// when the edge kernel ran with attention dropout and carried it itself), q . CK / sqrt(dh) joins the saved maximum, which the
// backward compares with logits that contain it
__global__ __launch_bounds__(256) void k_seg_merge(const int32_t* __restrict__ segptr, const float* __restrict__ rec, EdgeCount ec, int C_host,
                                                   int64_t R, float* __restrict__ agg, float* __restrict__ stats, int heads,
                                                   const float* __restrict__ img, const float* __restrict__ q, int add_cv, int rec_layout) {
  const int C = stream_len(ec, edge_count(ec), C_host);
  const int lane = threadIdx.x & 63;
  const int64_t node = int64_t(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= R) return;
  const int beg = segptr[node], end = segptr[node + 1];
  float out = 0.f, m_out = 0.f, inv_out = 0.f;
  if (end > beg) {
    const StreamMap smap = stream_map(C);
    const int c0 = int(stream_of(smap, beg)), c1 = int(stream_of(smap, end - 1));
    // index of this feature's (m, s) inside a record: by lane group and tile quad (k_edge_attn2), or by 8-feature slot (k_edge_attn3)
    const int ms = rec_layout ? (lane >> 3) : 4 * ((lane >> 2) & 3) + (lane >> 4);
    const float* r = rec + (node + c0) * SEG_REC;
    float m = r[64 + ms], s = r[80 + ms], acc = r[lane];
    for (int c = c0 + 1; c <= c1; ++c) {
      r += SEG_REC;
      const float mp = r[64 + ms], sp = r[80 + ms], ap = r[lane];
      const float mn = fmaxf(m, mp);
      const float a = fast_exp(m - mn), b = fast_exp(mp - mn);
      s = s * a + sp * b;
      acc = acc * a + ap * b;
      m = mn;
    }
    out = acc / (s + 1e-16f);                              // torch_geometric.utils.softmax denominator
    if (add_cv) out += img[EdgeL6F::CV + lane];
    m_out = m;
    inv_out = 1.0f / (s + 1e-16f);
  }
  agg[node * 64 + lane] = out;
  if (stats != nullptr) {                                  // lane = feature: the first lane of every head writes its pair
    const int lph = 64 / heads;
    const float ck = q[node * 64 + lane] * img[EdgeL6F::CK + lane];
    if (end > beg) m_out += (heads == 4 ? head_sum16(ck) * 0.25f : head_sum(ck) * INV_SQRT_DH);
    if ((lane & (lph - 1)) == 0) *reinterpret_cast<float2*>(stats + (node * heads + lane / lph) * 2) = float2{m_out, inv_out};
  }
}

// global interactor: relative-pose embedding only (AGG:42-51), reused by all layers
template <bool X6>
__global__ __launch_bounds__(1024) void k_edge_embed(const float* __restrict__ img_g, const float* __restrict__ geom, EdgeCount ec,
                                                    float* __restrict__ emb_out, int st_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int64_t E = edge_count(ec);
  stage_blob(lds, img_g, X6 ? int(EdgeL6::EMB_SIZE) : int(EdgeL::EMB_SIZE));
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (E + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t e = tile * 16 + L.n, ec = e < E ? e : E - 1;
    const f4 ge = *reinterpret_cast<const f4*>(geom + 4 * ec);
    f4 emb[4];
    edge_embed<X6>(emb, ge, lds, L);
    if (e < E) store_row_st(emb, emb_out, e, L.g, st_bf16 != 0);
  }
}

#if TSDE_SPLIT_H3
// The same embedding with the machinery of the fused edge attention (round 4): two 16-edge tiles per wave share every weight
// fragment, the two Linear(2,64)+LN first layers are one matrix instruction per 16 features, the matrices are feature-centred so a
// LayerNorm is one variance reduction (attn_common.hpp edge_embed_fused_n on the EdgeL6G image); the last LayerNorm's gamma / beta
// are applied here, the rows go out in the state storage type.  Per row the arithmetic is that of the agent-agent kernel's
// embedding, not bit for bit that of k_edge_embed (other summation order inside the LayerNorms): the backward's recomputation
// (EdgeL6 image) differs from either by rounding, as it always did from the fused forward.
__global__ __launch_bounds__(1024) void k_edge_embed2(const float* __restrict__ img_g, const float* __restrict__ geom, EdgeCount ec,
                                                     float* __restrict__ emb_out, int st_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using E = EdgeL6G;
  const int64_t n_edges = edge_count(ec);
  stage_blob(lds, img_g, E::SIZE);
  float* const stg = lds + E::SIZE;                         // [wave][16 rows][68]: the store tiles
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t npairs = (n_edges + 31) / 32;
  for (int64_t pair = int64_t(blockIdx.x) * waves + wave; pair < npairs; pair += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    int64_t e[2];
    f4 ge[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      e[t] = pair * 32 + 16 * t + L.n;
      ge[t] = *reinterpret_cast<const f4*>(geom + 4 * (e[t] < n_edges ? e[t] : n_edges - 1));
    }
    f4 nrm[2][4];
    NoStamps st;
    edge_embed_fused_n<2, NoStamps, E>(nrm, ge, lds, L, st);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f4 ga = *reinterpret_cast<const f4*>(lds + E::AG3 + 16 * jt + 4 * L.g);
        const f4 be = *reinterpret_cast<const f4*>(lds + E::AE3 + 16 * jt + 4 * L.g);
        nrm[t][jt] = nrm[t][jt] * ga + be;
      }
      if (st_bf16 == 2) {
        // split-precision image (gattn_h3.hip): the row as fp16 hi[64] | fp16 lo[64] -- the operand pieces its three readers would
        // otherwise each form again -- through the same LDS tile, whole rows out
        store_tile_rows_split(stg + wave * ROWSTAGE, nrm[t], emb_out, pair * 32 + 16 * t, n_edges, L);
      } else if (st_bf16 != 0) {                             // bf16 rows (128 B): stored from the row-on-lane registers as before
        if (e[t] < n_edges) store_row_st(nrm[t], emb_out, e[t], L.g, true);
      } else {
        // fp32 rows leave as whole rows through the wave's LDS tile (tile.hpp store_tile_rows): they are 535 MB per forward
        store_tile_rows(stg + wave * ROWSTAGE, nrm[t], emb_out, pair * 32 + 16 * t, n_edges, L);
      }
    }
  }
}
#endif

// global layer: k = k_node[src] + lin_k_edge(rel), v = v_node[src] + lin_v_edge(rel)  (AGG:108-117)
// The kernel is latency-bound (ids -> dependent row gathers -> little compute), so it is software-pipelined by
// hand: while tile i is on the matrix cores, the rows of tile i+1 are in flight and the ids of tile i+2 are loading.
struct GEdgeIn {
  f4 r[4], qv[4], kns[4], vns[4];
};
template <bool X6>
__global__ __launch_bounds__(512) void k_global_edge(const float* __restrict__ img_g, const float* __restrict__ rel,
                                                     const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                     const float* __restrict__ q, const float* __restrict__ kn,
                                                     const float* __restrict__ vn, int64_t E, float* __restrict__ logits,
                                                     float* __restrict__ v) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using GL = typename std::conditional<X6, GEdgeL6, GEdgeL>::type;
  stage_blob(lds, img_g, GL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (E + 15) / 16;
  const int64_t stride = int64_t(gridDim.x) * waves;
  auto edge_of = [&](int64_t tile) {
    const int64_t e = tile * 16 + L.n;
    return e < E ? e : E - 1;
  };
  auto issue = [&](GEdgeIn& in, int64_t ec, int s, int d) {
    load_row(in.r, rel, ec, L.g);
    load_row(in.qv, q, d, L.g);
    load_row(in.kns, kn, s, L.g);
    load_row(in.vns, vn, s, L.g);
  };
  auto compute = [&](const GEdgeIn& in, int64_t tile) {
    const int64_t e = tile * 16 + L.n;
    f4 kv[8];
    if constexpr (X6) linear_x6<8, 4>(kv, in.r, lds + GL::WKV, lds + GL::BKV, L);
    else linear<8, 4>(kv, in.r, lds + GL::WKV, lds + GL::BKV, L);
    f4 k[4], vv[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      k[jt] = in.kns[jt] + kv[jt];
      vv[jt] = in.vns[jt] + kv[4 + jt];
    }
    store_logits(in.qv, k, logits, e, e < E, L);
    if (e < E) store_row(vv, v, e, L.g);
  };
  int64_t t0 = int64_t(blockIdx.x) * waves + wave;
  if (t0 >= ntiles) return;
  GEdgeIn A, B;
  int64_t t1 = t0 + stride, t2 = t1 + stride;
  {
    const int64_t e0 = edge_of(t0);
    issue(A, e0, src[e0], dst[e0]);
  }
  int s1 = 0, d1 = 0;
  if (t1 < ntiles) { const int64_t e1 = edge_of(t1); s1 = src[e1]; d1 = dst[e1]; }
  while (true) {
    // ---- A holds tile t0 (in flight or landed), (s1, d1) are the ids of t1
    keep_lds_reads_here();
    int s2 = 0, d2 = 0;
    if (t1 < ntiles) issue(B, edge_of(t1), s1, d1);
    if (t2 < ntiles) { const int64_t e2 = edge_of(t2); s2 = src[e2]; d2 = dst[e2]; }
    compute(A, t0);
    if (t1 >= ntiles) break;
    // ---- roles swapped: B holds tile t1, (s2, d2) are the ids of t2
    keep_lds_reads_here();
    const int64_t t3 = t2 + stride;
    int s3 = 0, d3 = 0;
    if (t2 < ntiles) issue(A, edge_of(t2), s2, d2);
    if (t3 < ntiles) { const int64_t e3 = edge_of(t3); s3 = src[e3]; d3 = dst[e3]; }
    compute(B, t1);
    if (t2 >= ntiles) break;
    t0 = t2; t1 = t3; t2 = t3 + stride;
    s1 = s3; d1 = d3;
  }
}

// ------------------------------------------------------------------------------------------------ segment softmax
// torch_geometric.utils.softmax + add-aggregate: alpha = exp(l - max_seg) / (sum_seg + 1e-16); out = sum alpha * v.
// Single pass over the slab with an online softmax in chunks of 8 edges (8 logits + 8 value rows in flight per
// lane); the running maximum is the segment maximum at the end, so the result equals the two-pass form up to rounding.
__global__ __launch_bounds__(256) void k_seg_softmax_agg(const int32_t* __restrict__ segptr, const float* __restrict__ logits,
                                                         const float* __restrict__ v, int64_t R, float* __restrict__ agg, int heads,
                                                         DropArg drop) {
  const int lane = threadIdx.x & 63;
  const int head = heads == 4 ? lane >> 4 : lane >> 3, slot = heads == 4 ? head : 4 * (head & 1) + (head >> 1);
  // the wave index is uniform: saying so keeps the segment bounds, the edge index and every row address in SGPRs
  const int64_t node = int64_t(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= R) return;
  const int beg = segptr[node], end = segptr[node + 1];
  float m = -INFINITY, s = 0.f, acc = 0.f;
  constexpr int CH = 16;                                   // edges in flight per round trip (a segment's rows are contiguous)
  for (int e0 = beg; e0 < end; e0 += CH) {
    float p[CH], vv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int e = e0 + u < end ? e0 + u : end - 1;
      const float* lrow = logits + int64_t(e) * 8;
      const float* vrow = v + int64_t(e) * 64;
      p[u] = __builtin_nontemporal_load(lrow + slot);       // streamed once: keep them out of the caches' way
      vv[u] = __builtin_nontemporal_load(vrow + lane);
    }
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      p[u] = e0 + u < end ? p[u] : -INFINITY;
      cm = fmaxf(cm, p[u]);
    }
    const float mn = fmaxf(m, cm);
    const float sc = fast_exp(m - mn);
    m = mn;
    s *= sc;
    acc *= sc;
    if (drop.p > 0.f) {                                      // attention dropout: kept edges only, scaled, in the weighted sum
      float kp[CH];
      drop_attn_chunk<CH>(kp, drop, uint32_t(node), uint32_t(e0 - beg), lane, head);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float ex = fast_exp(p[u] - m);
        s += ex;
        acc = fmaf(ex * kp[u], vv[u], acc);
      }
    } else {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float ex = fast_exp(p[u] - m);
        s += ex;
        acc = fmaf(ex, vv[u], acc);
      }
    }
  }
  agg[node * 64 + lane] = end > beg ? acc / (s + 1e-16f) : 0.f;
}

// ------------------------------------------------------------------------------------------------ fused global attention
// One wave per target actor: the whole message/softmax/aggregate of a GlobalInteractorLayer (AGG:101-117) without
// any per-edge GEMM and without materialising per-edge k / v / logits:
//   logit_h(e) = [ q_h . k_node[src]_h  +  (Wke_h^T q_h) . rel_e  +  q_h . bke_h ] / sqrt(8)
//   out        = sum_e alpha_e v_node[src]  +  Wve ( sum_e alpha_{e,h} rel_e )  +  bve * sum_e alpha_e
// i.e. lin_k_edge is folded into the query once per target (U_h = Wke_h^T q_h, 8 x 64) and lin_v_edge is applied
// once per target after the aggregation.  Per edge the wave streams one rel row (256 B) and gathers two node rows.
// Lane l is feature l of the node rows (head l>>3) and holds slice 8*(l&7).. of the rel row / of U for head l>>3.
// NODE = false: the same attention over edge rows alone (k = lin_k(row), v = lin_v(row): the AA / AL encoders' attention on
// stored embedding rows, training path) -- kn / vn are not read.
template <int HEADS, bool ST_BF16, bool DROP, bool NODE>
__global__ __launch_bounds__(256, 4) void k_global_attn(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                     const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                     const float* __restrict__ q, const float* __restrict__ kn,
                                                     const float* __restrict__ vn, int64_t N, float* __restrict__ agg,
                                                     float* __restrict__ stats, DropArg drop) {
  // stats (training tape, or null): the softmax statistics (max logit, 1 / (sum + 1e-16)) of every (target, head), so that
  // the backward does not need a first pass over the segment to rebuild them
  constexpr int LPH = 64 / HEADS;          // lanes (= head dims) per head: 8, or 16 with 4 heads
  constexpr int SL = 64 / LPH;             // rel-row columns per lane: a head's 64-wide row is spread over its LPH lanes
  constexpr int NV = SL / 4;               // ... as float4s
  constexpr float INV = HEADS == 4 ? 0.25f : INV_SQRT_DH;
  __shared__ __attribute__((aligned(16))) float sbuf[4][HEADS][64 + 4];
  __shared__ __attribute__((aligned(16))) float srel[4][8][64];   // a wave's chunk of rel rows (wave-private, no block barrier)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: addresses in SGPRs
  const int h = lane / LPH, j = lane % LPH;
  const int64_t node = xcd_block() * 4 + wv;                 // launched with xcd_grid(): a scene's targets share an L2
  const int64_t nc = node < N ? node : N - 1;
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  // The logits' 1 / sqrt(dh) is folded into the query (ql, U) once per target.  With node rows (the global interactor) the
  // key bias term q_h . bke_h is dropped: it shifts every logit of a (target, head) alike, and the softmax does not see it.
  const float ql = q[nc * 64 + lane] * INV;
  const float cb = NODE ? 0.f : head_sum_n<HEADS>(ql * img[GAttnL::BKE + lane]);
  float U[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) U[e] = 0.f;
#pragma unroll
  for (int d = 0; d < LPH; ++d) {
    const float qd = __shfl(ql, LPH * h + d);
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const f4 w = *reinterpret_cast<const f4*>(wke + (LPH * h + d) * 64 + SL * j + 4 * v4);
#pragma unroll
      for (int e = 0; e < 4; ++e) U[4 * v4 + e] = fmaf(w[e], qd, U[4 * v4 + e]);
    }
  }
  const int beg = segptr[nc], end = node < N ? segptr[nc + 1] : beg;
  // Row loads go through buffer descriptors: address = descriptor base (SGPRs) + scalar row offset + lane * 4, so a load costs no
  // vector instruction (plain pointers made the compiler add a 64-bit per-lane address for each of the 24 loads of a chunk:
  // 11 % of the loop's vector instructions).  Offsets are 32-bit: the rel descriptor starts at the target's segment.
  const __amdgpu_buffer_rsrc_t rs_rel = row_rsrc(rel + (ST_BF16 ? 0 : int64_t(beg) * 64));
  const __amdgpu_buffer_rsrc_t rs_kn = row_rsrc(kn), rs_vn = row_rsrc(vn);
  float m = -INFINITY, s = 0.f, sk = 0.f, accv = 0.f, accr[SL];      // sk: sum of the KEPT, scaled weights (= s without dropout)
#pragma unroll
  for (int e = 0; e < SL; ++e) accr[e] = 0.f;
  constexpr bool dropping = DROP;                             // train-mode instantiation (attention dropout, AGG:116)
  for (int e0 = beg; e0 < end; e0 += 8) {
    f4 r[8][NV];
    float knv[8], vnv[8], lg[8], kp[8];
    if (dropping) drop_attn_chunk<8>(kp, drop, uint32_t(nc), uint32_t(e0 - beg), lane, h);
    // the chunk's 8 source indices in one coalesced load, handed out as scalars: every row address below is an SGPR
    // base plus a per-lane constant offset, so the loop spends no VALU cycles on address arithmetic
    const int sv = NODE ? src[e0 + (lane & 7) < end ? e0 + (lane & 7) : end - 1] : 0;
    // every head needs the whole 64-wide rel row, so a per-lane 32-B slice load would fetch each row HEADS times through the
    // texture-address unit: the wave loads the chunk's 8 rows ONCE, parks them in wave-private LDS and the lanes pick their
    // slices out of LDS (same-address reads across the heads broadcast)
    float rl[8];
    f4 ra, rb;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u < end ? e0 + u : end - 1;
      const int sidx = __builtin_amdgcn_readlane(sv, u);
      if (!ST_BF16) rl[u] = row_load(rs_rel, lane, e - beg);  // fp32 rows: one dword per lane
      knv[u] = NODE ? row_load(rs_kn, lane, sidx) : 0.f;
      vnv[u] = NODE ? row_load(rs_vn, lane, sidx) : 0.f;
    }
    if (ST_BF16) {                                            // bf16 storage: 8 lanes per row, 16 B (8 elements) per lane, widened here
      const int64_t er = e0 + (lane >> 3) < end ? e0 + (lane >> 3) : end - 1;
      const bf4* p = reinterpret_cast<const bf4*>(reinterpret_cast<const __bf16*>(rel) + er * 64 + 8 * (lane & 7));
      ra = widen4(p[0]);
      rb = widen4(p[1]);
    }
    __builtin_amdgcn_wave_barrier();                          // the previous chunk's slice reads are done (same wave, in order)
    if (ST_BF16) {
      *reinterpret_cast<f4*>(&srel[wv][lane >> 3][8 * (lane & 7)]) = ra;
      *reinterpret_cast<f4*>(&srel[wv][lane >> 3][8 * (lane & 7) + 4]) = rb;
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) srel[wv][u][lane] = rl[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4) r[u][v4] = *reinterpret_cast<const f4*>(&srel[wv][u][SL * j + 4 * v4]);
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float p = ql * knv[u];
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
        for (int e = 0; e < 4; ++e) p = fmaf(r[u][v4][e], U[4 * v4 + e], p);
      p = head_sum_n<HEADS>(p);
      if (!NODE) p += cb;
      lg[u] = e0 + u < end ? p : -INFINITY;
      cm = fmaxf(cm, lg[u]);
    }
    const float mn = fmaxf(m, cm);
    const float sc = fast_exp(m - mn);          // m = -inf on the first chunk -> 0
    m = mn;
    s *= sc;
    sk *= sc;
    accv *= sc;
#pragma unroll
    for (int e = 0; e < SL; ++e) accr[e] *= sc;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float ex = fast_exp(lg[u] - m);           // masked lanes: exp(-inf) = 0
      s += ex;
      if (dropping) ex *= kp[u];                // attention dropout (AGG:116): the softmax sum keeps every edge
      sk += ex;
      accv = fmaf(ex, vnv[u], accv);
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
        for (int e = 0; e < 4; ++e) accr[4 * v4 + e] = fmaf(ex, r[u][v4][e], accr[4 * v4 + e]);
    }
  }
  const float inv = 1.0f / (s + 1e-16f);          // PyG softmax denominator
  // S_h = sum_e alpha_{e,h} rel_e, spread over the lanes of head h -> LDS so every lane of the head sees all 64
#pragma unroll
  for (int v4 = 0; v4 < NV; ++v4)
    *reinterpret_cast<f4*>(&sbuf[wv][h][SL * j + 4 * v4]) =
        f4{accr[4 * v4] * inv, accr[4 * v4 + 1] * inv, accr[4 * v4 + 2] * inv, accr[4 * v4 + 3] * inv};
  float out = fmaf(img[GAttnL::BVE + lane], sk * inv, accv * inv);
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
    const f4 wr = *reinterpret_cast<const f4*>(wve + lane * 64 + 4 * k4);
    const f4 sv = *reinterpret_cast<const f4*>(&sbuf[wv][h][4 * k4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
  }
  if (node < N) agg[node * 64 + lane] = out;
  if (stats != nullptr && node < N && j == 0) *reinterpret_cast<float2*>(stats + (node * HEADS + h) * 2) = float2{m, inv};
}
template __global__ void k_global_attn<8, false, false, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<8, false, true, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<8, true, false, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<8, true, true, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, false, false, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, false, true, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, true, false, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, true, true, true>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<8, false, false, false>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<8, false, true, false>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, false, false, false>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);
template __global__ void k_global_attn<4, false, true, false>(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*,
                                                       const float*, int64_t, float*, float*, DropArg);

// ------------------------------------------------------------------------------------------------ update + FFN
// gate = sigmoid(lin_ih(agg) + lin_hh(xn)); upd = agg + gate*(lin_self(xn) - agg); x1 = x + out_proj(upd); xn2 = norm2(x1)
template <bool X6>
__global__ __launch_bounds__(512) void k_node_update(const float* __restrict__ img_g, const float* __restrict__ agg,
                                                     const float* __restrict__ xn, const float* __restrict__ x, int64_t R,
                                                     float* __restrict__ x1, float* __restrict__ xn2, DropArg drop, SegMerge mg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using U = typename std::conditional<X6, UpdL6, UpdL>::type;
  stage_blob(lds, img_g, U::SIZE);
  const Lane L_;
  auto lin = [&](f4 (&out)[4], const f4 (&in)[4], int wo, int bo) {
    if constexpr (X6) linear_x6<4, 4>(out, in, lds + wo, lds + bo, L_);
    else linear<4, 4>(out, in, lds + wo, lds + bo, L_);
  };
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 a[4], n[4], g[4], s[4];
    if (mg.rec != nullptr) {
      // the aggregate straight from the fused edge attention's records (k_seg_merge's arithmetic, in the records' own lane layout:
      // lane (n, g) reads its 16 sums and the (m, s) of its 4 head slots): no agg rows through HBM, one launch less
      const int C = stream_len(mg.ec, edge_count(mg.ec), mg.C_host);
      const int beg = mg.segptr[r], end = mg.segptr[r + 1];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) a[jt] = f4{0.f, 0.f, 0.f, 0.f};
      if (end > beg) {
        const StreamMap smap = stream_map(C);
        const int c0 = int(stream_of(smap, beg)), c1 = int(stream_of(smap, end - 1));
        const float* rr = mg.rec + (r + c0) * SEG_REC;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) a[jt] = *reinterpret_cast<const f4*>(rr + 16 * jt + 4 * L.g);
        f4 m = *reinterpret_cast<const f4*>(rr + 64 + 4 * L.g), sm = *reinterpret_cast<const f4*>(rr + 80 + 4 * L.g);
        for (int c = c0 + 1; c <= c1; ++c) {
          rr += SEG_REC;
          const f4 mp = *reinterpret_cast<const f4*>(rr + 64 + 4 * L.g), sp = *reinterpret_cast<const f4*>(rr + 80 + 4 * L.g);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) {
            const f4 ap = *reinterpret_cast<const f4*>(rr + 16 * jt + 4 * L.g);
            const float mn = fmaxf(m[jt], mp[jt]);
            const float wa = fast_exp(m[jt] - mn), wb = fast_exp(mp[jt] - mn);
            sm[jt] = sm[jt] * wa + sp[jt] * wb;
            a[jt] = a[jt] * wa + ap * wb;
            m[jt] = mn;
          }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const f4 cv = *reinterpret_cast<const f4*>(mg.cv + 16 * jt + 4 * L.g);
#pragma unroll
          for (int c = 0; c < 4; ++c) a[jt][c] = a[jt][c] / (sm[jt] + 1e-16f) + cv[c];      // torch_geometric.utils.softmax denominator
        }
      }
    } else {
      load_row(a, agg, r, L.g);
    }
    load_row(n, xn, r, L.g);
    f4 xr[4];
    load_row(xr, x, r, L.g);                                  // the residual row: requested with the tile's other rows, used last
    range_note(absmax<4>(a), RS_NODE_AGG);
    lin(g, a, U::WIH, U::BIH);
    {
      f4 h[4];
      lin(h, n, U::WHH, U::BHH);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) g[jt] += h[jt];
    }
    sigmoid_<4>(g);
    lin(s, n, U::WSELF, U::BSELF);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) a[jt][c] = a[jt][c] + g[jt][c] * (s[jt][c] - a[jt][c]);
    range_note(absmax<4>(a), RS_NODE_AGG);
    lin(s, a, U::WOUT, U::BOUT);
    if (drop.p > 0.f) {                                       // proj_drop(out_proj(..)) (ENC:611, AGG:132)
      f4 mk[4];
      drop_feat16(mk, drop, DK_PROJ, uint32_t(r), 0, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) s[jt] *= mk[jt];
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) n[jt] = xr[jt] + s[jt];
    if (row < R) store_row(n, x1, row, L.g);
    layer_norm<4>(n, lds + U::N2G, lds + U::N2B, L.g);
    if (row < R) store_row(n, xn2, row, L.g);
  }
}

// out = x1 + mlp.3(relu(mlp.0(xn2)))
__global__ __launch_bounds__(512) void k_ffn(const float* __restrict__ img_g, const float* __restrict__ x1,
                                             const float* __restrict__ xn2, int64_t R, float* __restrict__ out, DropArg drop, int out_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, FfnL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 n[4], hid[16], o[4], xr[4];
    load_row(n, xn2, r, L.g);
    load_row(xr, x1, r, L.g);                                 // the residual row, requested up front
    linear<16, 4>(hid, n, lds + FfnL::W1, lds + FfnL::B1, L);
    relu<16>(hid);
    if (drop.p > 0.f) {                                       // mlp: Linear - ReLU - Dropout - Linear - Dropout (ENC:529-533)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        f4 mk[4];
        drop_feat16(mk, drop, DK_HIDDEN, uint32_t(r), blk, L.g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) hid[4 * blk + jt] *= mk[jt];
      }
    }
    range_note(absmax<16>(hid), RS_FFN_HIDDEN);
    linear<4, 16>(o, hid, lds + FfnL::W2, lds + FfnL::B2, L);
    if (drop.p > 0.f) {
      f4 mk[4];
      drop_feat16(mk, drop, DK_OUT, uint32_t(r), 0, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) o[jt] *= mk[jt];
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) o[jt] += xr[jt];
    if (row < R) store_row_st(o, out, row, L.g, out_bf16 != 0);
  }
}

// split-precision FFN in two passes over this workgroup's tiles, one per half of the 256 hidden units:
//   pass 0: out = x1 + b2 + W2[:, :128] relu(W1[:128] xn2 + b1[:128]);   pass 1: out += W2[:, 128:] relu(W1[128:] xn2 + b1[128:])
__global__ __launch_bounds__(512) void k_ffn6(const float* __restrict__ img_g, const float* __restrict__ x1,
                                              const float* __restrict__ xn2, int64_t R, float* __restrict__ out, DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int hf = 0; hf < 2; ++hf) {
    if (hf) __syncthreads();                                   // everyone is done reading the first half image
    stage_blob(lds, img_g + hf * FfnL6::HALF, FfnL6::HALF);
    for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
      keep_lds_reads_here();
      const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
      f4 n[4], hid[8], o[4];
      load_row(n, xn2, r, L.g);
      linear_x6<8, 4>(hid, n, lds + FfnL6::W1, lds + FfnL6::B1, L);
      relu<8>(hid);
      f4 mo[4];
      if (drop.p > 0.f) {                                      // hidden units 128*hf .. +127 = 64-blocks 2*hf, 2*hf + 1
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          f4 mk[4];
          drop_feat16(mk, drop, DK_HIDDEN, uint32_t(r), 2 * hf + b2, L.g);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) hid[4 * b2 + jt] *= mk[jt];
        }
        drop_feat16(mo, drop, DK_OUT, uint32_t(r), 0, L.g);
      }
      if (hf == 0) {
        load_vec<4>(o, lds + FfnL6::B2, L.g);
        load_row(n, x1, r, L.g);
      } else {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) o[jt] = f4{0.f, 0.f, 0.f, 0.f};
        load_row(n, out, r, L.g);                              // partial sum of pass 0 (same wave wrote it)
      }
      linear_acc_x6<4, 8>(o, hid, lds + FfnL6::W2, L.lane);
      if (drop.p > 0.f) {                                      // the output mask distributes over the two half sums
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) o[jt] *= mo[jt];
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) o[jt] += n[jt];
      if (row < R) store_row(o, out, row, L.g);
    }
  }
}

// xn = norm1(x); NQ stacked projections of xn (AL: q; global layer: q, k_node, v_node)
// SPLITKV (round 6, gattn_h3.hip): the second and third projection (k_node, v_node of a global layer) leave as split-precision rows
// -- fp16 hi[64] | fp16 lo[64], the same 256 bytes -- for the attention kernel that multiplies them on the fp16 matrix cores
template <int NQ, bool SPLITKV>
__global__ __launch_bounds__(512) void k_node_proj(const float* __restrict__ img_g, const float* __restrict__ x, int64_t R,
                                                   float* __restrict__ xn_out, float* __restrict__ p0, float* __restrict__ p1,
                                                   float* __restrict__ p2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using P = NodeProjL<NQ>;
  stage_blob(lds, img_g, P::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 a[4], pr[4 * NQ];
    load_row(a, x, r, L.g);
    layer_norm<4>(a, lds + P::N1G, lds + P::N1B, L.g);
    if (row < R) store_row(a, xn_out, row, L.g);
    linear<4 * NQ, 4>(pr, a, lds + P::W, lds + P::B, L);
    float* outs[3] = {p0, p1, p2};
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      f4 t4[4] = {pr[4 * j], pr[4 * j + 1], pr[4 * j + 2], pr[4 * j + 3]};
#if TSDE_SPLIT_H3
      if (SPLITKV && j > 0) {
        if (row < R) {
          char* o = reinterpret_cast<char*>(outs[j] + row * 64);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) {
            unsigned h0, l0, h1, l1;
            split_pair(t4[jt][0], t4[jt][1], h0, l0);
            split_pair(t4[jt][2], t4[jt][3], h1, l1);
            *reinterpret_cast<uint2*>(o + 32 * jt + 8 * L.g) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(o + 128 + 32 * jt + 8 * L.g) = uint2{l0, l1};
          }
        }
        continue;
      }
#endif
      if (row < R) store_row(t4, outs[j], row, L.g);
    }
  }
}

// global_embed[k] = multihead_proj_k(norm(x))   (AGG:55-57); blockIdx.y = mode
__global__ __launch_bounds__(512) void k_mode_proj(const float* __restrict__ norm_g, const float* __restrict__ proj_g,
                                                   const float* __restrict__ x, int64_t N, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int k = blockIdx.y;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) lds[i] = norm_g[i];
  stage_blob(lds + 128, proj_g + int64_t(k) * (MAT64 + 64), MAT64 + 64);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (N + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < N ? row : N - 1;
    f4 a[4], o[4];
    load_row(a, x, r, L.g);
    layer_norm<4>(a, lds, lds + 64, L.g);
    linear<4, 4>(o, a, lds + 128, lds + 128 + MAT64, L);
    if (row < N) store_row(o, out + int64_t(k) * N * 64, row, L.g);
  }
}

template __global__ void k_edge_kv<false>(const float*, const float*, const int32_t*, const float*, int64_t, float*, float*, int);
template __global__ void k_edge_kv<true>(const float*, const float*, const int32_t*, const float*, int64_t, float*, float*, int);
template __global__ void k_edge_embed<false>(const float*, const float*, EdgeCount, float*, int);
template __global__ void k_edge_embed<true>(const float*, const float*, EdgeCount, float*, int);
template __global__ void k_global_edge<false>(const float*, const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, int64_t, float*, float*);
template __global__ void k_global_edge<true>(const float*, const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, int64_t, float*, float*);
template __global__ void k_node_update<false>(const float*, const float*, const float*, const float*, int64_t, float*, float*, DropArg, SegMerge);
template __global__ void k_node_update<true>(const float*, const float*, const float*, const float*, int64_t, float*, float*, DropArg, SegMerge);
template __global__ void k_node_proj<1, false>(const float*, const float*, int64_t, float*, float*, float*, float*);
template __global__ void k_node_proj<3, false>(const float*, const float*, int64_t, float*, float*, float*, float*);
template __global__ void k_node_proj<3, true>(const float*, const float*, int64_t, float*, float*, float*, float*);

}  // namespace tsde
