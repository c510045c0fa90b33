// gattn_h3.hip -- the fused global attention of a GlobalInteractorLayer (reference models/aggregators/agg_hivt.py:92-135; AGG:101-117) on
// the fp16x3 matrix cores, reading relative-pose rows that their PRODUCER already stored as split-precision operand pieces (round 6).
//
// Why.  The global interactor is 28 % of the one-stream forward and its attention (three layers) most of that.  The default kernel
// (gattn_f32.hip) multiplies on the fp32 matrix instruction -- 32 x v_mfma_f32_16x16x4_f32 = 1 024 cycles of the matrix pipe per 16
// edges -- because the fp16x3 form (gattn.hip, round 4) had to split every rel row twice inside the kernel (once per product, in two
// different pairings): 128 values a lane and tile, at what tools/microbench/vissue.hip now shows to be 10 SIMD cycles a value.  The
// rel rows are written once per forward (attn.hip k_edge_embed2) and read by three layers: the split belongs to the writer.
//
// Image.  A rel row is 256 bytes either way: fp32[64], or here  fp16 hi[64] | fp16 lo[64]  with hi = fp16(x) toward zero and lo =
// fp16(x - hi) (tile.hpp split_pair): `TRAJSDE_REL_SPLIT`, inference with fp32 state only (training keeps fp32 rows: its backward
// kernels read them; bf16 state storage keeps bf16 rows).
//
// Kernel.  One wave per target, 16 in-edges a tile, the algebra and the hand-offs of gattn.hip:
//   P1   logits [16 edges x 8 heads]  = [rel_e | k_node[src_e]] (K = 128) . W1         W1 = [U ; Q]: U_h = Wke_h^T q_h, Q = q masked to its head
//   P2   O [8 heads x 128]           += alpha^T [heads x 16 edges] . [rel_e | v_node[src_e]]
// The tile's 16 rel rows are loaded whole (16 B a lane, 16 lanes a row, two tiles ahead) and parked in a wave-private 4 KB LDS tile
// whose 16-byte chunks are XOR-swizzled by the row (chunk c of row r at position c ^ r: every access below is conflict-free):
//   * P1's A operand is the tile read back with an EDGE on a lane -- ds_read_b128 of chunk 8 p + 4 s + g (plane p, k-step s): the eight
//     consecutive features 32 s + 8 g .. of piece p;
//   * P2's B operand contracts over EDGES, the index the rows are not contiguous in: ds_read_b64_tr_b16 hands lane i of a 16-lane group
//     column i (feature 16 cb + i) of the group's four rows (edges 4 g .. 4 g + 3) -- the high pieces in k-slots 0 .. 3, the low pieces
//     in 4 .. 7 of one K = 32 instruction, against the weights' piece repeated in both halves (gattn.hip: all four terms of the split
//     product in two instructions).
// No vector instruction touches a rel value.  The gathered node rows (k_node, v_node: 2 MB a layer, L2-resident) stay fp32 and are split
// here as in gattn.hip.
#include "attn_common.hpp"
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "stamps.hpp"
#include "tile.hpp"

TSDE_STAMP_TABLE(gh3, 8)        // diagnostic builds (tools/phase_stamps.py gh3): phases of one 16-edge tile of k_global_attn_h3

namespace tsde {
#ifndef TSDE_STAMPS
static unsigned long long* const g_stamps_gh3 = nullptr;
#endif

#if TSDE_SPLIT_H3
constexpr float H3_LAZY = 8.0f;              // the running maximum follows a tile's maximum only past this margin (gattn.hip)

// (timing experiments only: -DTSDE_H3_EXP=1 every gathered node row is row (index & 3), =2 every rel row is one of the segment's first 4)
// (=4: a tile's softmax and second product use the previous tile's logits: the dataflow of a software pipeline; =8: no scheduling fences;
//  =16: one workgroup per CU, i.e. one wave per SIMD; =32: W1 fragments kept in registers (results right); =64: no k tile in LDS)
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 8)
#define H3_FENCE() ((void)0)
#else
#define H3_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 1)
#define H3_EXP_NODE(i) ((i) & 3)
#else
#define H3_EXP_NODE(i) (i)
#endif
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 2)
#define H3_EXP_REL(i) ((i) & 3)
#else
#define H3_EXP_REL(i) (i)
#endif
struct H3Rows {                              // what a lane brings for one tile, one kind of row: edges 4g .. 4g+3, 16 bytes of each
  f4 x[4];
};
// (k_global_attn_sc) at most N requests of this wave outstanding; tied to the registers it releases (see h3_wait_vm)
template <int N>
__device__ __forceinline__ void sc_wait_vm(H3Rows& R, int& a, int& b) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(R.x[0]), "+v"(R.x[1]), "+v"(R.x[2]), "+v"(R.x[3]), "+v"(a), "+v"(b) : "n"(N));
}

typedef short s4v __attribute__((ext_vector_type(4)));

// Second form (round 6, after the first one measured on par with the fp32-matrix kernel and the ablations said "latency, not pipes"):
// the same products with the REGISTER budget of three waves per SIMD.  k_node / v_node rows arrive split like the rel rows (their
// writer k_node_proj_split stores fp16 hi | lo), so no vector instruction splits anything but the 16 softmax weights of a tile; every
// row kind takes the same road -- 16 B a lane from memory, ds_write_b128 into a swizzled LDS tile, fragment reads -- and holds registers
// only while in flight: rel two tiles ahead in two sets, k / v one tile ahead in one set each (re-requested as soon as the set has been
// parked).  The k tile and the v tile share one LDS region (k is read by the first product, v written behind it); the per-target
// operand W1 lives in LDS as well (4 KB: columns 8 .. 15 of the logits mirror 0 .. 7, nothing reads them).
#ifndef TSDE_H3_ORDER
#define TSDE_H3_ORDER 0         // 1: k and v tiles of their own, a step's requests in the order they are needed (below: measured, not faster); 0: the shipped order
#endif
constexpr int H3_KV_TILES = TSDE_H3_ORDER ? 2 : 1;
constexpr int H3_WAVE_LDS = 4096 + 4096 * H3_KV_TILES + 4096 + (64 + 64 + 16) * 4;       // rel tile | k tile (| v tile) | W1 | q, node sums, head scalars
// -DTSDE_H3_ASMLD=1: the tile loop's row / index requests are issued from inline assembly and waited for by hand.  Reason (the listing of
// the builtin form): the compiler's own waits in this loop are `s_waitcnt vmcnt(12)` before the rel rows are parked and `vmcnt(0)` before
// the k rows are -- it drains EVERY outstanding request once a tile, so the rel rows "PF tiles ahead" have always arrived one tile
// after they were requested, whatever PF and whatever the order of the requests.  Requests the compiler cannot see are not waited for
// by it; the counts below are those of the request order written in tile_step (listing checked: vmcnt(28) / (8) / (12) a tile at PF = 2,
// 16 for the first tile).  MEASURED (one box, alternating, parity tests green): 158.0 / 156.9 us a layer against 157.3 / 157.9 with the
// compiler's waits; PF = 3: 159.0 / 158.1; PF = 4: 158.7 / 159.5.  So even with the look-ahead really in flight the layer takes what it
// took: the rows were never what a tile waits for.  Kept as a switch (default off) for the record of that experiment -- and UNSAFE as a
// product form: see TSDE_SC_ASMLD (a register the compiler believes loaded may be copied by it before the hand-written wait).
#ifndef TSDE_H3_ASMLD
#define TSDE_H3_ASMLD 0
#endif
typedef int h3_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h3_rsrc h3_rsrc_words(const void* base) {      // raw buffer descriptor in four scalar registers (attn_common.hpp row_rsrc)
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  h3_rsrc r;
  r.x = __builtin_amdgcn_readfirstlane(int(uint32_t(a)));
  r.y = __builtin_amdgcn_readfirstlane(int(uint32_t(a >> 32) & 0xFFFFu));
  r.z = -1;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ f4 h3_asm_load128(h3_rsrc rs, int off) {
  f4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs));
  return v;
}
__device__ __forceinline__ int h3_asm_load32(h3_rsrc rs, int off) {
  int v;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs));
  return v;
}
// at most N requests of this wave still outstanding.  No "memory" clobber (with one the compiler drains every counter around each of
// these statements and keeps the row registers in scratch): the wait is tied to the registers it releases -- their readers cannot be
// moved in front of it -- and volatile statements keep their order among themselves
template <int N>
__device__ __forceinline__ void h3_wait_vm(f4& a, f4& b, f4& c, f4& d) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void h3_wait_vm(int& a, int& b, int& c, int& d) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
#ifndef TSDE_H3_OCC
#define TSDE_H3_OCC 2
#endif
#ifndef TSDE_H3_SKIP
#define TSDE_H3_SKIP 0          // timing experiments (wrong results): 1 no first product, 2 no softmax reductions, 4 no second product, 8 no parking
#endif
__global__ __launch_bounds__(256, TSDE_H3_OCC) void k_global_attn_h3(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                     const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                     const float* __restrict__ q, const float* __restrict__ kn,
                                                                     const float* __restrict__ vn, int64_t N, float* __restrict__ agg,
                                                                     const int64_t* __restrict__ scene_of, const int32_t* __restrict__ scene_ptr,
                                                                     int skip_cap, int skip_flag) {
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 16)              // (timing experiment: one workgroup per CU -- the LDS of a second one is taken)
  __shared__ __attribute__((aligned(16))) char wave_lds[4][H3_WAVE_LDS + 12288];
#else
  __shared__ __attribute__((aligned(16))) char wave_lds[4][H3_WAVE_LDS];
#endif
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nn = lane & 15, g = lane >> 4;
  char* const rt = &wave_lds[wv][0];                         // [16 rows][16 chunks of 16 B]: chunk c of row r at position c ^ r
  char* const kvt = rt + 4096;                               // the same image for the k_node rows, then (first order) for the v_node rows
  char* const vt = rt + 4096 * H3_KV_TILES;                  // the v_node rows' tile (the k tile in the first order)
  char* const w1 = vt + 4096;                                // [k-step 4][piece 2][g 4][head 8][16 B]
  float* const qbuf = reinterpret_cast<float*>(w1 + 4096);
  float* const obuf = qbuf + 64;
  float* const hbuf = obuf + 64;
  float (*const sbuf)[68] = reinterpret_cast<float (*)[68]>(rt);   // the epilogue's S_h rows: over the rel tile, which is free by then
  const int64_t node = xcd_block() * 4 + wv;                 // launched with xcd_grid(): a scene's targets share an L2
  const int64_t nc = node < N ? node : N - 1;
  if (scene_of != nullptr) {                                 // beside k_global_attn_sc: only the targets of scenes too large for its cache
    const int sc = int(scene_of[nc]);
    if (scene_ptr[skip_flag] == 0 && scene_ptr[sc + 1] - scene_ptr[sc] <= skip_cap) return;   // (uniform per wave; flag: k_scene_ptr)
  }
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  // ---- W1 as B operand: lane (head nn & 7, g), step s, slot j = W1[32 s + 8 g + j][head]
  const float ql = q[nc * 64 + lane] * INV_SQRT_DH;          // the logits' 1 / sqrt(dh) rides in the query
  qbuf[lane] = ql;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u4 bh[4], bl[4];                                           // this lane's W1 fragments (what it reads back from `w1` below: lane nn computes those of head nn & 7)
  {
    const int hh = nn & 7;
    const f4 qa = *reinterpret_cast<const f4*>(&qbuf[8 * hh]), qb = *reinterpret_cast<const f4*>(&qbuf[8 * hh + 4]);
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
    split_kstep(w[0][0], w[0][1], bh[0], bl[0]);
    split_kstep(w[1][0], w[1][1], bh[1], bl[1]);
    // k_node part: slot (s, g, j) is node feature d = 32 (s - 2) + 8 g + j, which belongs to head 4 (s - 2) + g
    const f4 z = f4{0.f, 0.f, 0.f, 0.f};
    split_kstep(hh == g ? qa : z, hh == g ? qb : z, bh[2], bl[2]);
    split_kstep(hh == 4 + g ? qa : z, hh == 4 + g ? qb : z, bh[3], bl[3]);
    if (nn < 8) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        *reinterpret_cast<u4*>(w1 + (((2 * s + 0) * 4 + g) * 8 + hh) * 16) = bh[s];
        *reinterpret_cast<u4*>(w1 + (((2 * s + 1) * 4 + g) * 8 + hh) * 16) = bl[s];
      }
    }
  }
  const int w1off = (g * 8 + (nn & 7)) * 16;                 // + (2 s + piece) * 512
  const int beg = segptr[nc], end = node < N ? segptr[nc + 1] : beg;
  f4 O[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) O[c] = f4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, spart = 0.f;
  PhaseClock<8> clk;                                         // (diagnostic builds only: stamps.hpp)
  clk.start();
  unsigned long long units = 0;
  (void)units;

  // (buffer loads: descriptor base in SGPRs + a 32-bit byte offset per lane; the rel descriptor starts at the target's segment, node rows
  //  are addressed from row 0: N < 2^23 is checked on the host)
  const __amdgpu_buffer_rsrc_t rs_rel = row_rsrc(rel + int64_t(beg) * 64), rs_kn = row_rsrc(kn), rs_vn = row_rsrc(vn);
  const __amdgpu_buffer_rsrc_t rs_src = row_rsrc(reinterpret_cast<const float*>(src + beg));
#if TSDE_H3_ASMLD
  const h3_rsrc as_rel = h3_rsrc_words(rel + int64_t(beg) * 64), as_kn = h3_rsrc_words(kn), as_vn = h3_rsrc_words(vn);
  const h3_rsrc as_src = h3_rsrc_words(src + beg);
  auto row4 = [&](__amdgpu_buffer_rsrc_t rs, int row) __attribute__((always_inline)) {
    (void)rs;
    return f4{0.f, 0.f, 0.f, 0.f};
  };
#else
  auto row4 = [&](__amdgpu_buffer_rsrc_t rs, int row) __attribute__((always_inline)) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 256 + 16 * nn, 0, 0));
  };
#endif
  const int deg = end - beg, lim = deg - 1 - 4 * g;
  auto fetch_idx = [&](int (&sidx)[4], int e0) {
    const int o = e0 - beg;                                  // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if TSDE_H3_ASMLD
      sidx[j] = h3_asm_load32(as_src, (min(o + j, lim) + 4 * g) * 4);
#else
      sidx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_src, (min(o + j, lim) + 4 * g) * 4, 0, 0);
#endif
    }
  };
  auto fetch_rel = [&](H3Rows& R, int e0) __attribute__((always_inline)) {                  // chunk nn of the rows 4g .. 4g+3: the stored pieces, as they are
    const int o = e0 - beg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if TSDE_H3_ASMLD
      R.x[j] = h3_asm_load128(as_rel, H3_EXP_REL(min(o + j, lim) + 4 * g) * 256 + 16 * nn);
#else
      R.x[j] = row4(rs_rel, H3_EXP_REL(min(o + j, lim) + 4 * g));
#endif
    }
  };
#if TSDE_H3_ASMLD
  auto fetch_rows = [&](H3Rows& X, h3_rsrc rs, const int (&sidx)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) X.x[j] = h3_asm_load128(rs, H3_EXP_NODE(sidx[j]) * 256 + 16 * nn);
  };
#define rs_kn as_kn
#define rs_vn as_vn
#else
  auto fetch_rows = [&](H3Rows& X, __amdgpu_buffer_rsrc_t rs, const int (&sidx)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) X.x[j] = row4(rs, H3_EXP_NODE(sidx[j]));
  };
#endif
  auto park = [&](char* tile, const H3Rows& X) __attribute__((always_inline)) {             // rows 4g .. 4g+3, chunk nn -> position nn ^ row
#if TSDE_H3_SKIP & 8                                           // (timing only: the rows arrive but are not written to the LDS)
    (void)tile;
    asm volatile("" :: "v"(X.x[0]), "v"(X.x[1]), "v"(X.x[2]), "v"(X.x[3]));
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * g + j;
      *reinterpret_cast<f4*>(tile + r * 256 + 16 * (nn ^ r)) = X.x[j];
    }
#endif
  };
  // per-lane LDS offsets of the fragments (bytes into a tile)
  int a1off[2][2];                                           // A operand of P1: [piece][k-step]: row nn, chunk 8 p + 4 s + g
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) a1off[p][s] = nn * 256 + 16 * ((8 * p + 4 * s + g) ^ nn);
  const int trow = 4 * g + (nn >> 2), tpp = nn & 3;           // B operand of P2: this lane addresses row trow, columns 4 tpp .. of a 16-column block
  auto troff = [&](int piece, int cb) __attribute__((always_inline)) { return trow * 256 + 16 * ((8 * piece + 2 * cb + (tpp >> 1)) ^ trow) + 8 * (tpp & 1); };

#ifndef TSDE_H3_PF
#define TSDE_H3_PF 2
#endif
  constexpr int PF = TSDE_H3_PF;                              // tiles the rel rows travel ahead of their use (register sets)
  H3Rows R[PF], K, V;
  int idx[2][4];
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 4)
  f4 lg_prev = f4{0.f, 0.f, 0.f, 0.f};      // (timing experiment: a tile's softmax uses the PREVIOUS tile's logits -- the dataflow of a software pipeline)
#endif
  auto tile_at = [&](int i) __attribute__((always_inline)) { return beg + 16 * i; };
  auto tile_step = [&](int i, auto U_, auto V_, auto W_) __attribute__((always_inline)) {     // u = i % PF, v = i & 1: compile-time (the register sets must stay registers)
    constexpr int u = decltype(U_)::value, v = decltype(V_)::value;
    // W_: how many requests may still be outstanding when this tile's rel rows are needed.  The request order of a tile is rel (PF ahead),
    // k (next tile) | v (next tile), indices (three ahead) -- 16 a tile -- so in the steady state rel rows requested PF tiles ago have
    // 12 + 16 (PF - 1) younger requests behind them; in the first tiles of a target the prologue's order decides (see the loop below)
    constexpr int wait_rel = decltype(W_)::value;
    (void)wait_rel;
    const int e0 = tile_at(i);
    __builtin_amdgcn_wave_barrier();                          // the previous tile's fragment reads are done (same wave, in order)
    clk.mark(0);                                              // [0] loop overhead
#if TSDE_H3_ASMLD
    h3_wait_vm<wait_rel>(R[u].x[0], R[u].x[1], R[u].x[2], R[u].x[3]);
#endif
    park(rt, R[u]);
#if TSDE_H3_ASMLD
    h3_wait_vm<8>(K.x[0], K.x[1], K.x[2], K.x[3]);            // this tile's k rows: its v rows and the indices two ahead stay in flight
#endif
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 64)
    asm volatile("" :: "v"(K.x[0]), "v"(K.x[1]), "v"(K.x[2]), "v"(K.x[3]));      // (the rows still arrive)
#else
    park(kvt, K);
#endif
#if TSDE_H3_ORDER
    // EXPERIMENT (-DTSDE_H3_ORDER=1; measured, NOT faster).  The memory counter is in order: a wait for a request also waits for
    // everything requested before it.  In the shipped order the next tile's k rows are requested right behind the rel rows of PF tiles
    // ahead and its v rows half a tile later, so the waits for them at the next tile pull the rel look-ahead in.  Here a step requests
    // what is needed FIRST first -- next tile's k rows, its v rows, then the rel rows PF tiles ahead, then the indices three ahead --
    // so every wait leaves the younger requests in flight (costs a v tile of its own).  One box, alternating: 161.2 / 161.3 us a layer
    // against 158.2 / 159.4 shipped; with PF = 3 / 4 on top 160.8-162.0 / 162.1-162.9.  So neither the depth nor the order of the
    // look-ahead bounds this kernel: the hypothesis that the in-order counter is what stops the tile arithmetic from hiding is refuted.
    park(vt, V);
    clk.mark(1);                                              // [1] waiting for this tile's rel / k / v rows, parking them
    fetch_rows(K, rs_kn, idx[v ^ 1]);                         // the next tile's k and v rows (their indices came two steps ago)
    fetch_rows(V, rs_vn, idx[v ^ 1]);
    fetch_rel(R[u], tile_at(i + PF));                         // into the set just parked: PF tiles ahead
    fetch_idx(idx[v ^ 1], tile_at(i + 3));                    // the indices of the tile after the next two
#else
    clk.mark(1);                                              // [1] waiting for this tile's rel / k rows, parking them
    fetch_rel(R[u], tile_at(i + PF));                         // into the set just parked: PF tiles ahead
    fetch_rows(K, rs_kn, idx[v ^ 1]);                         // the next tile's k rows (its indices came two steps ago)
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- P1: the tile's logits, lane (head nn & 7, g): edges 4g .. 4g+3.  Three chains, one per term of the split product (gattn.hip)
    f4 lg;
    {
      f4 t0 = f4{0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
      // (every fragment of the product requested before its first matrix instruction: see k_global_attn_sc)
      u4 fa[4][2], fb[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s) {                           // k-steps 0, 1: the rel half of K; 2, 3: the k_node half
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 64)                 // (timing only: the k_node half reads the rel fragments again -- no k tile traffic)
        const char* tile = rt;
#else
        const char* tile = s < 2 ? rt : kvt;
#endif
        fa[s][0] = *reinterpret_cast<const u4*>(tile + a1off[0][s & 1]);
        fa[s][1] = *reinterpret_cast<const u4*>(tile + a1off[1][s & 1]);
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 32)                 // (W1 from the registers that computed it: 4 KB of LDS reads a tile less)
        fb[s][0] = bh[s];
        fb[s][1] = bl[s];
#else
        fb[s][0] = *reinterpret_cast<const u4*>(w1 + (2 * s + 0) * 512 + w1off);
        fb[s][1] = *reinterpret_cast<const u4*>(w1 + (2 * s + 1) * 512 + w1off);
#endif
      }
      H3_FENCE();
#if TSDE_H3_SKIP & 1                                           // (timing only: no first product -- its fragment reads are dead code too)
      t0 = f4{float(e0), float(i), float(nn), float(g)};
#else
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const h8 ah = __builtin_bit_cast(h8, fa[s][0]), al = __builtin_bit_cast(h8, fa[s][1]);
        const h8 bh = __builtin_bit_cast(h8, fb[s][0]), bl = __builtin_bit_cast(h8, fb[s][1]);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, t2, 0, 0, 0);
      }
#endif
      lg = t0 + (t1 + t2);
#if defined(TSDE_H3_EXP) && (TSDE_H3_EXP & 4)
      { const f4 tmp = lg; lg = lg_prev; lg_prev = tmp; }
#endif
    }
    __builtin_amdgcn_wave_barrier();                          // the k fragments have been read: the region takes the v rows
    clk.mark(2);                                              // [2] P1: fragment reads + 12 matrix instructions
#if !TSDE_H3_ORDER
#if TSDE_H3_ASMLD
    h3_wait_vm<12>(V.x[0], V.x[1], V.x[2], V.x[3]);           // this tile's v rows: younger are the indices, the rel rows PF ahead, the next k rows
#endif
    park(kvt, V);
    fetch_rows(V, rs_vn, idx[v ^ 1]);                         // the next tile's v rows ...
    fetch_idx(idx[v ^ 1], tile_at(i + 3));                    // ... and the indices of the tile after the next two
#endif
    // ---- online softmax of this lane's head over the tile's 16 edges
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e0 + 4 * g + r >= end) lg[r] = -INFINITY;
      cm = fmaxf(cm, lg[r]);
    }
#if !(TSDE_H3_SKIP & 2)                                        // (timing only, bit 2: no cross-lane maximum, no rescaling)
    cm = row_max(cm);                                         // over the four lane groups: the head's 16 edges
#endif
    if (!(TSDE_H3_SKIP & 2) && __builtin_amdgcn_ballot_w64(cm > m + H3_LAZY) != 0ull) {
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
    unsigned eh0, el0, eh1, el1;
    split_pair(ex[0], ex[1], eh0, el0);
    split_pair(ex[2], ex[3], eh1, el1);
    const h8 a2h = __builtin_bit_cast(h8, u4{eh0, eh1, eh0, eh1}), a2l = __builtin_bit_cast(h8, u4{el0, el1, el0, el1});
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    clk.mark(3);                                              // [3] v rows parked, softmax, the weights' split
    // ---- P2 over the tile's 16 edges with K = 32 instructions (gattn.hip): B carries the rows' high pieces in slots 0..3 and their low
    //      pieces in 4..7 -- the transposing read hands lane i column 16 cb + i of the lane group's four rows --, A one piece of the weights
#if TSDE_H3_SKIP & 4                                           // (timing only: no second product, no transposing reads)
#pragma unroll
    for (int c = 0; c < 8; ++c) O[c] += ex * float(c + 1);
    (void)a2h; (void)a2l;
#else
    s4v fh[8], fl[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const char* tile = c < 4 ? rt : vt;
      const int cb = c & 3;
      fh[c] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(tile + troff(0, cb)));
      fl[c] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(tile + troff(1, cb)));
    }
    H3_FENCE();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint2 hw = __builtin_bit_cast(uint2, fh[c]), lw = __builtin_bit_cast(uint2, fl[c]);
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h, __builtin_bit_cast(h8, u4{hw.x, hw.y, lw.x, lw.y}), O[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint2 hw = __builtin_bit_cast(uint2, fh[c]), lw = __builtin_bit_cast(uint2, fl[c]);
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l, __builtin_bit_cast(h8, u4{hw.x, hw.y, lw.x, lw.y}), O[c], 0, 0, 0);
    }
#endif
#ifdef TSDE_STAMPS
    asm volatile("" : "+v"(O[0]), "+v"(O[7]));
    clk.mark(4);                                              // [4] P2: 16 transposing reads, 16 matrix instructions
    ++units;
#endif
  };
  // Every fetch is UNCONDITIONAL (offsets clamp to the segment's last row) so that the compiler's counted waits stay counted.
  if (beg < end) {
    fetch_idx(idx[0], tile_at(0));
    fetch_idx(idx[1], tile_at(1));
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch_rel(R[u], tile_at(u));
#if TSDE_H3_ASMLD
    h3_wait_vm<4 + 4 * PF>(idx[0][0], idx[0][1], idx[0][2], idx[0][3]);      // the first tile's indices: the second tile's and the rel rows stay in flight
#endif
    fetch_rows(K, rs_kn, idx[0]);
    fetch_rows(V, rs_vn, idx[0]);
    fetch_idx(idx[0], tile_at(2));
    constexpr int UN = (PF % 2 == 0) ? PF : 2 * PF;           // a trip of the loop: every set index a compile-time constant
    // one trip = UN tiles.  In the FIRST trip the tiles 0 .. PF - 2 find the prologue's request order behind their rel rows: the later rel
    // sets, k, v and indices of the prologue (4 (PF - 1 - j) + 12) and 16 a tile since; from tile PF - 1 on the steady count holds
    auto trip = [&](int i0, auto FIRST_) __attribute__((always_inline)) {
      bool done = false;
      static_for_<UN>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        constexpr int steady = 12 + 16 * (PF - 1);
        constexpr int w = (decltype(FIRST_)::value && j < PF - 1) ? 4 * (PF - 1 - j) + 12 + 16 * j : steady;
        if (!done) {
          if (tile_at(i0 + j) < end)
            tile_step(i0 + j, std::integral_constant<int, j % PF>{}, std::integral_constant<int, (j & 1)>{}, std::integral_constant<int, w>{});
          else done = true;
        }
      });
      return done;
    };
    if (!trip(0, std::true_type{}))
      for (int i0 = UN; tile_at(i0) < end; i0 += UN)
        if (trip(i0, std::false_type{})) break;
#if TSDE_H3_ASMLD
    // the look-ahead of the last tiles: nothing may land in a register after this
    static_for_<PF>([&](auto U) { h3_wait_vm<0>(R[decltype(U)::value].x[0], R[decltype(U)::value].x[1], R[decltype(U)::value].x[2], R[decltype(U)::value].x[3]); });
    h3_wait_vm<0>(K.x[0], K.x[1], K.x[2], K.x[3]);
    h3_wait_vm<0>(V.x[0], V.x[1], V.x[2], V.x[3]);
    h3_wait_vm<0>(idx[0][0], idx[0][1], idx[0][2], idx[0][3]);
    h3_wait_vm<0>(idx[1][0], idx[1][1], idx[1][2], idx[1][3]);
#endif
  }
  clk.mark(5);
  __builtin_amdgcn_wave_barrier();                            // the last tile's reads of the rel tile are done: sbuf takes its place
  // ---- per target: normalise, lin_v_edge on the aggregated rel rows, store
  const float s = row_sum(spart);                             // lanes (head nn & 7, every g): the head's sum
  const float inv = 1.0f / (s + 1e-16f);                      // PyG softmax denominator
  if (g == 0 && nn < 8) {
    hbuf[nn] = inv;
    hbuf[8 + nn] = s * inv;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (g < 2) {                                                // S_h[16 cb + nn] of the heads 4g + r -> sbuf, normalised
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float iv = hbuf[4 * g + r];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) sbuf[4 * g + r][16 * cb + nn] = O[cb][r] * iv;
    }
  }
  // sum_e alpha v_node of node feature f = 16 cb + nn: head f >> 3 = 2 cb + (nn >> 3), held by lane group (f >> 3) >> 2, register (f >> 3) & 3
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int head = 2 * cb + (nn >> 3);
    if (g == (head >> 2)) {
      const int r = head & 3;
      const float o = r == 0 ? O[4 + cb][0] : (r == 1 ? O[4 + cb][1] : (r == 2 ? O[4 + cb][2] : O[4 + cb][3]));
      obuf[16 * cb + nn] = o * hbuf[head];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int h = lane >> 3;
  float out = fmaf(img[GAttnL::BVE + lane], hbuf[8 + h], obuf[lane]);
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
    const f4 wr = *reinterpret_cast<const f4*>(wve + lane * 64 + 4 * k4);
    const f4 sv = *reinterpret_cast<const f4*>(&sbuf[h][4 * k4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
  }
  if (node < N) agg[node * 64 + lane] = out;
#ifdef TSDE_STAMPS
  clk.mark(6);                                                // [6] the per-target epilogue ([5]: loop exit)
  if (lane == 0) clk.flush(g_stamps_gh3, units);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Third form (round 6): the node rows of a SCENE resident in LDS.  What the two forms above and the fp32-matrix kernel have in common
// is 12 KB through the CU per 16 edges -- 4 KB of rel rows that are genuinely streamed and 8 KB of gathered k_node / v_node rows that the
// 255 other targets of the scene gather as well (every edge of the global graph joins two actors of one scene: AGG:41, the collated
// `edge_index` of complete per-scene graphs) -- through L1, the registers and an LDS tile: ~450 of the ~500 CU cycles a tile takes.
// Here a workgroup owns up to 32 consecutive targets of ONE scene and parks that scene's k_node / v_node rows (split, swizzled: 512 B a
// node, 128 KB for 256 nodes) once; a tile's key fragments and the transposed value fragments are then read straight from the cache at
// the rows its source indices name, and only the rel rows travel.  8 waves, two per SIMD: 160 KB = cache + eight 4 KB rel tiles, the
// per-target scratch (query, head scalars, S rows) aliases a wave's rel tile outside its tile loop, W1 stays in registers.
// Scenes with more than SC_CAP actors are left to k_global_attn_h3 (launched beside this kernel with the complementary guard).
constexpr int SC_CAP = 256, SC_CH = 32, SC_WAVES = 8, SC_CHUNKS = SC_CAP / SC_CH;
#ifndef TSDE_SC_SKIP
#define TSDE_SC_SKIP 0          // timing experiments: 1 no first product, 2 no softmax, 4 no second product, 8 no parking (wrong results)
#endif
constexpr int SC_LDS_BYTES = SC_CAP * 512 + SC_WAVES * 4096;
// -DTSDE_SC_ASMLD=1: the tile loop's row / index requests from inline assembly, waited for by hand (h3_asm_load*, above), because the
// compiler's own wait in this loop is `s_waitcnt vmcnt(0)` at the loop latch -- every request drained once a trip, the rows "two tiles
// ahead" included.  UNSAFE, kept for the record only: 139 against 142 us a layer, but the full GPU suite fails with it (18 tests, results
// that differ run to run).  A value the compiler believes to be in a register the moment the assembly statement has been issued may be
// COPIED by it before the hand-written wait (the listing shows such a move of an index register at the loop head): the copy reads the
// register before the request has landed.  (Longer straight-line trips, the safe way to fewer drains, spill: see the tile loop.)
#ifndef TSDE_SC_ASMLD
#define TSDE_SC_ASMLD 0
#endif
#ifndef TSDE_SC_W1_UNROLL
#define TSDE_SC_W1_UNROLL 4
#endif

// scene_ptr[s] = first node of scene s (batch ids ascending, as collate builds them), scene_ptr[A] = N; scene_ptr[A + 1] (zeroed by the
// launcher) becomes non-zero when the batch is NOT what the scene cache assumes -- scene ids that do not ascend over the nodes, or a
// global edge that joins two scenes (the reference's collated graphs never do: AGG:41, the per-scene complete graphs of the data set).
// The scene-cached kernel then does nothing and the gathering kernel beside it takes every target: a graph the reference would accept
// is never computed wrongly, only more slowly.
__global__ void k_scene_ptr(const int64_t* __restrict__ scene_of, int N, int A, const int32_t* __restrict__ esrc,
                            const int32_t* __restrict__ edst, const int32_t* __restrict__ segptr, int32_t* __restrict__ scene_ptr) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t E = segptr[N];                               // the list's true length (the host may only know a bound: sync-free graphs)
  if (i < E && scene_of[esrc[i]] != scene_of[edst[i]]) scene_ptr[A + 1] = 1;
  if (i > N) return;
  const int64_t cur = i < N ? scene_of[i] : A, prev = i > 0 ? scene_of[i - 1] : -1;
  if (cur < prev || cur > A || prev < -1) {
    scene_ptr[A + 1] = 1;
    return;
  }
  for (int64_t sidx = prev + 1; sidx <= cur; ++sidx) scene_ptr[sidx] = int32_t(i);      // (empty scenes, if any, get empty ranges)
}

__global__ __launch_bounds__(64 * SC_WAVES) void k_global_attn_sc(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                  const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                  const float* __restrict__ q, const float* __restrict__ kn,
                                                                  const float* __restrict__ vn, int64_t N, int A,
                                                                  const int32_t* __restrict__ scene_ptr, float* __restrict__ agg) {
  extern __shared__ __attribute__((aligned(16))) char sc_lds[];
  char* const kc = sc_lds;                                   // [node][16 chunks]: chunk c of node r at position c ^ (r & 15)
  char* const vc = sc_lds + SC_CAP * 256;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nn = lane & 15, g = lane >> 4;
  char* const rt = sc_lds + SC_CAP * 512 + wv * 4096;       // the wave's rel tile; outside the tile loop: its per-target scratch
  float* const qbuf = reinterpret_cast<float*>(rt + 2304);
  float* const obuf = qbuf + 64;
  float* const hbuf = obuf + 64;
  float (*const sbuf)[68] = reinterpret_cast<float (*)[68]>(rt);
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  const int n_units = A * SC_CHUNKS;
  if (scene_ptr[A + 1] != 0) return;                         // (uniform) not a batch of self-contained scenes: k_scene_ptr
  PhaseClock<8> clk;                                         // (diagnostic builds only: stamps.hpp; table shared with k_global_attn_h3)
  clk.start();
  unsigned long long units = 0;
  (void)units;
  for (int unit = int(xcd_block()); unit < n_units; unit += gridDim.x) {
    const int sc = unit / SC_CHUNKS, chunk = unit % SC_CHUNKS;
    const int ps = scene_ptr[sc], pe = scene_ptr[sc + 1], ns = pe - ps;
    if (ns > SC_CAP || chunk * SC_CH >= ns) continue;        // (uniform) left to the gathering kernel / no such chunk
    const int t_end = min(ps + (chunk + 1) * SC_CH, pe);
    // Per-target state of the stream (segment, descriptors) and the register sets of the rows in flight live OUTSIDE the target loop: the
    // first tiles of the NEXT target are requested before the current target's epilogue (open_target), and those of the unit's FIRST
    // target before the scene's rows are copied into the cache, so no target starts cold
    int deg = 0, limR = 0;
#if TSDE_SC_ASMLD
    h3_rsrc rs_rel = h3_rsrc_words(rel), rs_src = h3_rsrc_words(src);
    auto fetch_rel = [&](H3Rows& R, int o) __attribute__((always_inline)) {      // o: the tile's first edge, relative to the segment
#pragma unroll
      for (int j = 0; j < 4; ++j) R.x[j] = h3_asm_load128(rs_rel, (min(o + j, limR) + 4 * g) * 256 + 16 * nn);
    };
    // the cache rows a lane addresses: as an edge-on-a-lane reader (edge nn) and as a transposing reader (edge 4g + (nn >> 2))
    auto fetch_src = [&](int& sa, int& st_, int o) __attribute__((always_inline)) {
      sa = h3_asm_load32(rs_src, min(o + nn, deg - 1) * 4);
      st_ = h3_asm_load32(rs_src, min(o + 4 * g + (nn >> 2), deg - 1) * 4);
    };
#else
    __amdgpu_buffer_rsrc_t rs_rel = row_rsrc(rel), rs_src = row_rsrc(reinterpret_cast<const float*>(src));
    auto fetch_rel = [&](H3Rows& R, int o) {                 // o: the tile's first edge, relative to the segment
#pragma unroll
      for (int j = 0; j < 4; ++j)
        R.x[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_rel, (min(o + j, limR) + 4 * g) * 256 + 16 * nn, 0, 0));
    };
    // the cache rows a lane addresses: as an edge-on-a-lane reader (edge nn) and as a transposing reader (edge 4g + (nn >> 2))
    auto fetch_src = [&](int& sa, int& st_, int o) {
      sa = __builtin_amdgcn_raw_buffer_load_b32(rs_src, min(o + nn, deg - 1) * 4, 0, 0);
      st_ = __builtin_amdgcn_raw_buffer_load_b32(rs_src, min(o + 4 * g + (nn >> 2), deg - 1) * 4, 0, 0);
    };
#endif
    int a1off[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < 2; ++s) a1off[p][s] = nn * 256 + 16 * ((8 * p + 4 * s + g) ^ nn);
    const int trow = 4 * g + (nn >> 2), tpp = nn & 3;
#ifndef TSDE_SC_PF
#define TSDE_SC_PF 2
#endif
    constexpr int SPF = TSDE_SC_PF;                           // tiles the rel rows (and the source indices) travel ahead
    H3Rows R[SPF];
    int sA[SPF], sT[SPF];
    auto open_target = [&](int nd) __attribute__((always_inline)) {      // the target's segment; its first tiles' rows and indices on their way
      const int beg = segptr[nd], end = segptr[nd + 1];
#if TSDE_SC_ASMLD
      rs_rel = h3_rsrc_words(rel + int64_t(beg) * 64);
      rs_src = h3_rsrc_words(src + beg);
#else
      rs_rel = row_rsrc(rel + int64_t(beg) * 64);
      rs_src = row_rsrc(reinterpret_cast<const float*>(src + beg));
#endif
#if defined(TSDE_SC_EXP) && TSDE_SC_EXP == 1                   // (timing experiment: no tiles at all -- what a target costs by itself)
      deg = 0 * (end - beg);
#elif defined(TSDE_SC_EXP) && TSDE_SC_EXP == 2                 // (timing experiment: one tile per target)
      deg = min(end - beg, 16);
#else
      deg = end - beg;
#endif
      limR = deg - 1 - 4 * g;
      if (deg > 0) {
#pragma unroll
        for (int u = 0; u < SPF; ++u) {
          fetch_rel(R[u], 16 * u);
          fetch_src(sA[u], sT[u], 16 * u);
        }
      }
    };
    const int node0 = ps + chunk * SC_CH + wv;
    if (node0 < t_end) open_target(node0);
    __syncthreads();                                         // the previous unit's readers of the cache are done
    {
      // the scene's rows -> the cache: thread i takes 16-byte chunk (i & 15) of row (i >> 4), eight requests in flight
      const f4* ks = reinterpret_cast<const f4*>(kn + int64_t(ps) * 64);
      const f4* vs = reinterpret_cast<const f4*>(vn + int64_t(ps) * 64);
      const int total = ns * 16;
      for (int i0 = threadIdx.x; i0 < total; i0 += 4 * blockDim.x) {
        f4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * blockDim.x;
          if (i < total) { a[u] = ks[i]; b[u] = vs[i]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * blockDim.x;
          if (i < total) {
            const int r = i >> 4, c = i & 15, off = r * 256 + 16 * (c ^ (r & 15));
            *reinterpret_cast<f4*>(kc + off) = a[u];
            *reinterpret_cast<f4*>(vc + off) = b[u];
          }
        }
      }
    }
    __syncthreads();
    for (int node = node0; node < t_end; node += SC_WAVES) {
      __builtin_amdgcn_wave_barrier();
      f4 O[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) O[c] = f4{0.f, 0.f, 0.f, 0.f};
      float m = -INFINITY, spart = 0.f;
      // ---- W1 as B operand: lane (head nn & 7, g), step s, slot j = W1[32 s + 8 g + j][head]; columns 8 .. 15 mirror 0 .. 7 (never read)
      const float ql = q[int64_t(node) * 64 + lane] * INV_SQRT_DH;
      qbuf[lane] = ql;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      u4 b1h[4], b1l[4];
      {
        const int hh = nn & 7;
        const f4 qa = *reinterpret_cast<const f4*>(&qbuf[8 * hh]), qb = *reinterpret_cast<const f4*>(&qbuf[8 * hh + 4]);
        f4 w[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) w[s][0] = w[s][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll TSDE_SC_W1_UNROLL      // (rounds of row loads; the full unroll -- 32 in flight at once -- spilled 106 registers into the tile loop in the first form of this kernel: 183 us a layer against 147)
        for (int d = 0; d < 8; ++d) {
          const float qd = d < 4 ? qa[d & 3] : qb[d & 3];
#if defined(TSDE_SC_EXP) && TSDE_SC_EXP == 3                   // (timing experiment: the per-target weight rows from four cache lines)
          const float* row = wke + (d & 1) * 64 + 8 * (g & 1);
#else
          const float* row = wke + (8 * hh + d) * 64 + 8 * g;
#endif
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            w[s][0] += *reinterpret_cast<const f4*>(row + 32 * s) * qd;
            w[s][1] += *reinterpret_cast<const f4*>(row + 32 * s + 4) * qd;
          }
        }
        split_kstep(w[0][0], w[0][1], b1h[0], b1l[0]);
        split_kstep(w[1][0], w[1][1], b1h[1], b1l[1]);
        const f4 z = f4{0.f, 0.f, 0.f, 0.f};
        split_kstep(hh == g ? qa : z, hh == g ? qb : z, b1h[2], b1l[2]);
        split_kstep(hh == 4 + g ? qa : z, hh == 4 + g ? qb : z, b1h[3], b1l[3]);
      }
      __builtin_amdgcn_wave_barrier();                        // qbuf has been read: the region is the rel tile again
      auto tile_step = [&](int i, auto U_) __attribute__((always_inline)) {
        constexpr int u = decltype(U_)::value;
        const int o = 16 * i;
        __builtin_amdgcn_wave_barrier();                      // the previous tile's fragment reads are done (same wave, in order)
        clk.mark(0);                                          // [0] loop overhead
#if TSDE_SC_ASMLD
        // six requests a tile (four row pieces, two indices), issued SPF tiles ahead in tile order -- by open_target for a target's first
        // SPF tiles, by the step below afterwards: behind this tile's requests there are those of the SPF - 1 tiles after it (and
        // whatever the compiler issued since: its presence only makes this wait stricter, never laxer)
        sc_wait_vm<6 * (SPF - 1)>(R[u], sA[u], sT[u]);
#endif
#ifdef TSDE_STAMPS
        { f4 t0 = R[u].x[0], t1 = R[u].x[3]; int t2 = sA[u], t3 = sT[u]; asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)); }
        clk.mark(1);                                          // [1] waiting for this tile's rel rows / source indices
#endif
        if (TSDE_SC_SKIP & 8) {
          f4 t = R[u].x[0] + R[u].x[1] + R[u].x[2] + R[u].x[3];
          if (t[0] == 1234.5f) *reinterpret_cast<f4*>(rt) = t;
        } else
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          *reinterpret_cast<f4*>(rt + r * 256 + 16 * (nn ^ r)) = R[u].x[j];
        }
        const int ra = sA[u] - ps, rv = sT[u] - ps;           // this tile's cache rows (scene-local)
        fetch_rel(R[u], o + 16 * SPF);
        fetch_src(sA[u], sT[u], o + 16 * SPF);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        clk.mark(2);                                          // [2] parking, requests of the tiles ahead
        // ---- P1: the tile's logits, lane (head nn & 7, g): edges 4g .. 4g+3
        // (ALL fragment reads of a product are requested before its first matrix instruction, and kept there by a scheduling fence:
        //  left to itself the compiler reads a fragment, waits, multiplies, reads the next -- twenty LDS round trips in a row per tile.
        //  Measured: worth 1 % in k_global_attn_h3 (160.4 -> 158.8 us a layer), nothing here (183 -> 186): this kernel keeps W1 in
        //  registers and spills ~106 of them at 256 with or without the fence -- one reason it is the slowest of the three forms.)
        f4 lg;
        {
          f4 t0 = f4{0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
          const char* krow = kc + ra * 256;
          const int rk = ra & 15;
          u4 fa[4][2];
          if (TSDE_SC_SKIP & 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fa[s][0] = fa[s][1] = u4{unsigned(ra), unsigned(o), 0u, 0u};
          } else
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            fa[s][0] = *reinterpret_cast<const u4*>(rt + a1off[0][s]);
            fa[s][1] = *reinterpret_cast<const u4*>(rt + a1off[1][s]);
            fa[2 + s][0] = *reinterpret_cast<const u4*>(krow + 16 * ((4 * s + g) ^ rk));
            fa[2 + s][1] = *reinterpret_cast<const u4*>(krow + 16 * ((8 + 4 * s + g) ^ rk));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (TSDE_SC_SKIP & 1) {
            t0 = __builtin_bit_cast(f4, fa[0][0]) + __builtin_bit_cast(f4, b1h[0]) + __builtin_bit_cast(f4, b1l[3]);
          } else
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const h8 ah = __builtin_bit_cast(h8, fa[s][0]), al = __builtin_bit_cast(h8, fa[s][1]);
            const h8 bh = __builtin_bit_cast(h8, b1h[s]), bl = __builtin_bit_cast(h8, b1l[s]);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, t2, 0, 0, 0);
          }
          lg = t0 + (t1 + t2);
        }
#ifdef TSDE_STAMPS
        asm volatile("" : "+v"(lg));
        clk.mark(3);                                          // [3] P1
#endif
        // ---- online softmax of this lane's head over the tile's 16 edges
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (o + 4 * g + r >= deg) lg[r] = -INFINITY;
          cm = fmaxf(cm, lg[r]);
        }
        if (!(TSDE_SC_SKIP & 2)) cm = row_max(cm);
        if (!(TSDE_SC_SKIP & 2) && __builtin_amdgcn_ballot_w64(cm > m + H3_LAZY) != 0ull) {
          const float mn = fmaxf(m, cm);
          const float sc_ = fast_exp(m - mn);
          m = mn;
          spart *= sc_;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sr = __shfl(sc_, 4 * g + r);
#pragma unroll
            for (int c = 0; c < 8; ++c) O[c][r] *= sr;
          }
        }
        f4 ex;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ex[r] = fast_exp(lg[r] - m);
          spart += ex[r];
        }
        unsigned eh0, el0, eh1, el1;
        split_pair(ex[0], ex[1], eh0, el0);
        split_pair(ex[2], ex[3], eh1, el1);
        const h8 a2h = __builtin_bit_cast(h8, u4{eh0, eh1, eh0, eh1}), a2l = __builtin_bit_cast(h8, u4{el0, el1, el0, el1});
#ifdef TSDE_STAMPS
        asm volatile("" : "+v"(ex));
        clk.mark(4);                                          // [4] softmax
#endif
        // ---- P2: rel columns from the tile, v_node columns from the cache, both by the transposing read
        const char* vrow = vc + rv * 256;
        const int rvk = rv & 15;
        s4v fh[8], fl[8];
        if (TSDE_SC_SKIP & 4) {
#pragma unroll
          for (int c = 0; c < 8; ++c) O[c] += ex * float(rv + c);
        } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int cb = c & 3, idh = 2 * cb + (tpp >> 1);
          const char* ph = c < 4 ? rt + trow * 256 + 16 * (idh ^ trow) : vrow + 16 * (idh ^ rvk);
          const char* pl = c < 4 ? rt + trow * 256 + 16 * ((8 + idh) ^ trow) : vrow + 16 * ((8 + idh) ^ rvk);
          fh[c] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ph + 8 * (tpp & 1)));
          fl[c] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(pl + 8 * (tpp & 1)));
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the two instructions on one accumulator are a chain: all eight first terms, then all eight second ones)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint2 hw = __builtin_bit_cast(uint2, fh[c]), lw = __builtin_bit_cast(uint2, fl[c]);
          O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h, __builtin_bit_cast(h8, u4{hw.x, hw.y, lw.x, lw.y}), O[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint2 hw = __builtin_bit_cast(uint2, fh[c]), lw = __builtin_bit_cast(uint2, fl[c]);
          O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l, __builtin_bit_cast(h8, u4{hw.x, hw.y, lw.x, lw.y}), O[c], 0, 0, 0);
        }
        }
#ifdef TSDE_STAMPS
        asm volatile("" : "+v"(O[0]), "+v"(O[7]));
        clk.mark(5);                                          // [5] P2
        ++units;
#endif
      };
      if (deg > 0) {
        // (Tried: trips of 4 / 8 tiles as straight-line code, so that the compiler's drain at the loop latch -- `s_waitcnt vmcnt(0)`: it does
        //  not carry request counts around a back edge -- comes once per 4 / 8 tiles instead of once per SPF.  Inside such a trip its
        //  waits are exact (vmcnt(13) .. (8)), yet the layer takes 157 / 164 us against 142, with or without scheduling fences between the
        //  tiles; the 20 registers it spills are spilled in the cache fill, not here.)
        for (int i0 = 0; 16 * i0 < deg; i0 += SPF) {
          bool done = false;
          static_for_<SPF>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if (!done) {
              if (16 * (i0 + j) < deg) tile_step(i0 + j, J);
              else done = true;
            }
          });
          if (done) break;
        }
      }
      __builtin_amdgcn_wave_barrier();                        // the last tile's reads are done: the region is scratch again
      const int deg_done = deg;
#if TSDE_SC_ASMLD
      // the clamped look-ahead of the last tiles: nothing of it may land in a register set that is about to be requested into again
#pragma unroll
      for (int u = 0; u < SPF; ++u) sc_wait_vm<0>(R[u], sA[u], sT[u]);
#endif
      if (node + SC_WAVES < t_end) open_target(node + SC_WAVES);      // (the row sets are free: the next target's first tiles leave now)
      // ---- per target: normalise, lin_v_edge on the aggregated rel rows, store (k_global_attn_h3's epilogue)
      const float ssum = row_sum(spart);
      const float inv = 1.0f / (ssum + 1e-16f);
      if (g == 0 && nn < 8) {
        hbuf[nn] = inv;
        hbuf[8 + nn] = ssum * inv;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (g < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float iv = hbuf[4 * g + r];
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) sbuf[4 * g + r][16 * cb + nn] = O[cb][r] * iv;
        }
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int head = 2 * cb + (nn >> 3);
        if (g == (head >> 2)) {
          const int r = head & 3;
          const float o = r == 0 ? O[4 + cb][0] : (r == 1 ? O[4 + cb][1] : (r == 2 ? O[4 + cb][2] : O[4 + cb][3]));
          obuf[16 * cb + nn] = o * hbuf[head];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int h = lane >> 3;
      float out = fmaf(img[GAttnL::BVE + lane], hbuf[8 + h], obuf[lane]);
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
#if defined(TSDE_SC_EXP) && TSDE_SC_EXP == 3
        const f4 wr = *reinterpret_cast<const f4*>(wve + (lane & 1) * 64 + 4 * (k4 & 3));
#else
        const f4 wr = *reinterpret_cast<const f4*>(wve + lane * 64 + 4 * k4);
#endif
        const f4 sv = *reinterpret_cast<const f4*>(&sbuf[h][4 * k4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) out = fmaf(wr[e], sv[e], out);
      }
      agg[int64_t(node) * 64 + lane] = deg_done > 0 ? out : 0.f;
      clk.mark(6);                                            // [6] per-target prologue + epilogue (and the unit's staging)
    }
  }
#ifdef TSDE_STAMPS
  if (lane == 0 && (wv == 0 || wv == 5)) clk.flush(g_stamps_gh3, units);
#endif
}

// OFF by default.  Measured (profiles/r06_ab_runs.md, 32 x 256 agents, rocprofv3 kernel stats, one box):
//   k_global_attn_mf (fp32 matrix instruction, the default)   164-165 us a layer
//   k_global_attn_h3, first form (256 registers)              170.7
//   k_global_attn_h3, this form, two / three waves per SIMD    158.8 / 180 (12 spilled registers)
//   k_global_attn_sc (node rows of a scene resident in LDS)    183-186
// plus 5 us for k_edge_embed2's split epilogue and 3 x 1.7 us for k_node_proj's: 486 against 495 us a forward for the best form, which is
// not worth a second default path.  What the forms have in common, by ablation (-DTSDE_H3_EXP, -DTSDE_SC_SKIP, -DTSDE_SC_EXP;
// tools/microbench/relstream.hip): the rel stream alone, one wave per target at eight waves per CU, needs 100 us (5.3 TB/s); a target's
// own prologue / epilogue (W1 from Wke, the W_ve product: dependent global reads) 23 us a layer; and the tile's work -- first product 46,
// second 18, softmax 7 us -- ADDS to that instead of hiding under it (sc with every phase skipped: 126 us; with all of them: 186).
// Neither a 5x shorter matrix part, nor 128 fewer split values a lane and tile, nor a third wave per SIMD, nor the node rows out of an
// LDS cache (12 KB -> 4 KB through L1 per tile), nor rel rows three or four tiles ahead, nor every fragment read issued before its
// product's first matrix instruction moved the sum.  TRAJSDE_REL_SPLIT=1 selects this kernel, =2 the scene-cached one.
// Default since the end of round 6: 2 (the scene-cached kernel, with the gathering one beside it for scenes beyond the cache); 1 selects
// the gathering kernel alone, 0 the fp32-matrix kernel on fp32 rows (k_global_attn_mf, the default until then).
static int rel_split_mode() {
  static const int v = []() { const char* e = getenv("TRAJSDE_REL_SPLIT"); return e ? atoi(e) : 2; }();
  return v;
}
bool rel_split_enabled() { return rel_split_mode() != 0; }
int launch_global_attn_h3(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, hipStream_t st) {
  TS_LAUNCH_TAG("k_global_attn<8>", false, k_global_attn_h3, xcd_grid(cdiv(N, 4)), 256, 0, st, img, segptr, src, rel, q, kn, vn, N, agg,
                static_cast<const int64_t*>(nullptr), static_cast<const int32_t*>(nullptr), 0, 0);
  return TRAJSDE_OK;
}
// TRAJSDE_REL_SPLIT=2: the scene-cached form, with the gathering form beside it for scenes beyond the cache
bool rel_split_scene_cache() { return rel_split_mode() == 2; }
int launch_scene_ptr(const int64_t* scene_of, int N, int A, const int32_t* esrc, const int32_t* edst, const int32_t* segptr, int64_t E_bound,
                     int32_t* scene_ptr, hipStream_t st) {
  TS_HIP(hipMemsetAsync(scene_ptr + A + 1, 0, sizeof(int32_t), st));
  const int64_t n = E_bound > int64_t(N) + 1 ? E_bound : int64_t(N) + 1;
  TS_LAUNCH(k_scene_ptr, cdiv(n, 256), 256, 0, st, scene_of, N, A, esrc, edst, segptr, scene_ptr);
  return TRAJSDE_OK;
}
int launch_global_attn_sc(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, int A, const int64_t* scene_of, const int32_t* scene_ptr, float* agg, hipStream_t st) {
  const int units = A * SC_CHUNKS;
  TS_LAUNCH_TAG("k_global_attn<8>", false, k_global_attn_sc, xcd_grid(units < 256 ? units : 256), 64 * SC_WAVES, SC_LDS_BYTES, st, img, segptr, src, rel,
                q, kn, vn, N, A, scene_ptr, agg);
  // the targets of scenes larger than the cache (none in the shipped configurations: the launch returns at its first branch)
  TS_LAUNCH_TAG("k_global_attn<8>[big scenes]", false, k_global_attn_h3, xcd_grid(cdiv(N, 4)), 256, 0, st, img, segptr, src, rel, q, kn, vn, N, agg,
                scene_of, scene_ptr, SC_CAP, A + 1);
  return TRAJSDE_OK;
}
#else
bool rel_split_enabled() { return false; }
bool rel_split_scene_cache() { return false; }
int launch_scene_ptr(const int64_t*, int, int, const int32_t*, const int32_t*, const int32_t*, int64_t, int32_t*, hipStream_t) { return fail(TRAJSDE_ERR_UNSUPPORTED, "fp16x3 build only"); }
int launch_global_attn_sc(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*, int64_t, int,
                          const int64_t*, const int32_t*, float*, hipStream_t) {
  return fail(TRAJSDE_ERR_UNSUPPORTED, "the split-image global attention exists in the fp16x3 build only");
}
int launch_global_attn_h3(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*, int64_t, float*,
                          hipStream_t) {
  return fail(TRAJSDE_ERR_UNSUPPORTED, "the split-image global attention exists in the fp16x3 build only");
}
#endif

}  // namespace tsde
