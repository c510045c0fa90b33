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
typedef short s4v __attribute__((ext_vector_type(4)));

// Second form (round 6, after the first one measured on par with the fp32-matrix kernel and the ablations said "latency, not pipes"):
// the same products with the REGISTER budget of three waves per SIMD.  k_node / v_node rows arrive split like the rel rows (their
// writer k_node_proj_split stores fp16 hi | lo), so no vector instruction splits anything but the 16 softmax weights of a tile; every
// row kind takes the same road -- 16 B a lane from memory, ds_write_b128 into a swizzled LDS tile, fragment reads -- and holds registers
// only while in flight: rel two tiles ahead in two sets, k / v one tile ahead in one set each (re-requested as soon as the set has been
// parked).  The k tile and the v tile share one LDS region (k is read by the first product, v written behind it); the per-target
// operand W1 lives in LDS as well (4 KB: columns 8 .. 15 of the logits mirror 0 .. 7, nothing reads them).
constexpr int H3_WAVE_LDS = 4096 + 4096 + 4096 + (64 + 64 + 16) * 4;       // rel tile | k / v tile | W1 | q, node sums, head scalars
#ifndef TSDE_H3_OCC
#define TSDE_H3_OCC 2
#endif
__global__ __launch_bounds__(256, TSDE_H3_OCC) void k_global_attn_h3(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                     const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                     const float* __restrict__ q, const float* __restrict__ kn,
                                                                     const float* __restrict__ vn, int64_t N, float* __restrict__ agg) {
  __shared__ __attribute__((aligned(16))) char wave_lds[4][H3_WAVE_LDS];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nn = lane & 15, g = lane >> 4;
  char* const rt = &wave_lds[wv][0];                         // [16 rows][16 chunks of 16 B]: chunk c of row r at position c ^ r
  char* const kvt = rt + 4096;                               // the same image for the k_node rows, then for the v_node rows
  char* const w1 = rt + 8192;                                // [k-step 4][piece 2][g 4][head 8][16 B]
  float* const qbuf = reinterpret_cast<float*>(rt + 12288);
  float* const obuf = qbuf + 64;
  float* const hbuf = obuf + 64;
  float (*const sbuf)[68] = reinterpret_cast<float (*)[68]>(rt);   // the epilogue's S_h rows: over the rel tile, which is free by then
  const int64_t node = xcd_block() * 4 + wv;                 // launched with xcd_grid(): a scene's targets share an L2
  const int64_t nc = node < N ? node : N - 1;
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  // ---- W1 as B operand: lane (head nn & 7, g), step s, slot j = W1[32 s + 8 g + j][head]
  const float ql = q[nc * 64 + lane] * INV_SQRT_DH;          // the logits' 1 / sqrt(dh) rides in the query
  qbuf[lane] = ql;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
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
    u4 bh[4], bl[4];
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
  auto row4 = [&](__amdgpu_buffer_rsrc_t rs, int row) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, row * 256 + 16 * nn, 0, 0));
  };
  const int deg = end - beg, lim = deg - 1 - 4 * g;
  auto fetch_idx = [&](int (&sidx)[4], int e0) {
    const int o = e0 - beg;                                  // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) sidx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_src, (min(o + j, lim) + 4 * g) * 4, 0, 0);
  };
  auto fetch_rel = [&](H3Rows& R, int e0) {                  // chunk nn of the rows 4g .. 4g+3: the stored pieces, as they are
    const int o = e0 - beg;
#pragma unroll
    for (int j = 0; j < 4; ++j) R.x[j] = row4(rs_rel, H3_EXP_REL(min(o + j, lim) + 4 * g));
  };
  auto fetch_rows = [&](H3Rows& X, __amdgpu_buffer_rsrc_t rs, const int (&sidx)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) X.x[j] = row4(rs, H3_EXP_NODE(sidx[j]));
  };
  auto park = [&](char* tile, const H3Rows& X) {             // rows 4g .. 4g+3, chunk nn -> position nn ^ row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * g + j;
      *reinterpret_cast<f4*>(tile + r * 256 + 16 * (nn ^ r)) = X.x[j];
    }
  };
  // per-lane LDS offsets of the fragments (bytes into a tile)
  int a1off[2][2];                                           // A operand of P1: [piece][k-step]: row nn, chunk 8 p + 4 s + g
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) a1off[p][s] = nn * 256 + 16 * ((8 * p + 4 * s + g) ^ nn);
  const int trow = 4 * g + (nn >> 2), tpp = nn & 3;           // B operand of P2: this lane addresses row trow, columns 4 tpp .. of a 16-column block
  auto troff = [&](int piece, int cb) { return trow * 256 + 16 * ((8 * piece + 2 * cb + (tpp >> 1)) ^ trow) + 8 * (tpp & 1); };

  H3Rows R[2], K, V;
  int idx[2][4];
  auto tile_at = [&](int i) { return beg + 16 * i; };
  auto tile_step = [&](int i, int u) {                        // u = i & 1 (a constant after unrolling)
    const int e0 = tile_at(i);
    __builtin_amdgcn_wave_barrier();                          // the previous tile's fragment reads are done (same wave, in order)
    clk.mark(0);                                              // [0] loop overhead
    park(rt, R[u]);
    park(kvt, K);
    clk.mark(1);                                              // [1] waiting for this tile's rel / k rows, parking them
    fetch_rel(R[u], tile_at(i + 2));                          // into the set just parked: two tiles ahead
    fetch_rows(K, rs_kn, idx[u ^ 1]);                         // the next tile's k rows (its indices came two steps ago)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- P1: the tile's logits, lane (head nn & 7, g): edges 4g .. 4g+3.  Three chains, one per term of the split product (gattn.hip)
    f4 lg;
    {
      f4 t0 = f4{0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {                           // k-steps 0, 1: the rel half of K; 2, 3: the k_node half
        const char* tile = s < 2 ? rt : kvt;
        const h8 ah = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(tile + a1off[0][s & 1]));
        const h8 al = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(tile + a1off[1][s & 1]));
        const h8 bh = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(w1 + (2 * s + 0) * 512 + w1off));
        const h8 bl = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(w1 + (2 * s + 1) * 512 + w1off));
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, t2, 0, 0, 0);
      }
      lg = t0 + (t1 + t2);
    }
    __builtin_amdgcn_wave_barrier();                          // the k fragments have been read: the region takes the v rows
    clk.mark(2);                                              // [2] P1: fragment reads + 12 matrix instructions
    park(kvt, V);
    fetch_rows(V, rs_vn, idx[u ^ 1]);                         // the next tile's v rows ...
    fetch_idx(idx[u ^ 1], tile_at(i + 3));                    // ... and the indices of the tile after the next two
    // ---- online softmax of this lane's head over the tile's 16 edges
    float cm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e0 + 4 * g + r >= end) lg[r] = -INFINITY;
      cm = fmaxf(cm, lg[r]);
    }
    cm = row_max(cm);                                         // over the four lane groups: the head's 16 edges
    if (__builtin_amdgcn_ballot_w64(cm > m + H3_LAZY) != 0ull) {
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
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const char* tile = c < 4 ? rt : kvt;
      const int cb = c & 3;
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(tile + troff(0, cb)));
      const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(tile + troff(1, cb)));
      const uint2 hw = __builtin_bit_cast(uint2, hi), lw = __builtin_bit_cast(uint2, lo);
      const h8 br = __builtin_bit_cast(h8, u4{hw.x, hw.y, lw.x, lw.y});
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h, br, O[c], 0, 0, 0);
      O[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l, br, O[c], 0, 0, 0);
    }
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
    fetch_rel(R[0], tile_at(0));
    fetch_rel(R[1], tile_at(1));
    fetch_rows(K, rs_kn, idx[0]);
    fetch_rows(V, rs_vn, idx[0]);
    fetch_idx(idx[0], tile_at(2));
    for (int i0 = 0; tile_at(i0) < end; i0 += 2) {
      tile_step(i0, 0);
      if (tile_at(i0 + 1) < end) tile_step(i0 + 1, 1);
      else break;
    }
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

// OFF by default -- measured (profiles/r06_ab_runs.md, 32 x 256 agents, one box): 170.7 us a layer against 168.2 us for the fp32-matrix
// kernel (gattn_f32.hip) plus 5 us for the writer's split epilogue.  With a matrix part 5x shorter (28 x 16 cycles instead of 32 x 32)
// and no vector instruction on a rel value the time does not move: neither pipe bounds this kernel.  What does, by ablation
// (-DTSDE_H3_EXP): every row served from the cache 128 us (memory: 26 %); the rest is the tile's dependent chain (stage -> product ->
// softmax -> product, ~1 100 cycles of issue in ~4 000) at the two waves per SIMD that 256 registers allow.  TRAJSDE_REL_SPLIT=1 selects it.
bool rel_split_enabled() {
  static const bool v = []() { const char* e = getenv("TRAJSDE_REL_SPLIT"); return e && atoi(e) != 0; }();
  return v;
}
int launch_global_attn_h3(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, hipStream_t st) {
  TS_LAUNCH_TAG("k_global_attn<8>", false, k_global_attn_h3, xcd_grid(cdiv(N, 4)), 256, 0, st, img, segptr, src, rel, q, kn, vn, N, agg);
  return TRAJSDE_OK;
}
#else
bool rel_split_enabled() { return false; }
int launch_global_attn_h3(const float*, const int32_t*, const int32_t*, const float*, const float*, const float*, const float*, int64_t, float*,
                          hipStream_t) {
  return fail(TRAJSDE_ERR_UNSUPPORTED, "the split-image global attention exists in the fp16x3 build only");
}
#endif

}  // namespace tsde
