// dropout.hpp -- train-mode dropout of the attention blocks, with the masks cut from the in-kernel Philox stream.
//
// The reference's attention blocks (AAEncoder ENC:498-614, ALEncoder ENC:693-797, GlobalInteractorLayer AGG:61-135) apply
// nn.Dropout(p) at four sites each:
//   DK_ATTN    alpha = attn_drop(softmax(...))            per (edge, head)      ENC:592 / ENC:771 / AGG:116
//   DK_PROJ    proj_drop(out_proj(...))                   per (node, feature)   ENC:611 / ENC:794 / AGG:132
//   DK_HIDDEN  mlp: Linear - ReLU - Dropout - ...         per (node, 256 units) ENC:531 / ENC:721 / AGG:88
//   DK_OUT     mlp: ... - Linear - Dropout                per (node, feature)   ENC:533 / ENC:723 / AGG:90
// A mask element is a 16-bit field of a Philox4x32-7 block (philox.hpp PHILOX_ROUNDS), kept when field >= round(p * 65536) and then scaled by 1/(1-p)
// like torch's dropout.  The counter identifies the element, never a launch geometry, so the forward, the forward
// recomputation inside the backward entry points and the backward kernels all regenerate the same mask:
//   node sites   counter = (row, call, stream, 0)  call = 8*blk + 2g + (jt>>1), field 4(jt&1)+c  for feature 64*blk + 16jt + 4g + c
//                (one call serves the 8 features a lane of the row-on-lane layout holds in two of its quads)
//   DK_ATTN      counter = (target node, rank, stream, 0), field = head; rank = position of the edge inside its target's
//                segment of the compacted list (canonical order: ascending sender, duplicates consecutive)
//   stream = STREAM_DROPOUT + 4 * block + site, block: 0 = AAEncoder, 1 = ALEncoder, 2 + i = global layer i.
// Host twin: trajsde_amd/philox.py dropout_feature_mask / dropout_attn_fields (bit-exact: integer comparisons only).
#pragma once
#include "philox.hpp"
#include "tile.hpp"

namespace tsde {

constexpr uint32_t STREAM_DROPOUT = 16;
enum DropKind : int { DK_ATTN = 0, DK_PROJ = 1, DK_HIDDEN = 2, DK_OUT = 3 };

struct DropArg {
  uint64_t seed;
  float p, scale;      // p == 0: off; scale = 1 / (1 - p)
  uint32_t thr;        // keep when the 16-bit field >= thr
  int block;           // attention block id (see above)
};
inline DropArg make_drop(float p, uint64_t seed, int block) {
  DropArg d{seed, 0.f, 1.f, 0u, block};
  if (p > 0.f) {
    d.p = p;
    d.scale = 1.0f / (1.0f - p);
    d.thr = uint32_t(double(p) * 65536.0 + 0.5);
  }
  return d;
}
inline DropArg no_drop() { return DropArg{0, 0.f, 1.f, 0u, 0}; }
__device__ __forceinline__ uint32_t drop_stream(const DropArg& d, int kind) { return STREAM_DROPOUT + 4u * uint32_t(d.block) + uint32_t(kind); }
__device__ __forceinline__ float drop_pick(uint32_t word, int half, const DropArg& d) {
  return ((word >> (16 * half)) & 0xFFFFu) >= d.thr ? d.scale : 0.f;
}

// row-on-lane layout: the 16 factors (0 or 1/(1-p)) of the features 64*blk + 16jt + 4g + c that lane group g holds of `row`
__device__ __forceinline__ void drop_feat16(f4 (&mk)[4], const DropArg& d, int kind, uint32_t row, int blk, int g) {
  const uint32_t st = drop_stream(d, kind);
  uint32_t wa[4], wb[4];
  philox_words(d.seed, st, uint32_t(8 * blk + 2 * g), row, 0u, wa);          // jt = 0, 1
  philox_words(d.seed, st, uint32_t(8 * blk + 2 * g + 1), row, 0u, wb);      // jt = 2, 3
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int idx = 4 * (jt & 1) + c;
      mk[jt][c] = drop_pick(jt < 2 ? wa[idx >> 1] : wb[idx >> 1], idx & 1, d);
    }
}

// row-on-lane edge kernels: factors of the heads that lane group g's features belong to, for the edge at `rank` of target
// `node`: jt -> head 2jt + (g>>1) (8 heads) or head jt (4 heads)
__device__ __forceinline__ f4 drop_attn_row(const DropArg& d, uint32_t node, uint32_t rank, int g, int heads) {
  uint32_t w[4];
  philox_words(d.seed, drop_stream(d, DK_ATTN), rank, node, 0u, w);
  f4 k;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int h = heads == 4 ? jt : 2 * jt + (g >> 1);
    k[jt] = drop_pick(w[h >> 1], h & 1, d);
  }
  return k;
}

// wave-per-target kernels (lane = feature, head = lane / (64 / HEADS)): factors of this lane's head for the CH <= 16 edges
// rank0 .. rank0+CH-1 of `node`.  Lane l draws the block of edge l % 16; the head's word travels inside the 16-lane row.
template <int CH>
__device__ __forceinline__ void drop_attn_chunk(float (&k)[CH], const DropArg& d, uint32_t node, uint32_t rank0, int lane, int head) {
  uint32_t w[4];
  philox_words(d.seed, drop_stream(d, DK_ATTN), rank0 + uint32_t(lane & 15), node, 0u, w);
  const int sel = head >> 1;
  const uint32_t mine = sel == 0 ? w[0] : (sel == 1 ? w[1] : (sel == 2 ? w[2] : w[3]));
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const uint32_t x = uint32_t(__shfl(int(mine), (lane & 48) | u));       // a 16-lane row shares head >> 1, its lane u drew edge u
    k[u] = drop_pick(x, head & 1, d);
  }
}

// TemporalEncoder layers of the vanilla variant (grid.hip; GENC:256-283): block id 16 + layer.  Their node sites (dropout1 = DK_PROJ,
// the FFN's DK_HIDDEN / DK_OUT) use drop_feat16 on the token row n * 22 + s; the attention weights of nn.MultiheadAttention
// (dropout on the softmax output) are per (actor n, head h, query i, key j): block of counter (n, (i * HEADS + h) * 3 + (j >> 3),
// stream, 0), field j & 7.  Host twin: philox.py dropout_temporal_attn_mask.
constexpr int DROP_TEMPORAL_BLOCK0 = 16;    // (global layers take 2 .. 2 + layers - 1)
template <int HEADS>
__device__ __forceinline__ void drop_tr_attn8(float* k /*[8]*/, const DropArg& d, uint32_t n, int head, int i, int chunk) {
  uint32_t w[4];
  philox_words(d.seed, drop_stream(d, DK_ATTN), uint32_t((i * HEADS + head) * 3 + chunk), n, 0u, w);
#pragma unroll
  for (int u = 0; u < 8; ++u) k[u] = drop_pick(w[u >> 1], u & 1, d);
}

}  // namespace tsde
