// attn_common.hpp -- device pieces shared by the forward attention kernels (attn.hip) and their backward twins
// (encoder_bwd.hip, aggregator_bwd.hip): the edge embedding, the per-head logit layout, head-wise lane sums.
#pragma once
#include <type_traits>

#include "layouts.hpp"
#include "tile.hpp"

namespace tsde {

constexpr float INV_SQRT_DH = 0.35355339059327373f;   // 1/sqrt(64/8)  (ENC:589-590)

// MultipleInputEmbedding on the two pre-rotated 2-vectors of an edge (EMB:62-70) -> emb
template <bool X6>
__device__ __forceinline__ void edge_embed(f4 (&emb)[4], const f4 geom, const float* lds, const Lane& L) {
  using E = typename std::conditional<X6, EdgeL6, EdgeL>::type;
  f4 h0[4], h1[4], s[4];
  if constexpr (X6) {
#if TSDE_SPLIT_H3 && !defined(TSDE_IN2_VALU)
    // the two first layers on the matrix cores (layouts.hpp IN2F), as in the fused edge attention
    f4 ha[1][4], hb[1][4];
    const float xa[1] = {geom[0]}, xb[1] = {geom[1]}, xc[1] = {geom[2]}, xd[1] = {geom[3]};
    in2_mfma_relu_n<1>(ha, xa, xb, lds + EdgeL6::A_C, lds + EdgeL6::A_F, L.lane);
    in2_mfma_relu_n<1>(hb, xc, xd, lds + EdgeL6::B_C, lds + EdgeL6::B_F, L.lane);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      h0[jt] = ha[0][jt];
      h1[jt] = hb[0][jt];
    }
#else
    in2_ln_relu(h0, geom[0], geom[1], lds + EdgeL6::A_C, lds + E::A_E, L.g);
    in2_ln_relu(h1, geom[2], geom[3], lds + EdgeL6::B_C, lds + E::B_E, L.g);
#endif
  } else {
    linear_in2(h0, geom[0], geom[1], lds + E::A_W0, lds + E::A_B0, L.g);
    layer_norm<4>(h0, lds + E::A_G, lds + E::A_E, L.g);
    relu<4>(h0);
    linear_in2(h1, geom[2], geom[3], lds + E::B_W0, lds + E::B_B0, L.g);
    layer_norm<4>(h1, lds + E::B_G, lds + E::B_E, L.g);
    relu<4>(h1);
  }
  load_vec<4>(s, lds + E::B3, L.g);                       // b0.3 + b1.3
  if constexpr (X6) {
    linear_acc_x6<4, 4>(s, h0, lds + E::WA3, L.lane);
    linear_acc_x6<4, 4>(s, h1, lds + E::WB3, L.lane);
  } else {
    linear_acc<4, 4>(s, h0, lds + E::WA3, L.lane);
    linear_acc<4, 4>(s, h1, lds + E::WB3, L.lane);        // sum of the two branches = one K=128 contraction
  }
  layer_norm<4>(s, lds + E::AG0, lds + E::AE0, L.g);
  relu<4>(s);
  if constexpr (X6) linear_x6<4, 4>(emb, s, lds + E::W2, lds + E::B2, L);
  else linear<4, 4>(emb, s, lds + E::W2, lds + E::B2, L);
  layer_norm<4>(emb, lds + E::AG3, lds + E::AE3, L.g);
}

// the split-precision embedding for two edge tiles of one wave at once (linear_acc_x6_2: weight fragments read once)
__device__ __forceinline__ void edge_embed2_x6(f4 (&emb0)[4], f4 (&emb1)[4], const f4 ge0, const f4 ge1, const float* lds,
                                               const Lane& L) {
  using E = EdgeL6;
  f4 a0[4], a1[4], b0[4], b1[4], s0[4], s1[4];
#if TSDE_SPLIT_H3 && !defined(TSDE_IN2_VALU)
  {                                                        // the first layers on the matrix cores, like edge_embed<true>: same bits
    f4 h[2][4];
    const float xa[2] = {ge0[0], ge1[0]}, xb[2] = {ge0[1], ge1[1]}, xc[2] = {ge0[2], ge1[2]}, xd[2] = {ge0[3], ge1[3]};
    in2_mfma_relu_n<2>(h, xa, xb, lds + E::A_C, lds + E::A_F, L.lane);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { a0[jt] = h[0][jt]; a1[jt] = h[1][jt]; }
    in2_mfma_relu_n<2>(h, xc, xd, lds + E::B_C, lds + E::B_F, L.lane);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { b0[jt] = h[0][jt]; b1[jt] = h[1][jt]; }
  }
  load_vec<4>(s0, lds + E::B3, L.g);
  load_vec<4>(s1, lds + E::B3, L.g);
  linear_acc_x6_2<4, 4>(s0, s1, a0, a1, lds + E::WA3, L.lane);
  linear_acc_x6_2<4, 4>(s0, s1, b0, b1, lds + E::WB3, L.lane);
#else
  in2_ln_relu(a0, ge0[0], ge0[1], lds + E::A_C, lds + E::A_E, L.g);
  in2_ln_relu(a1, ge1[0], ge1[1], lds + E::A_C, lds + E::A_E, L.g);
  load_vec<4>(s0, lds + E::B3, L.g);
  load_vec<4>(s1, lds + E::B3, L.g);
  linear_acc_x6_2<4, 4>(s0, s1, a0, a1, lds + E::WA3, L.lane);
  in2_ln_relu(b0, ge0[2], ge0[3], lds + E::B_C, lds + E::B_E, L.g);
  in2_ln_relu(b1, ge1[2], ge1[3], lds + E::B_C, lds + E::B_E, L.g);
  linear_acc_x6_2<4, 4>(s0, s1, b0, b1, lds + E::WB3, L.lane);
#endif
  layer_norm<4>(s0, lds + E::AG0, lds + E::AE0, L.g);
  relu<4>(s0);
  layer_norm<4>(s1, lds + E::AG0, lds + E::AE0, L.g);
  relu<4>(s1);
  load_vec<4>(emb0, lds + E::B2, L.g);
  load_vec<4>(emb1, lds + E::B2, L.g);
  linear_acc_x6_2<4, 4>(emb0, emb1, s0, s1, lds + E::W2, L.lane);
  layer_norm<4>(emb0, lds + E::AG3, lds + E::AE3, L.g);
  layer_norm<4>(emb1, lds + E::AG3, lds + E::AE3, L.g);
}

// The embedding for the fused edge-attention kernel, on its own image (layouts.hpp EdgeL6F), for the NT row tiles of one wave: the
// two matrix layers hand out feature-centred rows, so each LayerNorm is one variance reduction, and the last LayerNorm stops at
// (y - mean) * rstd -- its gamma sits in the lin_k | lin_v image, its beta in the per-target constants.  nrm: those normalised rows.
// ST: phase stamps of a diagnostic build (attn.hip, TSDE_EDGE_STAMPS); NoStamps compiles to nothing
struct NoStamps {
  __device__ __forceinline__ void mark(int) {}
};
template <int NT, class ST, class E = EdgeL6F>
__device__ __forceinline__ void edge_embed_fused_n(f4 (&nrm)[NT][4], const f4 (&ge)[NT], const float* lds, const Lane& L, ST& st) {
  f4 a[NT][4], s[NT][4];
#if TSDE_SPLIT_H3 && !defined(TSDE_IN2_VALU)
  // the two first layers on the matrix cores (layouts.hpp IN2F): 4 matrix instructions per tile and branch instead of 48 fma
  float xa[NT], xb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    xa[t] = ge[t][0];
    xb[t] = ge[t][1];
    load_vec<4>(s[t], lds + E::B3, L.g);
  }
  in2_mfma_relu_n<NT>(a, xa, xb, lds + E::A_C, lds + E::A_F, L.lane);
  st.mark(1);
  linear_acc_x6_n<NT, 4, 4>(s, a, lds + E::WA3, L.lane);
  st.mark(2);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    xa[t] = ge[t][2];
    xb[t] = ge[t][3];
  }
  in2_mfma_relu_n<NT>(a, xa, xb, lds + E::B_C, lds + E::B_F, L.lane);
  st.mark(3);
  linear_acc_x6_n<NT, 4, 4>(s, a, lds + E::WB3, L.lane);
  st.mark(4);
#else
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    in2_ln_relu(a[t], ge[t][0], ge[t][1], lds + E::A_C, lds + E::A_E, L.g);
    load_vec<4>(s[t], lds + E::B3, L.g);
  }
  linear_acc_x6_n<NT, 4, 4>(s, a, lds + E::WA3, L.lane);
#pragma unroll
  for (int t = 0; t < NT; ++t) in2_ln_relu(a[t], ge[t][2], ge[t][3], lds + E::B_C, lds + E::B_E, L.g);
  linear_acc_x6_n<NT, 4, 4>(s, a, lds + E::WB3, L.lane);
#endif
  float r[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) r[t] = centred_rstd(s[t]);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const f4 ga = *reinterpret_cast<const f4*>(lds + E::AG0 + 16 * jt + 4 * L.g);
    const f4 be = *reinterpret_cast<const f4*>(lds + E::AE0 + 16 * jt + 4 * L.g);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) s[t][jt][c] = fmaxf(fmaf(s[t][jt][c] * r[t], ga[c], be[c]), 0.f);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) load_vec<4>(nrm[t], lds + E::B2, L.g);
  st.mark(5);
  linear_acc_x6_n<NT, 4, 4>(nrm, s, lds + E::W2, L.lane);
  st.mark(6);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float q = centred_rstd(nrm[t]);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) nrm[t][jt] *= q;
  }
  st.mark(7);
}
template <int NT>
__device__ __forceinline__ void edge_embed_fused_n(f4 (&nrm)[NT][4], const f4 (&ge)[NT], const float* lds, const Lane& L) {
  NoStamps st;
  edge_embed_fused_n<NT, NoStamps>(nrm, ge, lds, L, st);
}

// per-head logits of 16 edges: q.k over the dims of each head / sqrt(dh).
// 8 heads (dh = 8): heads sit pairwise on lane groups (g, g^1); stored as logits[e][slot], slot = 4*(head&1) + (head>>1),
//                   one 16-B store per lane pair.
// 4 heads (dh = 16): head jt spans all four lane groups; slots 0..3 = heads 0..3 (slots 4..7 unused).
__device__ __forceinline__ void store_logits(const f4 (&qv)[4], const f4 (&k)[4], float* __restrict__ logits, int64_t e,
                                             bool valid, const Lane& L, int heads = 8) {
  f4 lg;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float p = qv[jt][0] * k[jt][0];
#pragma unroll
    for (int c = 1; c < 4; ++c) p = fmaf(qv[jt][c], k[jt][c], p);
    p = xor16_sum(p);
    if (heads == 4) p = xor32_sum(p);
    lg[jt] = p * (heads == 4 ? 0.25f : INV_SQRT_DH);
  }
  if (heads == 4) {
    if (valid && L.g == 0) *reinterpret_cast<f4*>(logits + e * 8) = lg;
  } else if (valid && (L.g & 1) == 0) {
    *reinterpret_cast<f4*>(logits + e * 8 + 4 * (L.g >> 1)) = lg;
  }
}

// [rows][64] fp32 rows read with lane = feature as  descriptor base (SGPRs) + row * 256 (a scalar) + lane * 4: buffer loads that
// cost no vector instruction, where a plain pointer makes the compiler form a 64-bit per-lane address for every load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFFF, 0x00020000);      // raw buffer, 32-bit data format
}
__device__ __forceinline__ float row_load(__amdgpu_buffer_rsrc_t rs, int lane, int row /*uniform, < 2^23*/) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, row * 256, 0));
}

// sums over the 8 lanes that hold one head when lane = feature (segment / fused-attention kernels)
__device__ __forceinline__ float dpp_add(float v, int ctrl_tag) {
  // ctrl_tag 0: xor 1, 1: xor 2 (quad permutes), 2: mirror within 8 lanes
  int r;
  if (ctrl_tag == 0) r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);
  else if (ctrl_tag == 1) r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);
  else r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float head_sum(float v) { return dpp_add(dpp_add(dpp_add(v, 0), 1), 2); }   // over the 8 lanes of a head
__device__ __forceinline__ float head_sum16(float v) {                                                  // over the 16 lanes of a head (4 heads)
  v = head_sum(v);
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true);              // row_mirror: joins the two halves
  return v + __int_as_float(r);
}
template <int HEADS>
__device__ __forceinline__ float head_sum_n(float v) { return HEADS == 4 ? head_sum16(v) : head_sum(v); }


}  // namespace tsde
