// aggregator_bwd.hip -- backward of the GlobalInteractor stage (reference models/aggregators/agg_hivt.py:38-58,
// 92-135): given dL/d global_embed [K,N,64] it returns dL/d local_embed and the gradients of every aggregator
// parameter.  Second family of SURVEY.md 8(f) rank 1.
//
// The stage's forward is recomputed here with one buffer per layer (the "tape"), then walked backwards:
//   multihead_proj / norm            k_lin_t_acc x K  ->  k_node_proj_bwd<0>
//   per layer, last to first         node_block_backward (FFN, gated update)  ->  k_gattn_bwd  ->  k_node_proj_bwd<3>
//   rel_embed                        edge_embed_backward on the summed d rel rows
//
// k_gattn_bwd mirrors the fused forward (attn.hip k_global_attn): one wave per target actor, lin_k_edge folded into
// the query (U_h = Wke_h^T q_h), lin_v_edge folded into the incoming gradient (Z_h = Wve_h^T dagg_h), softmax
// statistics taken from the forward's tape (one pass over the segment).  With alpha the attention weights,
//   d alpha_e,h = dagg_h . (v_node[src] + lin_v_edge(rel_e))_h,   d logit_e,h = alpha (d alpha - dagg_h . agg_h)
// Per-target sums  RL_h = sum_e dlogit/sqrt(dh) rel_e  and  SS_h = sum_e alpha rel_e  give the lin_k_edge / lin_v_edge
// weight gradients as node-level outer products (k_headwise_outer) instead of per-edge ones.  Gradients of the
// source rows (k_node, v_node): the global graph of a scene is symmetric, so they are gathered per source in a fixed
// order from per-edge (alpha, dlogit) scalars through a reverse-edge index (k_reverse_edges, k_gattn_src_bwd); only for
// an asymmetric edge list (not produced by the reference's datasets) are they scattered with float atomics.
#include <string>
#include <unordered_map>

#include "attn_common.hpp"
#include "bwd.hpp"
#include "common.hpp"
#include "dropout.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "tile.hpp"

namespace tsde {

static bool rows_bwd_mm();

// NODE = false: attention over edge rows alone (the AA / AL encoders: k = lin_k(emb_e), v = lin_v(emb_e), `rel` = the stored
// embedding rows): no kn / vn / src, nothing to scatter; UZ may be null (k_gattn_drel then rebuilds U, Z from q and dagg).
template <int HEADS, bool NODE>
__global__ __launch_bounds__(256) void k_gattn_bwd(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                   const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                   const float* __restrict__ q, const float* __restrict__ kn,
                                                   const float* __restrict__ vn, const float* __restrict__ agg,
                                                   const float* __restrict__ dagg, const float* __restrict__ stats, int64_t N,
                                                   float* __restrict__ DQ, float* __restrict__ DKN, float* __restrict__ DVN,
                                                   float* __restrict__ RL, float* __restrict__ SS, float* __restrict__ DAGGM,
                                                   float* __restrict__ EA, float* __restrict__ ED, float* __restrict__ UZ,
                                                   const int32_t* __restrict__ asym, DropArg drop) {
  // One pass over the segment: the softmax statistics come from the forward's tape (stats).  Per edge and head the kernel
  // leaves the two scalars everything else is made of -- EA = alpha d_e (the weight the values were summed with) and
  // ED = d logit / sqrt(dh) -- and per target the vectors U_h = Wke_h^T q_h, Z_h = Wve_h^T dagg_h (UZ):
  //   d rel_e          = sum_layers sum_h ED U_h + EA Z_h             (k_gattn_drel, once for all layers)
  //   d k_node[src], d v_node[src]: gathered per source in a fixed order (k_gattn_src_bwd; symmetric graph, *asym == 0),
  //                                 or scattered right here with float atomics (*asym != 0)
  constexpr int LPH = 64 / HEADS, SL = 64 / LPH, NV = SL / 4;     // as in k_global_attn<HEADS>
  constexpr float INV = HEADS == 4 ? 0.25f : INV_SQRT_DH;
  __shared__ __attribute__((aligned(16))) float sbuf[4][HEADS][64 + 4];
  __shared__ __attribute__((aligned(16))) float srel[4][8][64];   // a wave's chunk of rel rows (wave-private, as in the forward)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row addresses in SGPRs
  const int h = lane / LPH, j = lane % LPH;
  const int64_t node = xcd_block() * 4 + wv;
  const int64_t nc = node < N ? node : N - 1;
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  const float ql = q[nc * 64 + lane];
  const float da = node < N ? dagg[nc * 64 + lane] : 0.f;
  const float cb = NODE ? 0.f : head_sum_n<HEADS>(ql * img[GAttnL::BKE + lane]);   // (the forward drops it with node rows: attn.hip)
  const float cz = head_sum_n<HEADS>(da * img[GAttnL::BVE + lane]);
  const float dlt = head_sum_n<HEADS>(da * agg[nc * 64 + lane]);
  float U[SL], Z[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) U[e] = Z[e] = 0.f;
#pragma unroll
  for (int d = 0; d < LPH; ++d) {
    const float qd = __shfl(ql, LPH * h + d), dd = __shfl(da, LPH * h + d);
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const f4 kw = *reinterpret_cast<const f4*>(wke + (LPH * h + d) * 64 + SL * j + 4 * v4);
      const f4 vw = *reinterpret_cast<const f4*>(wve + (LPH * h + d) * 64 + SL * j + 4 * v4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        U[4 * v4 + e] = fmaf(kw[e], qd, U[4 * v4 + e]);
        Z[4 * v4 + e] = fmaf(vw[e], dd, Z[4 * v4 + e]);
      }
    }
  }
  if (node < N && UZ != nullptr) {
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      *reinterpret_cast<f4*>(UZ + ((node * HEADS + h) * 2 + 0) * 64 + SL * j + 4 * v4) = f4{U[4 * v4], U[4 * v4 + 1], U[4 * v4 + 2], U[4 * v4 + 3]};
      *reinterpret_cast<f4*>(UZ + ((node * HEADS + h) * 2 + 1) * 64 + SL * j + 4 * v4) = f4{Z[4 * v4], Z[4 * v4 + 1], Z[4 * v4 + 2], Z[4 * v4 + 3]};
    }
  }
  const int beg = segptr[nc], end = node < N ? segptr[nc + 1] : beg;
  const float m = stats[(nc * HEADS + h) * 2], inv = stats[(nc * HEADS + h) * 2 + 1];
  const bool scatter = NODE && *asym != 0;
  // Attention dropout (AGG:116): agg_h = sum_e alpha_e d_e (v_node[src] + lin_v_edge(rel_e))_h with d_e = keep_e / (1 - p):
  // every "alpha" that multiplies a VALUE below becomes alpha d_e, d alpha_e = d_e (dagg . value_e), and lin_v_edge.bias
  // sees sum_e alpha_e d_e instead of 1
  const bool dropping = drop.p > 0.f;
  float sad = 0.f, sal = 0.f;
  float dqe = 0.f, Rl[SL], Sa[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) Rl[e] = Sa[e] = 0.f;
  // row loads through buffer descriptors as in the forward kernel (attn.hip k_global_attn): descriptor base in SGPRs + scalar row
  // offset + lane * 4 -- no per-lane 64-bit address arithmetic for the 24 loads of a chunk
  const __amdgpu_buffer_rsrc_t rs_rel = row_rsrc(rel + int64_t(beg) * 64);
  const __amdgpu_buffer_rsrc_t rs_kn = row_rsrc(kn), rs_vn = row_rsrc(vn);
  for (int e0 = beg; e0 < end; e0 += 8) {                  // 8 edges in flight per round trip, as in the forward kernel
    f4 r[8][NV];
    float knv[8], vnv[8], rl[8];
    float kp[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (dropping) drop_attn_chunk<8>(kp, drop, uint32_t(nc), uint32_t(e0 - beg), lane, h);
    const int sv = NODE ? src[e0 + (lane & 7) < end ? e0 + (lane & 7) : end - 1] : 0;   // coalesced, handed out as scalars
    int sx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u < end ? e0 + u : end - 1;
      sx[u] = __builtin_amdgcn_readlane(sv, u);
      rl[u] = row_load(rs_rel, lane, e - beg);             // the row once per wave, slices picked out of LDS below
      knv[u] = NODE ? row_load(rs_kn, lane, sx[u]) : 0.f;
      vnv[u] = NODE ? row_load(rs_vn, lane, sx[u]) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();                        // the previous chunk's slice reads are done (same wave, in order)
#pragma unroll
    for (int u = 0; u < 8; ++u) srel[wv][u][lane] = rl[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4) r[u][v4] = *reinterpret_cast<const f4*>(&srel[wv][u][SL * j + 4 * v4]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u;
      if (e >= end) break;
      float p = ql * knv[u];
      float t = da * vnv[u];
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          p = fmaf(r[u][v4][c], U[4 * v4 + c], p);
          t = fmaf(r[u][v4][c], Z[4 * v4 + c], t);
        }
      const float lg = (head_sum_n<HEADS>(p) + cb) * INV;
      const float alpha = fast_exp(lg - m) * inv;
      const float alk = alpha * kp[u];                     // the weight the values were summed with
      sad += alk;
      sal += alpha;
      const float dal = (head_sum_n<HEADS>(t) + cz) * kp[u];
      const float dls = alpha * (dal - dlt) * INV;
      dqe = fmaf(dls, knv[u], dqe);
      if (j == 0) {
        EA[int64_t(e) * HEADS + h] = alk;
        ED[int64_t(e) * HEADS + h] = dls;
      }
      if (scatter) {
        atomicAdd(DKN + int64_t(sx[u]) * 64 + lane, dls * ql);
        atomicAdd(DVN + int64_t(sx[u]) * 64 + lane, alk * da);
      }
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          Rl[4 * v4 + c] = fmaf(dls, r[u][v4][c], Rl[4 * v4 + c]);
          Sa[4 * v4 + c] = fmaf(alk, r[u][v4][c], Sa[4 * v4 + c]);
        }
    }
  }
#pragma unroll
  for (int v4 = 0; v4 < NV; ++v4) {
    const f4 rv = f4{Rl[4 * v4], Rl[4 * v4 + 1], Rl[4 * v4 + 2], Rl[4 * v4 + 3]};
    *reinterpret_cast<f4*>(&sbuf[wv][h][SL * j + 4 * v4]) = rv;
    if (node < N) {
      *reinterpret_cast<f4*>(RL + (node * HEADS + h) * 64 + SL * j + 4 * v4) = rv;
      *reinterpret_cast<f4*>(SS + (node * HEADS + h) * 64 + SL * j + 4 * v4) = f4{Sa[4 * v4], Sa[4 * v4 + 1], Sa[4 * v4 + 2], Sa[4 * v4 + 3]};
    }
  }
  // d q[d] = sum_e dlogit/sqrt(dh) (k_node[src][d] + lin_k_edge(rel_e)[d]) = dqe + Wke[d] . RL_head(d)
  float dq = dqe;
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
    const f4 wr = *reinterpret_cast<const f4*>(wke + lane * 64 + 4 * k4);
    const f4 rv = *reinterpret_cast<const f4*>(&sbuf[wv][h][4 * k4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) dq = fmaf(wr[e], rv[e], dq);
  }
  if (node < N) {
    DQ[node * 64 + lane] = dq;
    DAGGM[node * 64 + lane] = da * (dropping ? sad : sal);            // lin_v_edge.bias sees sum_e alpha d_e (= 1 without dropout)
  }
}

// d rel_e = sum over the layers l and heads h of  ED_l[e][h] U_l[tgt][h] + EA_l[e][h] Z_l[tgt][h]  -- the relative-pose embedding
// is shared by every layer (AGG:42-51), so its gradient is assembled ONCE from the per-edge scalars and per-target vectors the
// layers' k_gattn_bwd left, instead of a read-modify-write of the [E,64] rows per layer.  One wave per target, lane = column;
// a chunk's 64 / HEADS edges x HEADS scalars arrive in one coalesced load per (layer, EA | ED) and are handed out as scalars.
// FROM_ROWS (NL = 1): U, Z are rebuilt from the target's q / dagg rows and the GAttnL image instead of read from UZ (the AA
// encoder has H x Nt targets of ~20 edges each: storing 4 KB of (U, Z) per target would cost more than it saves).
template <int HEADS, int NL, bool FROM_ROWS>
__global__ __launch_bounds__(256) void k_gattn_drel(DrelArgs a, const int32_t* __restrict__ segptr, int64_t N, float* __restrict__ DREL,
                                                    int accumulate) {
  constexpr int EPC = 64 / HEADS, LPH = 64 / HEADS;
  const int lane = threadIdx.x & 63;
  const int64_t node = xcd_block() * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= N) return;
  float U[NL][HEADS], Z[NL][HEADS];
  if (FROM_ROWS) {
    static_assert(!FROM_ROWS || NL == 1, "FROM_ROWS handles one layer");
    const float ql = a.q[node * 64 + lane], dl = a.dagg[node * 64 + lane];
    const float* wke = a.img + GAttnL::WKE;
    const float* wve = a.img + GAttnL::WVE;
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      float u = 0.f, z = 0.f;
#pragma unroll
      for (int d = 0; d < LPH; ++d) {
        const float qd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ql), LPH * h + d));
        const float dd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), LPH * h + d));
        u = fmaf(wke[(LPH * h + d) * 64 + lane], qd, u);
        z = fmaf(wve[(LPH * h + d) * 64 + lane], dd, z);
      }
      U[0][h] = u;
      Z[0][h] = z;
    }
  } else {
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        U[l][h] = a.UZ[l][((node * HEADS + h) * 2 + 0) * 64 + lane];
        Z[l][h] = a.UZ[l][((node * HEADS + h) * 2 + 1) * 64 + lane];
      }
  }
  const int beg = segptr[node], end = segptr[node + 1];
  for (int e0 = beg; e0 < end; e0 += EPC) {
    float eav[NL], edv[NL];
    const int64_t idx = int64_t(e0) * HEADS + lane;
    const bool ok = idx < int64_t(end) * HEADS;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      eav[l] = ok ? a.EA[l][idx] : 0.f;
      edv[l] = ok ? a.ED[l][idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < EPC; ++u) {
      if (e0 + u >= end) break;
      float dr = 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
          const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edv[l]), u * HEADS + h));
          const float al = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eav[l]), u * HEADS + h));
          dr = fmaf(d, U[l][h], dr);
          dr = fmaf(al, Z[l][h], dr);
        }
      float* out = DREL + int64_t(e0 + u) * 64 + lane;
      *out = accumulate ? *out + dr : dr;
    }
  }
}

// The same sum as a tile product on the fp32 matrix cores (round 4; 8 heads, U / Z from UZ): DREL[16 edges x 64] = S[16 x 16 NL] . UZ[16 NL x 64],
// S = the edges' (ED | EA) scalars of the NL layers, contraction index k = 16 l + 8 type + h.  One wave per target: its 16 NL x 64 block of
// UZ is loaded once (lane (kk, n): rows 4 s + kk, columns 4n .. 4n+3 -- the B operand under the column order 4n + b), a tile's scalars are
// 4 NL dword loads per lane (lane (e, kk): heads kk and 4 + kk of every (layer, type)), and the result leaves as whole rows (lane (n, q):
// edges 4q + i, columns 4n .. 4n+3).  48 multiply-adds + 48 lane reads per edge and lane on the vector pipe became 3 matrix instructions.
template <int NL>
__global__ __launch_bounds__(256) void k_gattn_drel_mm(DrelArgs a, const int32_t* __restrict__ segptr, int64_t N, float* __restrict__ DREL,
                                                       int accumulate) {
  constexpr int HEADS = 8;
  const int lane = threadIdx.x & 63;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int64_t node = xcd_block() * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= N) return;
  // B operand: step s = 4 l + 2 type + hh covers k = 16 l + 8 type + 4 hh + kk, i.e. head 4 hh + kk of (layer l, type)
  f4 uz[4 * NL];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        uz[4 * l + 2 * ty + hh] = *reinterpret_cast<const f4*>(a.UZ[l] + ((node * HEADS + 4 * hh + q4) * 2 + ty) * 64 + 4 * c16);
  const int beg = segptr[node], end = segptr[node + 1];
  for (int e0 = beg; e0 < end; e0 += 16) {
    const int e = e0 + c16, ec = e < end ? e : end - 1;
    float sv[4 * NL];                                            // A operand: lane (e = c16, kk = q4), step s: S[e][16 l + 8 type + 4 hh + kk]
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        sv[4 * l + 0 + hh] = a.ED[l][int64_t(ec) * HEADS + 4 * hh + q4];      // type 0: ED multiplies U
        sv[4 * l + 2 + hh] = a.EA[l][int64_t(ec) * HEADS + 4 * hh + q4];      // type 1: EA multiplies Z
      }
    __builtin_amdgcn_sched_barrier(0);                            // all 4 NL loads requested before the first product waits for one
    f4 D[4];                                                      // (the scheduler otherwise recycles two registers: 2 loads in flight)
#pragma unroll
    for (int b = 0; b < 4; ++b) D[b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * NL; ++s)
#pragma unroll
      for (int b = 0; b < 4; ++b) D[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[s], uz[s][b], D[b], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int er = e0 + 4 * q4 + i;
      if (er < end) {
        float* out = DREL + int64_t(er) * 64 + 4 * c16;
        f4 v = f4{D[0][i], D[1][i], D[2][i], D[3][i]};
        if (accumulate) {
          const f4 old = *reinterpret_cast<const f4*>(out);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = old[c] + v[c];
        }
        *reinterpret_cast<f4*>(out) = v;
      }
    }
  }
}

int run_gattn_drel(hipStream_t st, int heads, int nl, bool from_rows, const DrelArgs& da, const int32_t* segptr, int64_t N, float* DREL,
                   int accumulate) {
#define TS_DREL(H_, N_, F_) TS_LAUNCH((k_gattn_drel<H_, N_, F_>), xcd_grid(cdiv(N, 4)), 256, 0, st, da, segptr, N, DREL, accumulate)
  if (from_rows) {
    if (heads == 4) TS_DREL(4, 1, true);
    else TS_DREL(8, 1, true);
  } else if (heads == 4) {
    switch (nl) { case 1: TS_DREL(4, 1, false); break; case 2: TS_DREL(4, 2, false); break; case 3: TS_DREL(4, 3, false); break; default: TS_DREL(4, 4, false); }
  } else if (rows_bwd_mm()) {                                    // the matrix-core form (TRAJSDE_ROWS_BWD_MM=0: the vector form)
#define TS_DREL_MM(N_) TS_LAUNCH_TAG("(k_gattn_drel<8, " #N_ ", false>)", false, (k_gattn_drel_mm<N_>), xcd_grid(cdiv(N, 4)), 256, 0, st, da, segptr, N, DREL, accumulate)
    switch (nl) { case 1: TS_DREL_MM(1); break; case 2: TS_DREL_MM(2); break; case 3: TS_DREL_MM(3); break; default: TS_DREL_MM(4); }
#undef TS_DREL_MM
  } else {
    switch (nl) { case 1: TS_DREL(8, 1, false); break; case 2: TS_DREL(8, 2, false); break; case 3: TS_DREL(8, 3, false); break; default: TS_DREL(8, 4, false); }
  }
#undef TS_DREL
  return TRAJSDE_OK;
}

// ---- k_gattn_bwd<8, false> on the fp32 matrix cores (round 4).  The kernel above spends 64 of its ~95 vector instructions per edge on
// multiply-adds: p = rel_e . U_h, t = rel_e . Z_h (16 columns for the 8 heads) and RL_h += dlogit rel_e, SS_h += alpha rel_e (16 rows).
// Both are tile products -- [16 edges x 64] x [64 x 16] and [16 x 16 edges] x [16 edges x 64] -- and v_mfma_f32_16x16x4_f32 does them in
// exact fp32 products with no operand split: 32 matrix instructions (1 024 pipe cycles) per 16 edges instead of 1 024 vector
// multiply-adds (4 096 issue cycles).  One wave per target, tiles of 16 of its edges:
//   * the tile's rows are loaded once, whole rows per 16 lanes (lane (kk, n), register r: row 4 kk + r, columns 4n..4n+3) -- which IS the
//     B operand of the second product under the column order 4n + b -- and pass through a wave-private LDS tile to become the A operand
//     of the first (row on lane: lane (e, kk) holds columns 16 kk .. 16 kk + 15; the contraction runs in that order on both sides);
//   * the first product's result -- lane (j, q) holds edges 4q + i of column j -- is where the softmax scalars are computed (columns j < 8
//     are the logits' p of head j, j >= 8 the t of head j - 8: partner lanes swap by a row rotation of 8), and their results
//     W[e][j] = (dlogit | alpha d) sit exactly where the second product's A operand wants them (row j, contraction index 4q + i): no
//     shuffle between the two products.
// Same inputs, outputs and per-edge scalars as the vector form; the sums run in another order (tolerances of the gradient tests).
// Persistent: one workgroup of 16 waves per CU, every wave walks its own targets.  What a target needs before its first tile -- the B
// operand [U | Z], 128 multiply-adds per lane over 128 weights that depend on the lane alone -- comes from an LDS image of lin_k_edge |
// lin_v_edge in the lanes' order (measured: the same weights re-read from L1 / L2 per target, as the vector form does, cost 0.6 of this
// kernel's 1.1 ms; the tile loop itself 0.1).
constexpr int RMM_WAVES = 16;
constexpr int RMM_TP = 68;                                         // padded row of a wave's staging tile / of the plain weight copy
constexpr int RMM_WIMG = 8192, RMM_WK = 64 * RMM_TP;              // floats: swizzled [Wke | Wve] image, plain Wke rows
constexpr int RMM_PER_WAVE = 16 * RMM_TP + 8 * RMM_TP;            // staging tile + the eight RL rows of the d q epilogue
constexpr int rows_mm_lds_bytes() { return (RMM_WIMG + RMM_WK + RMM_WAVES * RMM_PER_WAVE) * 4; }
// FUSEW: the lin_k / lin_v weight gradients -- head-wise outer products q_h (x) RL_h, dagg_h (x) SS_h summed over the targets -- are
// accumulated HERE, per wave, in 128 registers (lane = column c: GW[matrix][head][half][i] = dW[8 h + 4 half + i][c]) by 32
// `v_mfma_f32_4x4x1_16B_f32` per target (sixteen 4 x 4 outer products an instruction: a = four entries of q_h, b = four of RL_h), instead of
// RL / SS going to HBM (4 KB a target: 671 MB written and read again by k_headwise_outer at 64 x 128).  Workgroups of 8 waves then (the
// accumulators double the registers); a workgroup's eight partial sums are added in wave order through LDS at the end (fixed order:
// reproducible) and leave as ONE pair of 64 x 64 partials per workgroup, summed by the deferred reduce like any other weight gradient.
constexpr int RMMF_WAVES = 8;
constexpr int RMMF_PER_WAVE = 16 * RMM_TP + 16 * RMM_TP + 128;      // staging tile + the sixteen RL | SS rows + q | dagg
constexpr int rows_mmf_lds_bytes() { return (RMM_WIMG + RMM_WK + RMMF_WAVES * RMMF_PER_WAVE + 2 * 4096 + 64) * 4; }
template <bool DROP, bool FUSEW>
__global__ __launch_bounds__(64 * (FUSEW ? RMMF_WAVES : RMM_WAVES)) void k_edge_rows_bwd_mm(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                     const float* __restrict__ rel, const float* __restrict__ q,
                                                                     const float* __restrict__ agg, const float* __restrict__ dagg,
                                                                     const float* __restrict__ stats, int64_t N, float* __restrict__ DQ,
                                                                     float* __restrict__ RL, float* __restrict__ SS,
                                                                     float* __restrict__ DAGGM, float* __restrict__ EA,
                                                                     float* __restrict__ ED, DropArg drop, float* __restrict__ wpart,
                                                                     float* __restrict__ wcs) {
  constexpr int HEADS = 8;
  constexpr float INV = INV_SQRT_DH;
  constexpr int TP = RMM_TP;
  constexpr int WAVES = FUSEW ? RMMF_WAVES : RMM_WAVES, PER_WAVE = FUSEW ? RMMF_PER_WAVE : RMM_PER_WAVE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wimg = lds;                                         // [d][kk][v4][j][4]: W_(j<8 ? k : v)[8 (j&7) + d][16 kk + 4 v4 + e]
  float* const wk = lds + RMM_WIMG;                                // Wke[d][c], rows padded to TP
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const tile = lds + RMM_WIMG + RMM_WK + wv * PER_WAVE;
  float* const sb = tile + 16 * TP;                                // [8][TP] (FUSEW: [16][TP], the SS rows too)
  float* const xsm = sb + 16 * TP;                                 // FUSEW: [64] q | [64] dagg of the current target
  f4 GW[FUSEW ? 2 : 1][FUSEW ? 8 : 1][FUSEW ? 2 : 1];
  float gbv = 0.f;                                                 // FUSEW: lin_v.bias gradient, lane = channel (sum over the wave's targets of dagg * sum_e alpha d)
#pragma unroll
  for (int a_ = 0; a_ < (FUSEW ? 2 : 1); ++a_)
#pragma unroll
    for (int b_ = 0; b_ < (FUSEW ? 8 : 1); ++b_)
#pragma unroll
      for (int c_ = 0; c_ < (FUSEW ? 2 : 1); ++c_) GW[a_][b_][c_] = f4{0.f, 0.f, 0.f, 0.f};
  const int c16 = lane & 15, q4 = lane >> 4, hd = c16 & 7;
  const bool lo8 = c16 < 8;
  {
    const float* wke = img + GAttnL::WKE;
    const float* wve = img + GAttnL::WVE;
    for (int i = threadIdx.x; i < RMM_WIMG / 4; i += blockDim.x) {  // one f4 of the image per step
      const int j = i & 15, v4 = (i >> 4) & 3, kk = (i >> 6) & 3, d = i >> 8;
      const float* src = (j < 8 ? wke : wve) + (8 * (j & 7) + d) * 64 + 16 * kk + 4 * v4;
      *reinterpret_cast<f4*>(wimg + 4 * i) = *reinterpret_cast<const f4*>(src);
    }
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
      const int row = i >> 4, c4 = i & 15;
      *reinterpret_cast<f4*>(wk + row * TP + 4 * c4) = *reinterpret_cast<const f4*>(wke + row * 64 + 4 * c4);
    }
  }
  __syncthreads();
  const float bke = img[GAttnL::BKE + lane], bve = img[GAttnL::BVE + lane];
  const int64_t stride = int64_t(gridDim.x) * WAVES;
  // The first tile of the NEXT target is requested before the current target's epilogue, into the row registers its last tile has just
  // freed (the segment bounds of the next target a whole target ahead): a target is two tiles on average, and its first rows used to wait
  // for segptr -> address -> HBM at the top of every target.
  f4 nx[4];
  int beg_next = 0, end_next = 0;
  auto fetch_rows = [&](int e0, int end_) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = e0 + 4 * q4 + r;
      const int ec = e < end_ ? e : end_ - 1;
      nx[r] = *reinterpret_cast<const f4*>(rel + int64_t(ec) * 64 + 4 * c16);
    }
  };
  {
    const int64_t n0 = int64_t(blockIdx.x) * WAVES + wv;
    if (n0 < N) {
      beg_next = segptr[n0];
      end_next = segptr[n0 + 1];
      if (beg_next < end_next) fetch_rows(beg_next, end_next);
    }
  }
  for (int64_t node = int64_t(blockIdx.x) * WAVES + wv; node < N; node += stride) {
    // per-head constants, first in the "lane = channel" layout (lanes 8h .. 8h+7 hold head h), then handed to the lanes of column hd
    const float ql = q[node * 64 + lane];
    const float da = dagg[node * 64 + lane];
    const float ag = agg[node * 64 + lane];
    if (FUSEW) {
      __builtin_amdgcn_wave_barrier();                             // (the previous target's readers are done)
      xsm[lane] = ql;
      xsm[64 + lane] = da;
    }
    const float m = stats[(node * HEADS + hd) * 2], inv = stats[(node * HEADS + hd) * 2 + 1];
    const int beg = beg_next, end = end_next;
    {
      const int64_t nn = node + stride < N ? node + stride : node;  // (uniform; the last target reads its own bounds again)
      beg_next = segptr[nn];
      end_next = segptr[nn + 1];
    }
    auto fetch = [&](int e0) { fetch_rows(e0, end); };
    const float cb = __shfl(head_sum_n<HEADS>(ql * bke), 8 * hd);
    const float cz = __shfl(head_sum_n<HEADS>(da * bve), 8 * hd);
    const float dlt = __shfl(head_sum_n<HEADS>(da * ag), 8 * hd);
    // B operand of the first product: lane (kk = q4, j = c16) holds column j of [U | Z] at rows 16 kk + s
    float uz[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) uz[s] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float qd = __shfl(ql, 8 * hd + d), dd = __shfl(da, 8 * hd + d);
      const float x = lo8 ? qd : dd;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const f4 wr = *reinterpret_cast<const f4*>(wimg + (((d * 4 + q4) * 4 + v4) * 16 + c16) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) uz[4 * v4 + e] = fmaf(wr[e], x, uz[4 * v4 + e]);
      }
    }
    f4 R[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) R[b] = f4{0.f, 0.f, 0.f, 0.f};
    float sal = 0.f, sad = 0.f;
    for (int e0 = beg; e0 < end; e0 += 16) {
      __builtin_amdgcn_wave_barrier();                             // the previous readers of the tile are done (same wave, in order)
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(tile + (4 * q4 + r) * TP + 4 * c16) = nx[r];
      if (e0 + 16 < end) fetch(e0 + 16);                           // the next tile's rows, into the registers just staged
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f4 a[4];
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) a[v4] = *reinterpret_cast<const f4*>(tile + c16 * TP + 16 * q4 + 4 * v4);
      f4 P0 = f4{0.f, 0.f, 0.f, 0.f}, P1 = P0;                       // two chains: a matrix instruction waits for its accumulator
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][0], uz[4 * v4 + 0], P0, 0, 0, 0);
        P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][1], uz[4 * v4 + 1], P1, 0, 0, 0);
        P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][2], uz[4 * v4 + 2], P0, 0, 0, 0);
        P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][3], uz[4 * v4 + 3], P1, 0, 0, 0);
      }
      f4 W;
      uint32_t mine = 0u;
      if (DROP) {                                                  // lane (x, r) draws the block of edge x and keeps word r (heads 2r, 2r+1)
        uint32_t w[4];
        philox_words(drop.seed, drop_stream(drop, DK_ATTN), uint32_t(e0 - beg) + uint32_t(c16), uint32_t(node), 0u, w);
        mine = q4 == 0 ? w[0] : (q4 == 1 ? w[1] : (q4 == 2 ? w[2] : w[3]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + 4 * q4 + i;
        const float x = P0[i] + P1[i];
        const float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));   // row_ror:8 -- column j ^ 8
        const float p = lo8 ? x : y, t = lo8 ? y : x;
        float kp = 1.f;
        if (DROP) {
          const uint32_t word = uint32_t(__shfl(int(mine), 16 * (hd >> 1) + 4 * q4 + i));
          kp = drop_pick(word, hd & 1, drop);
        }
        const float lg = (p + cb) * INV;
        const float alpha = e < end ? fast_exp(lg - m) * inv : 0.f;
        const float alk = alpha * kp;
        const float dal = (t + cz) * kp;
        const float dls = alpha * (dal - dlt) * INV;
        sal += alpha;
        sad += alk;
        W[i] = lo8 ? dls : alk;
        if (e < end) (lo8 ? ED : EA)[int64_t(e) * HEADS + hd] = W[i];
      }
      {
        f4 rw[4];                                                  // the tile's rows again, whole rows per 16 lanes: B operand under the column order 4n + b
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[r] = *reinterpret_cast<const f4*>(tile + (4 * q4 + r) * TP + 4 * c16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int b = 0; b < 4; ++b) R[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i], rw[i][b], R[b], 0, 0, 0);
      }
    }
    if (node + stride < N && beg_next < end_next) fetch_rows(beg_next, end_next);      // the next target's first tile, under this epilogue
    // lane (n, q) holds rows 4q + i' of the 16 x 64 sums at columns 4n .. 4n+3: rows 0-7 are RL of heads 0-7, rows 8-15 SS
    __builtin_amdgcn_wave_barrier();                               // (the previous target's epilogue reads of sb are done)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f4 rv = f4{R[0][i], R[1][i], R[2][i], R[3][i]};
      const int row = 4 * q4 + i;
      if (FUSEW || row < 8) *reinterpret_cast<f4*>(sb + row * TP + 4 * c16) = rv;
      if (!FUSEW) *reinterpret_cast<f4*>((row < 8 ? RL : SS) + (node * HEADS + (row & 7)) * 64 + 4 * c16) = rv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (FUSEW) {
      // dW[8 h + 4 half + i][c] += x[8 h + 4 half + i] * row_h[c]: block b = c >> 2 of the instruction's sixteen, lane = c
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
          const float bv = sb[(8 * mt + hh) * TP + lane];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float av = xsm[64 * mt + 8 * hh + 4 * hf + (lane & 3)];
            GW[mt][hh][hf] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, GW[mt][hh][hf], 0, 0, 0);
          }
        }
    }
    // d q[d] = Wke[d] . RL_head(d)   (lane = channel d from here on)
    const int h = lane >> 3;
    float dq = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const f4 wr = *reinterpret_cast<const f4*>(wk + lane * TP + 4 * k4);
      const f4 rv = *reinterpret_cast<const f4*>(sb + h * TP + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dq = fmaf(wr[e], rv[e], dq);
    }
    // sum_e alpha (d_e) of head h: every lane of column hd holds a part (its edges 4q + i); columns hd and hd + 8 hold the same
    float sw = DROP ? sad : sal;
    sw = xor32_sum(xor16_sum(sw));
    const float swh = __shfl(sw, h);
    DQ[node * 64 + lane] = dq;
    if (FUSEW) gbv += da * swh;                                    // lin_v_edge.bias sees sum_e alpha d_e (= 1 without dropout)
    else DAGGM[node * 64 + lane] = da * swh;
  }
  if (FUSEW) {
    // the workgroup's partial: its waves add their registers into one LDS block in wave order, then it leaves as 2 x 4096 floats
    float* const acc = lds + RMM_WIMG + RMM_WK + WAVES * PER_WAVE;
    for (int w = 0; w < WAVES; ++w) {
      __syncthreads();
      if (wv == w) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int hh = 0; hh < 8; ++hh)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float* pa = acc + mt * 4096 + (8 * hh + 4 * hf + i) * 64 + lane;
                *pa = w == 0 ? GW[mt][hh][hf][i] : *pa + GW[mt][hh][hf][i];
              }
        float* pb = acc + 2 * 4096 + lane;
        *pb = w == 0 ? gbv : *pb + gbv;
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) wcs[(int64_t(gridDim.x) + blockIdx.x) * 64 + threadIdx.x] = acc[2 * 4096 + threadIdx.x];      // the cs slot of the lin_v partial
    for (int i = threadIdx.x; i < 2 * 1024; i += blockDim.x) {
      const int mt = i >> 10, k = i & 1023;
      *reinterpret_cast<f4*>(wpart + (int64_t(mt) * gridDim.x + blockIdx.x) * 4096 + 4 * k) = *reinterpret_cast<const f4*>(acc + mt * 4096 + 4 * k);
    }
  }
}

// ---- k_gattn_bwd<8, true> (the global interactor's attention backward) in the same form: the rel-row products of a 16-edge tile on the
// fp32 matrix cores, what belongs to the gathered node rows -- q_h . k_node[src]_h, dagg_h . v_node[src]_h and d q += dlogit k_node[src] --
// on the vector pipe over the rows as they are loaded (whole rows per 16 lanes: lane (kk, n), register r: the source of edge 4 kk + r,
// columns 4n .. 4n+3, half a head; gattn_f32.hip k_global_attn_mf is the forward twin).  Workgroups of 8 waves, one per CU; the rel rows of
// tile t+2 are requested when tile t is staged, the value rows of t+1 once t's node dots are done, its key rows once t's d q sums are.
constexpr int GBM_WAVES = 8;
constexpr int GBM_PER_WAVE = 16 * RMM_TP + 16 * 16 + 16 * 8 + 16 * 8 + 3 * 64;   // staging tile (RL rows at the end) | node dots | dlogit | alpha d | d q node part, q, dagg
constexpr int gattn_bwd_mm_lds_bytes() { return (RMM_WIMG + RMM_WK + GBM_WAVES * GBM_PER_WAVE) * 4; }
template <bool DROP>
__global__ __launch_bounds__(64 * GBM_WAVES) void k_gattn_bwd_mm(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                                 const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                                 const float* __restrict__ q, const float* __restrict__ kn,
                                                                 const float* __restrict__ vn, const float* __restrict__ agg,
                                                                 const float* __restrict__ dagg, const float* __restrict__ stats, int64_t N,
                                                                 float* __restrict__ DQ, float* __restrict__ DKN, float* __restrict__ DVN,
                                                                 float* __restrict__ RL, float* __restrict__ SS, float* __restrict__ DAGGM,
                                                                 float* __restrict__ EA, float* __restrict__ ED, float* __restrict__ UZ,
                                                                 const int32_t* __restrict__ asym, DropArg drop) {
  constexpr int HEADS = 8;
  constexpr float INV = INV_SQRT_DH;
  constexpr int TP = RMM_TP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wimg = lds;                                         // [d][kk][v4][j][4]: W_(j<8 ? k : v)[8 (j&7) + d][16 kk + 4 v4 + e]
  float* const wk = lds + RMM_WIMG;                                // Wke[d][c], rows padded to TP
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const tile = lds + RMM_WIMG + RMM_WK + wv * GBM_PER_WAVE;
  float* const pnt = tile + 16 * TP;                               // [16 edges][16]: node part of p (heads 0-7) | of t (8-15)
  float* const wtd = pnt + 256;                                    // [16 edges][8 heads]: dlogit / sqrt(dh)
  float* const wta = wtd + 128;                                    // [16 edges][8 heads]: alpha d
  float* const nsum = wta + 128;                                   // [64]: node part of d q
  float* const qsm = nsum + 64;                                    // [64] q row | [64] dagg row
  const int c16 = lane & 15, q4 = lane >> 4, hd = c16 & 7;
  const bool lo8 = c16 < 8;
  {
    const float* wke = img + GAttnL::WKE;
    const float* wve = img + GAttnL::WVE;
    for (int i = threadIdx.x; i < RMM_WIMG / 4; i += blockDim.x) {
      const int j = i & 15, v4 = (i >> 4) & 3, kk = (i >> 6) & 3, d = i >> 8;
      const float* sp = (j < 8 ? wke : wve) + (8 * (j & 7) + d) * 64 + 16 * kk + 4 * v4;
      *reinterpret_cast<f4*>(wimg + 4 * i) = *reinterpret_cast<const f4*>(sp);
    }
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
      const int row = i >> 4, c4 = i & 15;
      *reinterpret_cast<f4*>(wk + row * TP + 4 * c4) = *reinterpret_cast<const f4*>(wke + row * 64 + 4 * c4);
    }
  }
  __syncthreads();
  const float bve = img[GAttnL::BVE + lane];
  const bool scatter = *asym != 0;
  const int64_t stride = int64_t(gridDim.x) * GBM_WAVES;
  for (int64_t node = xcd_block() * GBM_WAVES + wv; node < N; node += stride) {        // (xcd_block: a scene's node rows through one L2, gattn_f32.hip)
    const float ql = q[node * 64 + lane];
    const float da = dagg[node * 64 + lane];
    const float ag = agg[node * 64 + lane];
    const float m = stats[(node * HEADS + hd) * 2], inv = stats[(node * HEADS + hd) * 2 + 1];
    const int beg = segptr[node], end = segptr[node + 1];
    const int last = end > beg ? end - 1 : beg;                    // (an empty segment loads row `beg` of the next target, or nothing is used of it)
    auto src_at = [&](int (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = end > beg ? src[e < end ? e : last] : 0;
      }
    };
    auto rel_load = [&](f4 (&dst)[4], int e0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = e0 + 4 * q4 + r;
        dst[r] = *reinterpret_cast<const f4*>(rel + int64_t(e < end ? e : last) * 64 + 4 * c16);
      }
    };
    auto row_load = [&](f4 (&dst)[4], const float* __restrict__ base, const int (&sidx)[4]) {    // four whole rows per instruction
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const f4*>(base + int64_t(sidx[r]) * 64 + 4 * c16);
    };
    int s_cur[4], s_next[4];
    f4 nxa[4], nxb[4], kr[4], vr[4];
    if (end > beg) {                                               // (uniform)
      src_at(s_cur, beg);
      src_at(s_next, beg + 16);
      rel_load(nxa, beg);
      row_load(kr, kn, s_cur);
      row_load(vr, vn, s_cur);
      rel_load(nxb, beg + 16);
    }
    const float cz = __shfl(head_sum_n<HEADS>(da * bve), 8 * hd);
    const float dlt = __shfl(head_sum_n<HEADS>(da * ag), 8 * hd);
    // B operand of the first product: lane (kk = q4, j = c16) holds column j of [U | Z] at rows 16 kk + s; it is also what k_gattn_drel reads
    float uz[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) uz[s] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float qd = __shfl(ql, 8 * hd + d), dd = __shfl(da, 8 * hd + d);
      const float x = lo8 ? qd : dd;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const f4 wr = *reinterpret_cast<const f4*>(wimg + (((d * 4 + q4) * 4 + v4) * 16 + c16) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) uz[4 * v4 + e] = fmaf(wr[e], x, uz[4 * v4 + e]);
      }
    }
    if (UZ != nullptr) {
      float* up = UZ + ((node * HEADS + hd) * 2 + (lo8 ? 0 : 1)) * 64 + 16 * q4;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) *reinterpret_cast<f4*>(up + 4 * v4) = f4{uz[4 * v4], uz[4 * v4 + 1], uz[4 * v4 + 2], uz[4 * v4 + 3]};
    }
    __builtin_amdgcn_wave_barrier();                               // (the previous target's readers of the wave's LDS are done)
    qsm[lane] = ql;
    qsm[64 + lane] = da;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f4 qn = *reinterpret_cast<const f4*>(qsm + 4 * c16);      // the query's / the incoming gradient's columns 4n .. 4n+3
    const f4 dn = *reinterpret_cast<const f4*>(qsm + 64 + 4 * c16);
    f4 R[4], accq = f4{0.f, 0.f, 0.f, 0.f};                          // accq: columns 4n .. 4n+3 of sum_e dlogit k_node[src], this lane row's edges
#pragma unroll
    for (int b = 0; b < 4; ++b) R[b] = f4{0.f, 0.f, 0.f, 0.f};
    float sal = 0.f, sad = 0.f;
    auto tile_step = [&](f4 (&nx)[4], int e0) {
      int s_after[4];
      src_at(s_after, e0 + 32);
      __builtin_amdgcn_wave_barrier();                             // the previous readers of the wave's LDS are done (same wave, in order)
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(tile + (4 * q4 + r) * TP + 4 * c16) = nx[r];
      rel_load(nx, e0 + 32);                                       // two tiles ahead, into the registers just staged
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f4 P0 = f4{0.f, 0.f, 0.f, 0.f}, P1 = P0;                       // two chains: a matrix instruction waits for its accumulator
      {
        f4 a[4];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) a[v4] = *reinterpret_cast<const f4*>(tile + c16 * TP + 16 * q4 + 4 * v4);
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][0], uz[4 * v4 + 0], P0, 0, 0, 0);
          P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][1], uz[4 * v4 + 1], P1, 0, 0, 0);
          P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][2], uz[4 * v4 + 2], P0, 0, 0, 0);
          P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v4][3], uz[4 * v4 + 3], P1, 0, 0, 0);
        }
      }
      // node parts of p and t: edge 4 q4 + r, head n >> 1 -- this lane's four columns and its neighbour's (lane ^ 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = qn[0] * kr[r][0], t = dn[0] * vr[r][0];
#pragma unroll
        for (int e = 1; e < 4; ++e) {
          p = fmaf(qn[e], kr[r][e], p);
          t = fmaf(dn[e], vr[r][e], t);
        }
        p += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xf, 0xf, false));
        pnt[(4 * q4 + r) * 16 + (c16 >> 1) + 8 * (c16 & 1)] = (c16 & 1) ? t : p;
      }
      row_load(vr, vn, s_next);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint32_t mine = 0u;
      if (DROP) {                                                  // lane (x, r) draws the block of edge x and keeps word r (heads 2r, 2r+1)
        uint32_t w[4];
        philox_words(drop.seed, drop_stream(drop, DK_ATTN), uint32_t(e0 - beg) + uint32_t(c16), uint32_t(node), 0u, w);
        mine = q4 == 0 ? w[0] : (q4 == 1 ? w[1] : (q4 == 2 ? w[2] : w[3]));
      }
      f4 W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + 4 * q4 + i;
        const float x = (P0[i] + P1[i]) + pnt[(4 * q4 + i) * 16 + c16];
        const float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));   // row_ror:8 -- column j ^ 8
        const float p = lo8 ? x : y, t = lo8 ? y : x;
        float kp = 1.f;
        if (DROP) {
          const uint32_t word = uint32_t(__shfl(int(mine), 16 * (hd >> 1) + 4 * q4 + i));
          kp = drop_pick(word, hd & 1, drop);
        }
        const float lg = p * INV;                                  // (the key bias q_h . bke_h is not part of the node form: attn.hip)
        const float alpha = e < end ? fast_exp(lg - m) * inv : 0.f;
        const float alk = alpha * kp;
        const float dal = (t + cz) * kp;
        const float dls = alpha * (dal - dlt) * INV;
        sal += alpha;
        sad += alk;
        W[i] = lo8 ? dls : alk;
        (lo8 ? wtd : wta)[(4 * q4 + i) * 8 + hd] = W[i];
        if (e < end) (lo8 ? ED : EA)[int64_t(e) * HEADS + hd] = W[i];
      }
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
      // node part of d q: sum_e dlogit_h k_node[src]; an asymmetric edge list scatters the source rows' gradients here (float atomics)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float wd = wtd[(4 * q4 + r) * 8 + (c16 >> 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) accq[e] = fmaf(kr[r][e], wd, accq[e]);      // (element by element: no packed fp32 arithmetic in the backward files, tests/test_cabi_cpu.py)
        if (scatter && e0 + 4 * q4 + r < end) {
          const float wa = wta[(4 * q4 + r) * 8 + (c16 >> 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(DKN + int64_t(s_cur[r]) * 64 + 4 * c16 + e, wd * qn[e]);
            atomicAdd(DVN + int64_t(s_cur[r]) * 64 + 4 * c16 + e, wa * dn[e]);
          }
        }
      }
      row_load(kr, kn, s_next);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_cur[r] = s_next[r];
        s_next[r] = s_after[r];
      }
    };
    for (int e0 = beg; e0 < end; e0 += 32) {
      tile_step(nxa, e0);
      if (e0 + 16 < end) tile_step(nxb, e0 + 16);
    }
    // lane (n, q) holds rows 4q + i' of the 16 x 64 sums at columns 4n .. 4n+3: rows 0-7 are RL of heads 0-7, rows 8-15 SS
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f4 rv = f4{R[0][i], R[1][i], R[2][i], R[3][i]};
      const int row = 4 * q4 + i;
      if (row < 8) *reinterpret_cast<f4*>(tile + row * TP + 4 * c16) = rv;
      *reinterpret_cast<f4*>((row < 8 ? RL : SS) + (node * HEADS + (row & 7)) * 64 + 4 * c16) = rv;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) accq[e] = xor32_sum(xor16_sum(accq[e]));      // the four lane rows' edges
    if (q4 == 0) *reinterpret_cast<f4*>(nsum + 4 * c16) = accq;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // d q[d] = sum_e dlogit k_node[src][d] + Wke[d] . RL_head(d)   (lane = channel d from here on)
    const int h = lane >> 3;
    float dq = nsum[lane];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const f4 wr = *reinterpret_cast<const f4*>(wk + lane * TP + 4 * k4);
      const f4 rv = *reinterpret_cast<const f4*>(tile + h * TP + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dq = fmaf(wr[e], rv[e], dq);
    }
    float sw = DROP ? sad : sal;
    sw = xor32_sum(xor16_sum(sw));
    const float swh = __shfl(sw, h);
    DQ[node * 64 + lane] = dq;
    DAGGM[node * 64 + lane] = da * swh;                            // lin_v_edge.bias sees sum_e alpha d_e (= 1 without dropout)
  }
}

// the encoders' attention backward over stored embedding rows (NODE = false), see bwd.hpp
// TRAJSDE_ROWS_BWD_MM=0: the vector form (k_gattn_bwd<8, false>) for A/B runs and cross-checks
static bool rows_bwd_mm() {
  static const bool on = []() { const char* e = getenv("TRAJSDE_ROWS_BWD_MM"); return !(e && e[0] == '0'); }();
  return on;
}
// TRAJSDE_ROWS_BWD_FUSEW=0: RL / SS through HBM and k_headwise_outer, as before round 4's fused form
static bool rows_bwd_fusew() {
  static const bool on = []() { const char* e = getenv("TRAJSDE_ROWS_BWD_FUSEW"); return !(e && e[0] == '0'); }();
  return on;
}
int run_edge_attn_bwd(hipStream_t st, int heads, const float* img, const int32_t* segptr, const float* emb, const float* q, const float* agg,
                      const float* dagg, const float* stats, int64_t R, float* DQ, float* RL, float* SS, float* DAGGM, float* EA, float* ED,
                      const DropArg& drop, const WgradCtx* wc, float* wk, float* wv, float* bv, bool* weights_done) {
  const int32_t* ns = nullptr;
  const float* nf = nullptr;
  float* nw = nullptr;
  if (weights_done) *weights_done = false;
  if (heads == 4)
    TS_LAUNCH_TAG("k_edge_attn_rows_bwd<4>", false, (k_gattn_bwd<4, false>), xcd_grid(cdiv(R, 4)), 256, 0, st, img, segptr, ns, emb, q, nf, nf, agg, dagg, stats,
                  R, DQ, nw, nw, RL, SS, DAGGM, EA, ED, nw, ns, drop);
  else if (rows_bwd_mm()) {
    ReduceQueue* rq = active_reduce_queue();
    const int gridf = int(std::min<int64_t>(256, cdiv(R, RMMF_WAVES)));
    if (rq && wc && rq->part == wc->part && wk && wv && bv && weights_done && rows_bwd_fusew() && R > 0 && rq->cap >= 2 * int64_t(gridf)) {
      // the weight gradients of lin_k / lin_v accumulated inside the kernel: one pair of partials per workgroup, summed with the deferred sums
      int rc = TRAJSDE_OK;
      const int64_t base = rq->take(2 * int64_t(gridf), &rc);
      if (rc) return rc;
      float* wpart = wc->part + base * 4096;
      float* wcs = wc->cs + base * 64;                               // (the lin_v partials' column-sum slots carry the bias gradient)
      if (drop.p > 0.f) TS_LAUNCH_TAG("k_edge_attn_rows_bwd<8>", false, (k_edge_rows_bwd_mm<true, true>), gridf, 64 * RMMF_WAVES, rows_mmf_lds_bytes(), st, img, segptr, emb, q, agg, dagg, stats, R, DQ, RL, SS, DAGGM, EA, ED, drop, wpart, wcs);
      else TS_LAUNCH_TAG("k_edge_attn_rows_bwd<8>", false, (k_edge_rows_bwd_mm<false, true>), gridf, 64 * RMMF_WAVES, rows_mmf_lds_bytes(), st, img, segptr, emb, q, agg, dagg, stats, R, DQ, RL, SS, DAGGM, EA, ED, drop, wpart, wcs);
      rq->jobs.push_back(ReduceJob{wk, nullptr, base, gridf, 1, 64, 0, 0});
      rq->jobs.push_back(ReduceJob{wv, bv, base + gridf, gridf, 1, 64, 0, 0});      // bias = sum of the cs slots: d lin_v.bias
      *weights_done = true;
      return TRAJSDE_OK;
    }
    // one workgroup per CU (its LDS image fills most of one); fewer when the targets do not fill them
    const int grid = int(std::min<int64_t>(256, cdiv(R, RMM_WAVES)));
    if (drop.p > 0.f) TS_LAUNCH_TAG("k_edge_attn_rows_bwd<8>", false, (k_edge_rows_bwd_mm<true, false>), grid, 64 * RMM_WAVES, rows_mm_lds_bytes(), st, img, segptr, emb, q, agg, dagg, stats, R, DQ, RL, SS, DAGGM, EA, ED, drop, nw, nw);
    else TS_LAUNCH_TAG("k_edge_attn_rows_bwd<8>", false, (k_edge_rows_bwd_mm<false, false>), grid, 64 * RMM_WAVES, rows_mm_lds_bytes(), st, img, segptr, emb, q, agg, dagg, stats, R, DQ, RL, SS, DAGGM, EA, ED, drop, nw, nw);
  } else
    TS_LAUNCH_TAG("k_edge_attn_rows_bwd<8>", false, (k_gattn_bwd<8, false>), xcd_grid(cdiv(R, 4)), 256, 0, st, img, segptr, ns, emb, q, nf, nf, agg, dagg, stats,
                  R, DQ, nw, nw, RL, SS, DAGGM, EA, ED, nw, ns, drop);
  return TRAJSDE_OK;
}

// REV[e'] = index of the reverse edge (dst -> src) of e' = (src -> dst), or -1; *asym is raised when one is missing.
// The compacted rows list their senders in ascending order (prep.hip), so a binary search finds it; a row that is not sorted
// (a caller's own edge order) falls back to the scan.
__global__ void k_reverse_edges(const int32_t* __restrict__ segptr, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                int64_t E, int32_t* __restrict__ REV, int32_t* __restrict__ asym) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = src[e], jn = dst[e];
  const int r0 = segptr[i], r1 = segptr[i + 1];
  int lo = r0, hi = r1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (src[mid] < jn) lo = mid + 1;
    else hi = mid;
  }
  int found = lo < r1 && src[lo] == jn ? lo : -1;
  if (found < 0)
    for (int r = r0; r < r1; ++r)
      if (src[r] == jn) { found = r; break; }
  REV[e] = found;
  if (found < 0) atomicOr(asym, 1);
}

// gradients of the source rows, one wave per source node jn (lane = feature): for every incoming edge e' = (i -> jn) the
// outgoing edge (jn -> i) = REV[e'] carries the scalars:  d k_node[jn] += dlogit q[i],  d v_node[jn] += alpha dagg[i]
template <int HEADS>
__global__ __launch_bounds__(256) void k_gattn_src_bwd(const int32_t* __restrict__ segptr, const int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ REV, const float* __restrict__ EA,
                                                       const float* __restrict__ ED, const float* __restrict__ q,
                                                       const float* __restrict__ dagg, int64_t N, float* __restrict__ DKN,
                                                       float* __restrict__ DVN, const int32_t* __restrict__ asym) {
  const int lane = threadIdx.x & 63, h = lane / (64 / HEADS);
  const int64_t node = xcd_block() * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= N || *asym != 0) return;                     // asymmetric list: k_gattn_bwd scattered the rows with atomics
  float dk = 0.f, dv = 0.f;
  const int beg = segptr[node], end = segptr[node + 1];
  // 16 edges per round trip (the wave does nothing but wait for them: one target per wave, a handful of registers), their indices loaded
  // coalesced ONE ROUND AHEAD and used as scalars; same summation order as one edge at a time
  auto idx_at = [&](int e0, int& rv, int& sv) {
    const int ec = e0 + (lane & 15) < end ? e0 + (lane & 15) : end - 1;
    rv = REV[ec];
    sv = src[ec];
  };
  int rv_n = 0, sv_n = 0;
  if (beg < end) idx_at(beg, rv_n, sv_n);
  for (int e0 = beg; e0 < end; e0 += 16) {
    const int rv = rv_n, sv = sv_n;
    idx_at(e0 + 16 < end ? e0 + 16 : e0, rv_n, sv_n);
    float a[16], d[16], qv[16], gv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = __builtin_amdgcn_readlane(rv, u), i = __builtin_amdgcn_readlane(sv, u);
      a[u] = (EA + int64_t(e) * HEADS)[h];
      d[u] = (ED + int64_t(e) * HEADS)[h];
      qv[u] = (q + int64_t(i) * 64)[lane];
      gv[u] = (dagg + int64_t(i) * 64)[lane];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (e0 + u < end) {
        dk = fmaf(d[u], qv[u], dk);
        dv = fmaf(a[u], gv[u], dv);
      }
  }
  DKN[node * 64 + lane] = dk;
  DVN[node * 64 + lane] = dv;
}

// ---- workspace
struct AggBwdWs {
  // forward tape
  float *rel, *xn[8], *q[8], *kn[8], *vn[8], *agg[8], *x1[8], *xn2[8], *out[8];
  float* stats[8];               // softmax statistics (max logit, 1 / sum) per (target, head) of every layer
  // backward scratch
  float *dcur, *dnext, *dagg, *dxn, *DQ, *DKN, *DVN, *DREL, *RL, *SS, *DAGGM, *XF, *part, *cs, *varena;
  float *EA[8], *ED[8], *UZ[8];  // per layer: the per-edge (alpha d, d logit) scalars and per-target (U, Z) vectors (k_gattn_drel)
  int32_t *REV, *asym;
  NodeBlockScratch nb;
  EdgeEmbedScratch ee;
  int64_t total, parts;
  bool ok;
  AggBwdWs(int64_t N, int64_t E, int nl, int K, void* ws, int64_t bytes) {
    Carver c(ws, bytes);
    rel = c.take<float>(E * 64 + 64);
    for (int l = 0; l < nl; ++l) {
      xn[l] = c.take<float>(N * 64); q[l] = c.take<float>(N * 64); kn[l] = c.take<float>(N * 64); vn[l] = c.take<float>(N * 64);
      agg[l] = c.take<float>(N * 64); x1[l] = c.take<float>(N * 64); xn2[l] = c.take<float>(N * 64); out[l] = c.take<float>(N * 64);
      stats[l] = c.take<float>(N * 16);
    }
    float** singles[] = {&dcur, &dnext, &dagg, &dxn, &DQ, &DKN, &DVN, &DAGGM, &XF, &nb.dx1, &nb.UPD, &nb.DGP, &nb.DS};
    for (float** p : singles) *p = c.take<float>(N * 64);
    nb.H = c.take<float>(N * 256);
    nb.DH = c.take<float>(N * 256);
    RL = c.take<float>(N * 512);
    SS = c.take<float>(N * 512);
    DREL = c.take<float>(E * 64 + 64);
    for (int l = 0; l < nl; ++l) {
      EA[l] = c.take<float>(E * 8 + 64);
      ED[l] = c.take<float>(E * 8 + 64);
      UZ[l] = c.take<float>(N * 1024);
    }
    REV = c.take<int32_t>(E + 1);
    asym = c.take<int32_t>(4);
    float** edge[] = {&ee.DEP, &ee.DSP};
    for (float** p : edge) *p = c.take<float>(E * 64 + 64);
    ee.S = rel;                                             // the rel rows are dead once every layer's k_gattn_bwd has run
    nb.vpart = ee.vpart = c.take<float>(VPART_FLOATS);
    varena = c.take<float>(VPART_ARENA_SLABS * VPART_FLOATS);
    const int64_t rows = E > N ? E : N;
    parts = wgrad_max_parts(rows, 1);
    part = c.take<float>(parts * 4096);
    cs = c.take<float>(parts * 64);
    (void)K;
    total = c.off + 256;
    ok = c.ok;
  }
};

}  // namespace tsde

using namespace tsde;

// The aggregator's forward with one activation buffer per layer kept in `w` (the tape the backward walks): run once per
// training step, by trajsde_aggregator_forward_train or by the backward itself when no tape was handed over.
template <typename DropOf>
static int aggregator_tape(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, int nl, int num_heads,
                           const float* local_embed, AggBwdWs& w, DropOf drop_of, hipStream_t st) {
  const int64_t N = b->N, E = g->E_g, ntiles = (N + 15) / 16, etiles = (E + 15) / 16;
#if TSDE_SPLIT_H3
  if (E > 0 && rel_embed_fused())
    TS_LAUNCH_TAG("k_edge_embed<true>", false, k_edge_embed2, tile_grid((E + 31) / 32, 1024, edge_embed2_lds(1024)), 1024, edge_embed2_lds(1024), st,
                  blob_fwd + AggBlob::REL6G, g->g_geom, EdgeCount{E, nullptr, 0}, w.rel, 0);
  else
#endif
  if (E > 0)
    TS_LAUNCH(k_edge_embed<true>, tile_grid(etiles, 1024, EdgeL6::EMB_SIZE * 4), 1024, EdgeL6::EMB_SIZE * 4, st, blob_fwd + AggBlob::REL6,
              g->g_geom, EdgeCount{E, nullptr, 0}, w.rel, 0);
  const float* x = local_embed;
  for (int l = 0; l < nl; ++l) {
    const float* lb = blob_fwd + AggBlob::layer(l);
    TS_LAUNCH(k_node_proj<3>, tile_grid(ntiles, 512, NodeProjL<3>::SIZE * 4), 512, NodeProjL<3>::SIZE * 4, st, lb + AggLayerL::QKV, x, N,
              w.xn[l], w.q[l], w.kn[l], w.vn[l]);
    {
      const DropArg dl = drop_of(l);
      TS_REQUIRE(N < (1 << 23), "aggregator_forward_train: node rows are addressed with 32-bit byte offsets (N < 2^23)");
      if (num_heads == 8 && gattn_f32mm_enabled() && g->E_g > 0) {
        if (int rc = launch_global_attn_mf(lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l], N, w.agg[l], w.stats[l], dl, st)) return rc;
      } else
        TS_GLOBAL_ATTN(num_heads, false, dl, xcd_grid(cdiv(N, 4)), 256, 0, st, lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l], N, w.agg[l], w.stats[l]);
    }
    TS_LAUNCH(k_node_update<true>, tile_grid(ntiles, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, lb + AggLayerL::UPD6, w.agg[l], w.xn[l], x,
              N, w.x1[l], w.xn2[l], drop_of(l), no_merge());
    TS_LAUNCH(k_ffn6, tile_grid(ntiles, 512, FfnL6::HALF * 4), 512, FfnL6::HALF * 4, st, lb + AggLayerL::FFN6, w.x1[l], w.xn2[l], N, w.out[l],
              drop_of(l));
    x = w.out[l];
  }

  return TRAJSDE_OK;
}

extern "C" {

int64_t trajsde_aggregator_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_layers, int num_modes) {
  if (!b || !g || num_layers < 0 || num_layers > 8) return -1;
  AggBwdWs w(b->N, g->E_g, num_layers, num_modes, nullptr, 0);
  return w.total;
}

int trajsde_aggregator_backward(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, const float* blob_bwd,
                                int num_layers, int num_modes, const float* local_embed, const float* d_global, void* ws,
                                int64_t ws_bytes, float* const* grads, int n_grads, float* d_local, void* stream_) {
  return trajsde_aggregator_backward_heads(b, g, blob_fwd, blob_bwd, num_layers, num_modes, 8, local_embed, d_global, ws, ws_bytes, grads,
                                           n_grads, d_local, nullptr, 0, stream_);
}

int trajsde_aggregator_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, int num_layers, int num_modes,
                                     int num_heads, const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed,
                                     const trajsde_dropout* dropout, void* stream_) {
  TS_REQUIRE(b && g && blob_fwd && local_embed && ws && global_embed, "aggregator_forward_train: null pointer");
  TS_REQUIRE(num_heads == 8 || num_heads == 4, "aggregator_forward_train: num_heads must be 8 or 4");
  TS_REQUIRE(g->g_src && g->g_segptr, "aggregator_forward_train: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(g->exact, "aggregator_forward_train: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(num_layers >= 1 && num_layers <= 8 && num_modes > 0, "aggregator_forward_train: bad layer/mode count");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "aggregator_forward_train: dropout p must be in [0, 1)");
  TS_REQUIRE(!state_bf16(), "aggregator_forward_train: the training tape is fp32; switch trajsde_state_storage(0)");
  auto drop_of = [&](int layer) { return dropout ? make_drop(dropout->p, dropout->seed, 2 + layer) : no_drop(); };
  AggBwdWs w(b->N, g->E_g, num_layers, num_modes, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "aggregator_forward_train: workspace too small (trajsde_aggregator_backward_ws_bytes)");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = aggregator_tape(b, g, blob_fwd, num_layers, num_heads, local_embed, w, drop_of, st)) return rc;
  const int64_t N = b->N, ntiles = (N + 15) / 16;
  const int lds = (128 + MAT64 + 64) * 4;
  dim3 grid(tile_grid(ntiles, 512, lds), num_modes);
  TS_LAUNCH(k_mode_proj, grid, 512, lds, st, blob_fwd + AggBlob::norm(num_layers), blob_fwd + AggBlob::proj(num_layers, 0), w.out[num_layers - 1], N,
            global_embed);
  return TRAJSDE_OK;
}

int trajsde_aggregator_backward_heads(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, const float* blob_bwd,
                                      int num_layers, int num_modes, int num_heads, const float* local_embed, const float* d_global,
                                      void* ws, int64_t ws_bytes, float* const* grads, int n_grads, float* d_local,
                                      const trajsde_dropout* dropout, int tape_valid, void* stream_) {
  TS_REQUIRE(b && g && blob_fwd && blob_bwd && local_embed && d_global && ws && grads && d_local, "aggregator_backward: null pointer");
  TS_REQUIRE(!state_bf16(), "aggregator_backward: the backward pass keeps its tape in fp32; switch trajsde_state_storage(0) for training");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "aggregator_backward: dropout p must be in [0, 1)");
  auto drop_of = [&](int layer) { return dropout ? make_drop(dropout->p, dropout->seed, 2 + layer) : no_drop(); };   // dropout.hpp block ids
  TS_REQUIRE(num_heads == 8 || num_heads == 4, "aggregator_backward: num_heads must be 8 or 4");
  TS_REQUIRE(g->g_src && g->g_segptr, "aggregator_backward: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(g->exact, "aggregator_backward: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(num_layers >= 1 && num_layers <= 8 && num_modes > 0, "aggregator_backward: bad layer/mode count");
  const std::vector<std::string> names = stage_param_names(TRAJSDE_STAGE_AGGREGATOR_BWD, num_layers, num_modes);
  TS_REQUIRE(n_grads == int(names.size()), "aggregator_backward: gradient count does not match trajsde_param_count(AGGREGATOR_BWD)");
  std::unordered_map<std::string, float*> slot;
  for (int i = 0; i < n_grads; ++i) {
    TS_REQUIRE(grads[i] != nullptr, "aggregator_backward: null gradient buffer " + names[i]);
    slot[names[i]] = grads[i];
  }
  bool missing = false;
  auto G = [&](const std::string& n) -> float* {
    auto it = slot.find(n);
    if (it == slot.end()) { missing = true; return nullptr; }
    return it->second;
  };
  AggBwdWs w(b->N, g->E_g, num_layers, num_modes, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "aggregator_backward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int64_t N = b->N, E = g->E_g, ntiles = (N + 15) / 16, etiles = (E + 15) / 16;
  const int nl = num_layers, K = num_modes;
  const WgradCtx wc{st, w.part, w.cs, nullptr, w.parts};
  DeferredSums sums(st, w.part, w.cs, w.parts, nullptr, w.varena, VPART_ARENA_SLABS * VPART_FLOATS);    // this call's reductions: at the end

  if (!tape_valid)
    if (int rc = aggregator_tape(b, g, blob_fwd, nl, num_heads, local_embed, w, drop_of, st)) return rc;
  (void)etiles;

  // ---- multihead_proj + final norm
  if (K > 0)                                                    // d xn = sum_k W_k^T d global_k: one launch, the sum in registers
    TS_LAUNCH(k_lin_t_sum, unsigned(ntiles > 0 ? (ntiles + 3) / 4 : 1), 256, 2 * MAT64 * 4, st, blob_bwd + AggBwdBlob::proj(nl, 0),
              int64_t(AggBwdBlob::proj(nl, 1) - AggBwdBlob::proj(nl, 0)), d_global, N * 64, K, N, w.dxn);
  {
    const int gp = vec_grid(ntiles, 256, ProjBwdL<0>::SIZE * 4);
    float* const vp = vpart_slab(w.nb.vpart, int64_t(gp) * 4, 128);
    TS_LAUNCH(k_node_proj_bwd<0>, gp, 256, ProjBwdL<0>::SIZE * 4, st, blob_bwd + AggBwdBlob::norm(nl), w.out[nl - 1], nullptr, w.dxn,
              nullptr, nullptr, nullptr, N, w.dcur, w.XF, vp);
    {
      ColsumBatch cb(st, gp * 4, 128);
      cb.add(vp, 64, G("norm.weight"));
      cb.add(vp + 64, 64, G("norm.bias"));
      if (int rc = cb.flush()) return rc;
    }
    float* pw = G("multihead_proj.weight");
    float* pb = G("multihead_proj.bias");
    TS_REQUIRE(!missing, "aggregator_backward: parameter table lacks norm / multihead_proj");
    WgradBatch wb(wc, N, N);
    for (int k = 0; k < K; ++k)
      if (int rc = wb.add(d_global + int64_t(k) * N * 64, 64, w.XF, 64, pw + int64_t(k) * MAT64, 64, 0, pb + 64 * k, 0)) return rc;
    if (int rc = wb.flush()) return rc;
  }

  // ---- layers, last to first
  // the global graph of a scene is symmetric (all ordered pairs of its valid actors), which lets the source-row
  // gradients be gathered in a fixed order instead of scattered with atomics; checked on the device (w.asym), no read back
  TS_HIP(hipMemsetAsync(w.asym, 0, sizeof(int32_t), st));
  if (E > 0) TS_LAUNCH(k_reverse_edges, cdiv(E, 256), 256, 0, st, g->g_segptr, g->g_src, g->g_dst, E, w.REV, w.asym);
  float* dcur = w.dcur;
  float* dnext = w.dnext;
  for (int l = nl - 1; l >= 0; --l) {
    const std::string p = "global_interactor_layers." + std::to_string(l);
    const float* lb = blob_bwd + AggBwdBlob::layer(l);
    const float* x_in = l == 0 ? local_embed : w.out[l - 1];
    NodeBlockGrads gr{G(p + ".lin_ih.weight"), G(p + ".lin_ih.bias"), G(p + ".lin_hh.weight"), G(p + ".lin_hh.bias"),
                      G(p + ".lin_self.weight"), G(p + ".lin_self.bias"), G(p + ".out_proj.weight"), G(p + ".out_proj.bias"),
                      G(p + ".norm2.weight"), G(p + ".norm2.bias"), G(p + ".mlp.0.weight"), G(p + ".mlp.0.bias"),
                      G(p + ".mlp.3.weight"), G(p + ".mlp.3.bias")};
    float* wke = G(p + ".lin_k_edge.weight");
    float* bke = G(p + ".lin_k_edge.bias");
    float* wve = G(p + ".lin_v_edge.weight");
    float* bve = G(p + ".lin_v_edge.bias");
    float* n1g = G(p + ".norm1.weight");
    float* n1b = G(p + ".norm1.bias");
    float* qkv_w[3] = {G(p + ".lin_q_node.weight"), G(p + ".lin_k_node.weight"), G(p + ".lin_v_node.weight")};
    float* qkv_b[3] = {G(p + ".lin_q_node.bias"), G(p + ".lin_k_node.bias"), G(p + ".lin_v_node.bias")};
    TS_REQUIRE(!missing, "aggregator_backward: parameter table lacks an entry of " + p);
    const NodeBlockTape tp{w.agg[l], w.xn[l], w.x1[l], w.xn2[l]};
    WgradBatch wb(wc, N, N);                                       // the layer's seven N-row problems in one launch (gated update + q, k, v)
    if (int rc = node_block_backward(lb + AggLayerBwdL::NODE, tp, dcur, N, w.nb, wc, gr, w.dagg, w.dxn, st, drop_of(l), &wb)) return rc;
    // (carved one behind the other: one fill when nothing sits between them)
    if (w.DVN >= w.DKN + N * 64 && w.DVN - w.DKN < N * 64 + 1024) {
      TS_HIP(hipMemsetAsync(w.DKN, 0, size_t(w.DVN - w.DKN + N * 64) * sizeof(float), st));
    } else {
      TS_HIP(hipMemsetAsync(w.DKN, 0, size_t(N) * 64 * sizeof(float), st));
      TS_HIP(hipMemsetAsync(w.DVN, 0, size_t(N) * 64 * sizeof(float), st));
    }
    if (num_heads == 4) {
      TS_LAUNCH((k_gattn_bwd<4, true>), xcd_grid(cdiv(N, 4)), 256, 0, st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l],
                w.agg[l], w.dagg, w.stats[l], N, w.DQ, w.DKN, w.DVN, w.RL, w.SS, w.DAGGM, w.EA[l], w.ED[l], w.UZ[l], w.asym, drop_of(l));
      TS_LAUNCH(k_gattn_src_bwd<4>, xcd_grid(cdiv(N, 4)), 256, 0, st, g->g_segptr, g->g_src, w.REV, w.EA[l], w.ED[l], w.q[l], w.dagg, N, w.DKN, w.DVN,
                w.asym);
    } else {
      if (rows_bwd_mm()) {                                           // the matrix-core form (TRAJSDE_ROWS_BWD_MM=0: the vector form)
        const int grid_mm = xcd_grid(std::min<int64_t>(256, cdiv(N, GBM_WAVES)));
        const DropArg dl = drop_of(l);
        if (dl.p > 0.f)
          TS_LAUNCH_TAG("(k_gattn_bwd<8, true>)", false, (k_gattn_bwd_mm<true>), grid_mm, 64 * GBM_WAVES, gattn_bwd_mm_lds_bytes(), st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l],
                        w.vn[l], w.agg[l], w.dagg, w.stats[l], N, w.DQ, w.DKN, w.DVN, w.RL, w.SS, w.DAGGM, w.EA[l], w.ED[l], w.UZ[l], w.asym, dl);
        else
          TS_LAUNCH_TAG("(k_gattn_bwd<8, true>)", false, (k_gattn_bwd_mm<false>), grid_mm, 64 * GBM_WAVES, gattn_bwd_mm_lds_bytes(), st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l],
                        w.vn[l], w.agg[l], w.dagg, w.stats[l], N, w.DQ, w.DKN, w.DVN, w.RL, w.SS, w.DAGGM, w.EA[l], w.ED[l], w.UZ[l], w.asym, dl);
      } else
      TS_LAUNCH((k_gattn_bwd<8, true>), xcd_grid(cdiv(N, 4)), 256, 0, st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l],
                w.agg[l], w.dagg, w.stats[l], N, w.DQ, w.DKN, w.DVN, w.RL, w.SS, w.DAGGM, w.EA[l], w.ED[l], w.UZ[l], w.asym, drop_of(l));
      TS_LAUNCH(k_gattn_src_bwd<8>, xcd_grid(cdiv(N, 4)), 256, 0, st, g->g_segptr, g->g_src, w.REV, w.EA[l], w.ED[l], w.q[l], w.dagg, N, w.DKN, w.DVN,
                w.asym);
    }
    if (int rc = run_headwise_outer(wc, w.q[l], w.RL, N, wke, num_heads)) return rc;
    if (int rc = run_headwise_outer(wc, w.dagg, w.SS, N, wve, num_heads)) return rc;
    TS_HIP(hipMemsetAsync(bke, 0, 64 * sizeof(float), st));          // a key bias shifts every logit of a target alike
    if (int rc = run_colsum_tall(st, w.DAGGM, N, 64, 64, bve, w.nb.vpart)) return rc;    // (one workgroup over N rows was 30 us at N = 8 192)
    const int gp = vec_grid(ntiles, 256, ProjBwdL<3>::SIZE * 4);
    float* const vp = vpart_slab(w.nb.vpart, int64_t(gp) * 4, 128);
    TS_LAUNCH(k_node_proj_bwd<3>, gp, 256, ProjBwdL<3>::SIZE * 4, st, lb + AggLayerBwdL::PROJ, x_in, w.nb.dx1, w.dxn, w.DQ, w.DKN, w.DVN, N,
              dnext, nullptr, vp);
    {
      ColsumBatch cb(st, gp * 4, 128);
      cb.add(vp, 64, n1g);
      cb.add(vp + 64, 64, n1b);
      if (int rc = cb.flush()) return rc;
    }
    const float* dps[3] = {w.DQ, w.DKN, w.DVN};
    for (int j = 0; j < 3; ++j)
      if (int rc = wb.add(dps[j], 64, w.xn[l], 64, qkv_w[j], 64, 0, qkv_b[j], 0)) return rc;
    if (int rc = wb.flush()) return rc;
    float* t = dcur; dcur = dnext; dnext = t;
  }
  TS_HIP(hipMemcpyAsync(d_local, dcur, size_t(N) * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));

  // ---- rel_embed (shared by every layer: DREL is the sum of their d rel rows, assembled here in one pass)
  if (E > 0)
    for (int l0 = 0; l0 < nl; l0 += 4) {
      const int cnt = nl - l0 < 4 ? nl - l0 : 4;
      DrelArgs da{};
      for (int i = 0; i < 4; ++i) {
        const int l = l0 + (i < cnt ? i : 0);
        da.EA[i] = w.EA[l]; da.ED[i] = w.ED[l]; da.UZ[i] = w.UZ[l];
      }
      if (int rc = run_gattn_drel(st, num_heads, cnt, false, da, g->g_segptr, N, w.DREL, l0 > 0 ? 1 : 0)) return rc;
    }
  {
    const std::string p = "rel_embed";
    EdgeEmbedGrads eg{G(p + ".module_list.0.0.weight"), G(p + ".module_list.0.0.bias"), G(p + ".module_list.0.1.weight"),
                      G(p + ".module_list.0.1.bias"), G(p + ".module_list.1.0.weight"), G(p + ".module_list.1.0.bias"),
                      G(p + ".module_list.1.1.weight"), G(p + ".module_list.1.1.bias"), G(p + ".module_list.0.3.weight"),
                      G(p + ".module_list.0.3.bias"), G(p + ".module_list.1.3.weight"), G(p + ".module_list.1.3.bias"),
                      G(p + ".aggr_embed.0.weight"), G(p + ".aggr_embed.0.bias"), G(p + ".aggr_embed.2.weight"),
                      G(p + ".aggr_embed.2.bias"), G(p + ".aggr_embed.3.weight"), G(p + ".aggr_embed.3.bias")};
    TS_REQUIRE(!missing, "aggregator_backward: parameter table lacks an entry of rel_embed");
    // E == 0: the embedding never ran and its gradients stay as the caller initialised them (zeros)
    if (int rc = edge_embed_backward(blob_bwd + AggBwdBlob::REL, g->g_geom, w.DREL, E, w.ee, wc, eg, st)) return rc;
  }
  return sums.finish();
}

}  // extern "C"
