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
// statistics recomputed in a first pass over the segment.  With alpha the attention weights,
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

template <int HEADS>
__global__ __launch_bounds__(256) void k_gattn_bwd(const float* __restrict__ img, const int32_t* __restrict__ segptr,
                                                   const int32_t* __restrict__ src, const float* __restrict__ rel,
                                                   const float* __restrict__ q, const float* __restrict__ kn,
                                                   const float* __restrict__ vn, const float* __restrict__ agg,
                                                   const float* __restrict__ dagg, int64_t N, float* __restrict__ DQ,
                                                   float* __restrict__ DKN, float* __restrict__ DVN, float* __restrict__ DREL,
                                                   float* __restrict__ RL, float* __restrict__ SS, float* __restrict__ DAGGM,
                                                   float* __restrict__ EA, float* __restrict__ ED, DropArg drop) {
  // EA / ED non-null: the source-row gradients are gathered afterwards by k_gattn_src_bwd from the per-edge (alpha, dlogit)
  // scalars written here (symmetric graph, fixed summation order); null: scattered right here with float atomics
  constexpr int LPH = 64 / HEADS, SL = 64 / LPH, NV = SL / 4;     // as in k_global_attn<HEADS>
  constexpr float INV = HEADS == 4 ? 0.25f : INV_SQRT_DH;
  __shared__ __attribute__((aligned(16))) float sbuf[4][HEADS][64 + 4];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row addresses in SGPRs
  const int h = lane / LPH, j = lane % LPH;
  const int64_t node = int64_t(blockIdx.x) * 4 + wv;
  const int64_t nc = node < N ? node : N - 1;
  const float* wke = img + GAttnL::WKE;
  const float* wve = img + GAttnL::WVE;
  const float ql = q[nc * 64 + lane];
  const float da = node < N ? dagg[nc * 64 + lane] : 0.f;
  const float cb = head_sum_n<HEADS>(ql * img[GAttnL::BKE + lane]);
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
  const int beg = segptr[nc], end = node < N ? segptr[nc + 1] : beg;
  auto load_rel = [&](int e, f4 (&r)[NV]) {
    const float* rrow = rel + int64_t(e) * 64;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) r[v4] = *reinterpret_cast<const f4*>(rrow + SL * j + 4 * v4);
  };
  // the source indices of a chunk in one coalesced load, handed out as scalars (k_global_attn does the same)
  auto chunk_src = [&](int e0, int n) { return src[e0 + (lane & (n - 1)) < end ? e0 + (lane & (n - 1)) : end - 1]; };
  auto logit = [&](const f4 (&r)[NV], float knv) {
    float p = ql * knv;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
      for (int e = 0; e < 4; ++e) p = fmaf(r[v4][e], U[4 * v4 + e], p);
    return (head_sum_n<HEADS>(p) + cb) * INV;
  };
  // pass 1: softmax statistics of the segment
  float m = -INFINITY, s = 0.f;
  for (int e0 = beg; e0 < end; e0 += 8) {                  // 8 edges in flight per round trip, as in the forward kernel
    f4 r[8][NV];
    float knv[8], lg[8];
    const int sv = chunk_src(e0, 8);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u < end ? e0 + u : end - 1;
      load_rel(e, r[u]);
      knv[u] = (kn + int64_t(__builtin_amdgcn_readlane(sv, u)) * 64)[lane];
    }
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lg[u] = e0 + u < end ? logit(r[u], knv[u]) : -INFINITY;
      cm = fmaxf(cm, lg[u]);
    }
    const float mn = fmaxf(m, cm);
    s *= fast_exp(m - mn);
    m = mn;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += fast_exp(lg[u] - m);
  }
  const float inv = 1.0f / (s + 1e-16f);
  // pass 2: gradients.  Attention dropout (AGG:116): agg_h = sum_e alpha_e d_e (v_node[src] + lin_v_edge(rel_e) )_h with
  // d_e = keep_e / (1 - p): every "alpha" that multiplies a VALUE below becomes alpha d_e, d alpha_e = d_e (dagg . value_e),
  // and lin_v_edge.bias sees sum_e alpha_e d_e instead of 1
  const bool dropping = drop.p > 0.f;
  float sad = 0.f;
  float dqe = 0.f, Rl[SL], Sa[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) Rl[e] = Sa[e] = 0.f;
  for (int e0 = beg; e0 < end; e0 += 4) {                  // loads of 4 edges in flight, then their gradients in order
   f4 rr4[4][NV];
   float kn4[4], vn4[4];
   int sx4[4];
   float kp[4] = {1.f, 1.f, 1.f, 1.f};
   if (dropping) drop_attn_chunk<4>(kp, drop, uint32_t(nc), uint32_t(e0 - beg), lane, h);
   const int sv = chunk_src(e0, 4);
#pragma unroll
   for (int u = 0; u < 4; ++u) {
     const int e = e0 + u < end ? e0 + u : end - 1;
     sx4[u] = __builtin_amdgcn_readlane(sv, u);
     load_rel(e, rr4[u]);
     kn4[u] = (kn + int64_t(sx4[u]) * 64)[lane];
     vn4[u] = (vn + int64_t(sx4[u]) * 64)[lane];
   }
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int e = e0 + u;
    if (e >= end) break;
    const int sidx = sx4[u];
    const f4 (&r)[NV] = rr4[u];
    const float knv = kn4[u], vnv = vn4[u];
    const float alpha = fast_exp(logit(r, knv) - m) * inv;
    const float alk = alpha * kp[u];                       // the weight the values were summed with
    sad += alk;
    float t = da * vnv;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
      for (int c = 0; c < 4; ++c) t = fmaf(r[v4][c], Z[4 * v4 + c], t);
    const float dal = (head_sum_n<HEADS>(t) + cz) * kp[u];
    const float dls = alpha * (dal - dlt) * INV;
    dqe = fmaf(dls, knv, dqe);
    if (EA != nullptr) {
      if (j == 0) {
        EA[int64_t(e) * HEADS + h] = alk;
        ED[int64_t(e) * HEADS + h] = dls;
      }
    } else {
      atomicAdd(DKN + int64_t(sidx) * 64 + lane, dls * ql);
      atomicAdd(DVN + int64_t(sidx) * 64 + lane, alk * da);
    }
    float dr[SL];
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = 4 * v4 + c;
        Rl[i] = fmaf(dls, r[v4][c], Rl[i]);
        Sa[i] = fmaf(alk, r[v4][c], Sa[i]);
        dr[i] = fmaf(dls, U[i], alk * Z[i]);
      }
    // d rel_e = sum over heads: lanes j, j + LPH, ... hold the same SL columns
#pragma unroll
    for (int c = 0; c < SL; ++c) {
      float x = dr[c];
      if (LPH == 8) {                                       // lane ^ 8 inside the 16-lane row: DPP row_ror:8
        const int rr = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, true);
        x += __int_as_float(rr);
      }
      dr[c] = xor32_sum(xor16_sum(x));                      // lanes ^ 16, ^ 32: permlane swaps, no LDS crossbar
    }
    if (h == 0) {
      float* p = DREL + int64_t(e) * 64 + SL * j;
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4) {
        f4 a = *reinterpret_cast<f4*>(p + 4 * v4);
        a += f4{dr[4 * v4], dr[4 * v4 + 1], dr[4 * v4 + 2], dr[4 * v4 + 3]};
        *reinterpret_cast<f4*>(p + 4 * v4) = a;
      }
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
    DAGGM[node * 64 + lane] = da * (dropping ? sad : s * inv);        // lin_v_edge.bias sees sum_e alpha d_e (= 1 without dropout)
  }
}

// REV[e'] = index of the reverse edge (dst -> src) of e' = (src -> dst), or -1; *asym is raised when one is missing
__global__ void k_reverse_edges(const int32_t* __restrict__ segptr, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                int64_t E, int32_t* __restrict__ REV, int32_t* __restrict__ asym) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = src[e], jn = dst[e];
  int found = -1;
  for (int r = segptr[i]; r < segptr[i + 1]; ++r)
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
                                                       float* __restrict__ DVN) {
  const int lane = threadIdx.x & 63, h = lane / (64 / HEADS);
  const int64_t node = int64_t(blockIdx.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (node >= N) return;
  float dk = 0.f, dv = 0.f;
  const int beg = segptr[node], end = segptr[node + 1];
  for (int e0 = beg; e0 < end; e0 += 8) {                  // 8 edges per round trip; indices loaded coalesced, used as scalars
    const int ec = e0 + (lane & 7) < end ? e0 + (lane & 7) : end - 1;
    const int rv = REV[ec], sv = src[ec];
    float a[8], d[8], qv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = __builtin_amdgcn_readlane(rv, u), i = __builtin_amdgcn_readlane(sv, u);
      a[u] = (EA + int64_t(e) * HEADS)[h];
      d[u] = (ED + int64_t(e) * HEADS)[h];
      qv[u] = (q + int64_t(i) * 64)[lane];
      gv[u] = (dagg + int64_t(i) * 64)[lane];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u < end) {                                  // same summation order as one edge at a time
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
  // backward scratch
  float *dcur, *dnext, *dagg, *dxn, *DQ, *DKN, *DVN, *DREL, *RL, *SS, *DAGGM, *XF, *part, *cs, *EA, *ED;
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
    }
    float** singles[] = {&dcur, &dnext, &dagg, &dxn, &DQ, &DKN, &DVN, &DAGGM, &XF, &nb.dx1, &nb.UPD, &nb.DGP, &nb.DS};
    for (float** p : singles) *p = c.take<float>(N * 64);
    nb.H = c.take<float>(N * 256);
    nb.DH = c.take<float>(N * 256);
    RL = c.take<float>(N * 512);
    SS = c.take<float>(N * 512);
    DREL = c.take<float>(E * 64 + 64);
    EA = c.take<float>(E * 8 + 8);
    ED = c.take<float>(E * 8 + 8);
    REV = c.take<int32_t>(E + 1);
    asym = c.take<int32_t>(4);
    float** edge[] = {&ee.S, &ee.DEP, &ee.DSP};
    for (float** p : edge) *p = c.take<float>(E * 64 + 64);
    nb.vpart = ee.vpart = c.take<float>(VPART_FLOATS);
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
      TS_GLOBAL_ATTN(num_heads, false, dl, cdiv(N, 4), 256, 0, st, lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l], N, w.agg[l]);
    }
    TS_LAUNCH(k_node_update<true>, tile_grid(ntiles, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, lb + AggLayerL::UPD6, w.agg[l], w.xn[l], x,
              N, w.x1[l], w.xn2[l], drop_of(l));
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

  if (!tape_valid)
    if (int rc = aggregator_tape(b, g, blob_fwd, nl, num_heads, local_embed, w, drop_of, st)) return rc;
  (void)etiles;

  // ---- multihead_proj + final norm
  for (int k = 0; k < K; ++k)
    TS_LAUNCH(k_lin_t_acc, tile_grid(ntiles, 256, MAT64 * 4), 256, MAT64 * 4, st, blob_bwd + AggBwdBlob::proj(nl, k),
              d_global + int64_t(k) * N * 64, N, w.dxn, k > 0 ? 1 : 0);
  {
    const int gp = vec_grid(ntiles, 256, ProjBwdL<0>::SIZE * 4);
    TS_LAUNCH(k_node_proj_bwd<0>, gp, 256, ProjBwdL<0>::SIZE * 4, st, blob_bwd + AggBwdBlob::norm(nl), w.out[nl - 1], nullptr, w.dxn,
              nullptr, nullptr, nullptr, N, w.dcur, w.XF, w.nb.vpart);
    if (int rc = run_colsum(st, w.nb.vpart, gp * 4, 128, 64, G("norm.weight"))) return rc;
    if (int rc = run_colsum(st, w.nb.vpart + 64, gp * 4, 128, 64, G("norm.bias"))) return rc;
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
  // gradients be gathered in a fixed order instead of scattered with atomics; checked here, one 4-byte read back
  bool symmetric = false;
  if (E > 0) {
    TS_HIP(hipMemsetAsync(w.asym, 0, sizeof(int32_t), st));
    TS_LAUNCH(k_reverse_edges, cdiv(E, 256), 256, 0, st, g->g_segptr, g->g_src, g->g_dst, E, w.REV, w.asym);
    int32_t flag = 1;
    TS_HIP(hipMemcpyAsync(&flag, w.asym, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TS_HIP(hipStreamSynchronize(st));
    symmetric = flag == 0;
  }
  float* const ea = symmetric ? w.EA : nullptr;
  float* const ed = symmetric ? w.ED : nullptr;
  TS_HIP(hipMemsetAsync(w.DREL, 0, size_t(E * 64 + 64) * sizeof(float), st));
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
    if (int rc = node_block_backward(lb + AggLayerBwdL::NODE, tp, dcur, N, w.nb, wc, gr, w.dagg, w.dxn, st, drop_of(l))) return rc;
    TS_HIP(hipMemsetAsync(w.DKN, 0, size_t(N) * 64 * sizeof(float), st));
    TS_HIP(hipMemsetAsync(w.DVN, 0, size_t(N) * 64 * sizeof(float), st));
    if (num_heads == 4)
      TS_LAUNCH(k_gattn_bwd<4>, cdiv(N, 4), 256, 0, st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l],
                w.agg[l], w.dagg, N, w.DQ, w.DKN, w.DVN, w.DREL, w.RL, w.SS, w.DAGGM, ea, ed, drop_of(l));
    else
      TS_LAUNCH(k_gattn_bwd<8>, cdiv(N, 4), 256, 0, st, lb + AggLayerBwdL::ATTN, g->g_segptr, g->g_src, w.rel, w.q[l], w.kn[l], w.vn[l],
                w.agg[l], w.dagg, N, w.DQ, w.DKN, w.DVN, w.DREL, w.RL, w.SS, w.DAGGM, ea, ed, drop_of(l));
    if (symmetric) {
      if (num_heads == 4)
        TS_LAUNCH(k_gattn_src_bwd<4>, cdiv(N, 4), 256, 0, st, g->g_segptr, g->g_src, w.REV, w.EA, w.ED, w.q[l], w.dagg, N, w.DKN, w.DVN);
      else
        TS_LAUNCH(k_gattn_src_bwd<8>, cdiv(N, 4), 256, 0, st, g->g_segptr, g->g_src, w.REV, w.EA, w.ED, w.q[l], w.dagg, N, w.DKN, w.DVN);
    }
    if (int rc = run_headwise_outer(wc, w.q[l], w.RL, N, wke, num_heads)) return rc;
    if (int rc = run_headwise_outer(wc, w.dagg, w.SS, N, wve, num_heads)) return rc;
    TS_HIP(hipMemsetAsync(bke, 0, 64 * sizeof(float), st));          // a key bias shifts every logit of a target alike
    if (int rc = run_colsum(st, w.DAGGM, N, 64, 64, bve)) return rc;
    const int gp = vec_grid(ntiles, 256, ProjBwdL<3>::SIZE * 4);
    TS_LAUNCH(k_node_proj_bwd<3>, gp, 256, ProjBwdL<3>::SIZE * 4, st, lb + AggLayerBwdL::PROJ, x_in, w.nb.dx1, w.dxn, w.DQ, w.DKN, w.DVN, N,
              dnext, nullptr, w.nb.vpart);
    if (int rc = run_colsum(st, w.nb.vpart, gp * 4, 128, 64, n1g)) return rc;
    if (int rc = run_colsum(st, w.nb.vpart + 64, gp * 4, 128, 64, n1b)) return rc;
    const float* dps[3] = {w.DQ, w.DKN, w.DVN};
    WgradBatch wb(wc, N, N);
    for (int j = 0; j < 3; ++j)
      if (int rc = wb.add(dps[j], 64, w.xn[l], 64, qkv_w[j], 64, 0, qkv_b[j], 0)) return rc;
    if (int rc = wb.flush()) return rc;
    float* t = dcur; dcur = dnext; dnext = t;
  }
  TS_HIP(hipMemcpyAsync(d_local, dcur, size_t(N) * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));

  // ---- rel_embed (shared by every layer: DREL is the sum of their d rel rows)
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
  return TRAJSDE_OK;
}

}  // extern "C"
