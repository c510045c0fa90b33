// node_bwd.hip -- backward of the node-level blocks every attention family of the path shares (AAEncoder ENC:595-613,
// ALEncoder ENC:778-797, GlobalInteractorLayer AGG:119-135), plus the MultipleInputEmbedding backward (EMB:43-70):
//
//   k_ffn_bwd_a / k_ffn_bwd_b   out = x1 + W2 relu(W1 norm2(x1) + b1) + b2      -> dx1, saves (h, dh)
//   k_upd_bwd                   gate / lin_self / out_proj (+ residual)          -> dagg, dxn, saves (upd, dgate_pre, ds)
//   k_node_proj_bwd<NQ>         xn = norm1(x), NQ projections of xn              -> dx
//   k_lin_t_acc                 out (+)= W^T d   for one transposed image
//   k_edge_embed_bwd_tail / _branch   MultipleInputEmbedding over edge rows      -> saves the (delta, input) rows
//   k_headwise_outer            W[d][c] = sum_i X[i][d] Y[i][head(d)][c]
//
// Matrix products: tile.hpp linear_acc on forward images (recomputation) and linear_adj on transposed images (adjoints:
// row-scaled split precision in the fp16x3 build, exact fp32 in the bf16x6 build); parameter gradients
// that are matrices come from the saved rows through run_wgrad (decoder_bwd.hip), vectors from per-wave accumulators.
#include "bwd.hpp"
#include "common.hpp"
#include "dropout.hpp"
#include "layouts.hpp"
#include "tile.hpp"
#include "stamps.hpp"
#include "tile_bwd.hpp"

TSDE_STAMP_TABLE(tail, 8)

namespace tsde {

__device__ __forceinline__ void load_row256(f4 (&a)[16], const float* base, int64_t row, int g) {
  const float* p = base + row * 256 + 4 * g;
#pragma unroll
  for (int jt = 0; jt < 16; ++jt) a[jt] = *reinterpret_cast<const f4*>(p + 16 * jt);
}
__device__ __forceinline__ void store_row256(const f4 (&a)[16], float* base, int64_t row, int g) {
  float* p = base + row * 256 + 4 * g;
#pragma unroll
  for (int jt = 0; jt < 16; ++jt) *reinterpret_cast<f4*>(p + 16 * jt) = a[jt];
}

// H = relu(W1 xn2 + b1)  [R,256];  DH = (W2^T dout) * (H > 0)
// with dropout (mlp: Linear - ReLU - Dropout(m1) - Linear - Dropout(m2)): H = relu(..) m1 is what mlp.3 saw, the gradient
// entering mlp.3 is DOUT2 = dout m2 (saved: it is the delta of the mlp.3 weight gradient), DH = (W2^T DOUT2) m1 (pre > 0)
__global__ __launch_bounds__(256) void k_ffn_bwd_a(const float* __restrict__ img, const float* __restrict__ dout,
                                                   const float* __restrict__ xn2, int64_t R, float* __restrict__ H,
                                                   float* __restrict__ DH, float* __restrict__ DOUT2, DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, FfnBwdAL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 n[4], hid[16], dh[16], dn[4];
    load_row(n, xn2, r, L.g);
    load_row(dn, dout, r, L.g);                                  // requested with the tile's first row, not after the first product
    linear<16, 4>(hid, n, lds + FfnBwdAL::W1, lds + FfnBwdAL::B1, L);
    relu<16>(hid);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) n[jt] = dn[jt];
    const bool dropping = drop.p > 0.f;
    f4 m1[16];
    if (dropping) {
      f4 mk[4];
      drop_feat16(mk, drop, DK_OUT, uint32_t(r), 0, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) n[jt] *= mk[jt];
      if (row < R) store_row(n, DOUT2, row, L.g);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        drop_feat16(mk, drop, DK_HIDDEN, uint32_t(r), blk, L.g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          m1[4 * blk + jt] = mk[jt];
          hid[4 * blk + jt] *= mk[jt];
        }
      }
    }
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) dh[jt] = f4{0.f, 0.f, 0.f, 0.f};
    linear_adj<16, 4>(dh, n, lds + FfnBwdAL::W2T, L);
#pragma unroll
    for (int jt = 0; jt < 16; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!(hid[jt][c] > 0.f)) dh[jt][c] = 0.f;              // pre <= 0, or dropped
        else if (dropping) dh[jt][c] *= m1[jt][c];
      }
    if (row < R) {
      store_row256(hid, H, row, L.g);
      store_row256(dh, DH, row, L.g);
    }
  }
}

// dx1 = dout + norm2-backward(W1^T dh);  per-wave (dgamma2 | dbeta2) -> vpart[wave][128]
__global__ __launch_bounds__(256) void k_ffn_bwd_b(const float* __restrict__ img, const float* __restrict__ DH,
                                                   const float* __restrict__ dout, const float* __restrict__ x1, int64_t R,
                                                   float* __restrict__ dx1, float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, FfnBwdBL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  f4 dgam[4], dbet[4];
  zero4(dgam); zero4(dbet);
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 dh[16], t[4], x[4], dres[4];
    load_row256(dh, DH, r, L.g);
    load_row(x, x1, r, L.g);                                     // all three rows of the tile are requested up front
    load_row(dres, dout, r, L.g);
    zero4(t);
    linear_adj<4, 16>(t, dh, lds + FfnBwdBL::W1T, L);
    if (row >= R) zero4(t);
    const float rstd = ln_normalize(x);
    ln_backward(t, x, rstd, lds + FfnBwdBL::N2G, L.g, dgam, dbet);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) t[jt] += dres[jt];
    if (row < R) store_row(t, dx1, row, L.g);
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 128;
  flush_vec(dgam, vp, L);
  flush_vec(dbet, vp + 64, L);
}

// gate = sigmoid(Wih agg + Whh xn + b); s = Wself xn + b; upd = agg + gate (s - agg); x1 = x + Wout upd + b
// given dx1:  UPD, DGP (d gate pre-activation), DS (d s), DAGG (d agg), DXN (d xn from this block)
// with dropout x1 = x + m3 (Wout upd + b): the gradient entering out_proj is DX1M = dx1 m3 (saved: delta of its weight gradient)
__global__ __launch_bounds__(256) void k_upd_bwd(const float* __restrict__ img, const float* __restrict__ dx1,
                                                 const float* __restrict__ agg, const float* __restrict__ xn, int64_t R,
                                                 float* __restrict__ UPD, float* __restrict__ DGP, float* __restrict__ DS,
                                                 float* __restrict__ DAGG, float* __restrict__ DXN, float* __restrict__ DX1M, DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using U = UpdBwdL;
  stage_blob(lds, img, U::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 a[4], n[4], g[4], s[4], d[4], t[4];
    load_row(a, agg, r, L.g);
    load_row(n, xn, r, L.g);
    load_row(d, dx1, r, L.g);                                    // (ahead of the three forward products that do not need it)
    linear<4, 4>(g, a, lds + U::WIH, lds + U::BIH, L);
    linear<4, 4>(t, n, lds + U::WHH, lds + U::BHH, L);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) g[jt] += t[jt];
    sigmoid_<4>(g);
    linear<4, 4>(s, n, lds + U::WSELF, lds + U::BSELF, L);
    if (drop.p > 0.f) {
      f4 mk[4];
      drop_feat16(mk, drop, DK_PROJ, uint32_t(r), 0, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) d[jt] *= mk[jt];
      if (row < R) store_row(d, DX1M, row, L.g);
    }
    linear_t(t, d, lds + U::WOUT_T, L);                                 // t = d upd
    f4 upd[4], dgp[4], ds[4], dagg[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gg = g[jt][c], diff = s[jt][c] - a[jt][c];
        upd[jt][c] = a[jt][c] + gg * diff;
        dagg[jt][c] = t[jt][c] * (1.0f - gg);
        ds[jt][c] = t[jt][c] * gg;
        dgp[jt][c] = t[jt][c] * diff * gg * (1.0f - gg);
      }
    linear_adj<4, 4>(dagg, dgp, lds + U::WIH_T, L);
    linear_t(t, dgp, lds + U::WHH_T, L);
    linear_adj<4, 4>(t, ds, lds + U::WSELF_T, L);
    if (row < R) {
      store_row(upd, UPD, row, L.g);
      store_row(dgp, DGP, row, L.g);
      store_row(ds, DS, row, L.g);
      store_row(dagg, DAGG, row, L.g);
      store_row(t, DXN, row, L.g);
    }
  }
}

// dx = dres + norm1-backward(dxn_part + sum_j Wj^T dp_j);  optional xn_out = norm1(x);  per-wave (dgamma | dbeta)
template <int NQ>
__global__ __launch_bounds__(256) void k_node_proj_bwd(const float* __restrict__ img, const float* __restrict__ x,
                                                       const float* __restrict__ dres, const float* __restrict__ dxn_part,
                                                       const float* __restrict__ dp0, const float* __restrict__ dp1,
                                                       const float* __restrict__ dp2, int64_t R, float* __restrict__ dx_out,
                                                       float* __restrict__ xn_out, float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using P = ProjBwdL<NQ>;
  stage_blob(lds, img, P::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  f4 dgam[4], dbet[4];
  zero4(dgam); zero4(dbet);
  const float* dps[3] = {dp0, dp1, dp2};
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 t[4], xh[4], dq[NQ > 0 ? NQ : 1][4], dr[4];
    if (dxn_part) load_row(t, dxn_part, r, L.g);
    else zero4(t);
    // every row of the tile is requested before the first product (they used to be loaded one product at a time)
#pragma unroll
    for (int j = 0; j < NQ; ++j) load_row(dq[j], dps[j], r, L.g);
    load_row(xh, x, r, L.g);
    if (dres) load_row(dr, dres, r, L.g);
#pragma unroll
    for (int j = 0; j < NQ; ++j) linear_adj<4, 4>(t, dq[j], lds + P::WT + j * MAT64, L);
    if (row >= R) zero4(t);
    const float rstd = ln_normalize(xh);
    if (xn_out && row < R) {
      f4 o[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f4 ga = *reinterpret_cast<const f4*>(lds + P::N1G + 16 * jt + 4 * L.g);
        const f4 be = *reinterpret_cast<const f4*>(lds + P::N1B + 16 * jt + 4 * L.g);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[jt][c] = xh[jt][c] * ga[c] + be[c];
      }
      store_row(o, xn_out, row, L.g);
    }
    ln_backward(t, xh, rstd, lds + P::N1G, L.g, dgam, dbet);
    if (dres) {
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) t[jt] += dr[jt];
    }
    if (row < R) store_row(t, dx_out, row, L.g);
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 128;
  flush_vec(dgam, vp, L);
  flush_vec(dbet, vp + 64, L);
}
template __global__ void k_node_proj_bwd<0>(const float*, const float*, const float*, const float*, const float*, const float*,
                                            const float*, int64_t, float*, float*, float*);
template __global__ void k_node_proj_bwd<1>(const float*, const float*, const float*, const float*, const float*, const float*,
                                            const float*, int64_t, float*, float*, float*);
template __global__ void k_node_proj_bwd<3>(const float*, const float*, const float*, const float*, const float*, const float*,
                                            const float*, int64_t, float*, float*, float*);

// out (+)= W^T d  for the transposed 64x64 image `wt`
__global__ __launch_bounds__(256) void k_lin_t_acc(const float* __restrict__ wt, const float* __restrict__ d, int64_t R,
                                                   float* __restrict__ out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, wt, MAT64);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 a[4], t[4];
    load_row(a, d, r, L.g);
    if (accumulate) load_row(t, out, r, L.g);
    else zero4(t);
    linear_adj<4, 4>(t, a, lds, L);
    if (row < R) store_row(t, out, row, L.g);
  }
}

// out = sum_k W_k^T d_k over K transposed images `wt + k * wt_stride` and K row blocks `d + k * d_stride` (the aggregator's
// multihead_proj: K = num_modes launches of k_lin_t_acc, each reading and re-writing `out`, were 6-10 launches of ~6 us for 2 MB of
// rows).  One tile per wave, its sum kept in registers over k (same order of additions: the same bits); image k + 1 is copied into
// the other half of the LDS while image k is in use.
__global__ __launch_bounds__(256) void k_lin_t_sum(const float* __restrict__ wt, int64_t wt_stride, const float* __restrict__ d,
                                                   int64_t d_stride, int K, int64_t R, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Lane L;
  const int wave = threadIdx.x >> 6;
  const int64_t tile = int64_t(blockIdx.x) * (blockDim.x >> 6) + wave, ntiles = (R + 15) / 16;
  const bool live = tile < ntiles;
  const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
  f4 t[4];
  zero4(t);
  stage_blob(lds, wt, MAT64);
  for (int k = 0; k < K; ++k) {
    const float* cur = lds + (k & 1) * MAT64;
    if (k + 1 < K) stage_copy(lds + ((k + 1) & 1) * MAT64, wt + int64_t(k + 1) * wt_stride, MAT64);
    if (live) {
      keep_lds_reads_here();
      f4 a[4];
      load_row(a, d + int64_t(k) * d_stride, r, L.g);
      linear_adj<4, 4>(t, a, cur, L);
    }
    __syncthreads();
  }
  if (live && row < R) store_row(t, out, row, L.g);
}

// ------------------------------------------------------------------ MultipleInputEmbedding backward (EMB:62-70)
// forward: a0 = relu(LN(A_W0 in0 + b)), b0 likewise from in1; sp = WA3 a0 + WB3 b0 + b3; s = relu(LN0(sp));
//          ep = W2 s + b2; emb = LN3(ep).
// tail: given demb rows -> saves S (= s), DEP (d ep), DSP (d sp) (the branch activations a0, b0 are recomputed from the
//       geometry by the weight-gradient kernel: WgradBatch::add_in2); per-wave vectors
//       (dgamma3 | dbeta3 | dgamma0 | dbeta0) -> vpart[wave][256]
__device__ __forceinline__ void branch_fwd(f4 (&xh)[4], f4 (&act)[4], float& rstd, float i0, float i1, const float* w0,
                                           const float* b0, const float* gam, const float* bet, const Lane& L) {
  linear_in2(xh, i0, i1, w0, b0, L.g);
  rstd = ln_normalize(xh);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const f4 ga = *reinterpret_cast<const f4*>(gam + 16 * jt + 4 * L.g);
    const f4 be = *reinterpret_cast<const f4*>(bet + 16 * jt + 4 * L.g);
#pragma unroll
    for (int c = 0; c < 4; ++c) act[jt][c] = fmaxf(xh[jt][c] * ga[c] + be[c], 0.f);
  }
}

// ATTN: d emb is not read from `demb` but built here from the attention backward's per-edge scalars (bwd.hpp EdgeAttnGrad):
//   d emb_e = Wk^T (ED_e,h q[dst_e]) + Wv^T (EA_e,h dagg[dst_e])        (h = the head of each feature)
// -- two more transposed products per tile instead of a [E,64] row written by one kernel and read by this one.
#ifndef TSDE_TAIL_BOUNDS             // experiments: 768 caps the registers at 168 and forces spills
#define TSDE_TAIL_BOUNDS 512
#endif
template <bool ATTN>
__global__ __launch_bounds__(TSDE_TAIL_BOUNDS) void k_edge_embed_bwd_tail(const float* __restrict__ img, const float* __restrict__ geom,
                                                             const float* __restrict__ demb, EdgeAttnGrad ag, int64_t E,
                                                             float* __restrict__ S, float* __restrict__ DEP, float* __restrict__ DSP,
                                                             float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, EdgeBwdL::WA3T);                   // forward image + W2^T
  if (ATTN) {                                             // lin_k^T | lin_v^T behind it
    stage_copy(lds + EdgeBwdL::WA3T, ag.wkvt, 2 * MAT64);
    __syncthreads();
  }
  using EL = EdgeL6;                                     // split-precision recompute, fp32 transposes for the gradients
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int stage_off = EdgeBwdL::WA3T + (ATTN ? 2 * MAT64 : 0);      // the waves' store tiles sit behind the image (edge_embed_backward)
  const int64_t ntiles = (E + 15) / 16;
  f4 dg3[4], db3[4], dg0[4], db0[4];
  zero4(dg3); zero4(db3); zero4(dg0); zero4(db0);
  PhaseClock<8> clk;                                      // diagnostic builds only (stamps.hpp, tools/phase_stamps.py tail)
  clk.start();
  [[maybe_unused]] unsigned long long tiles_done = 0;          // (read by the phase clocks when they are compiled in)
  // (ATTN) the tile's targets are read ONE TILE AHEAD: the target index is the address of the q / dagg rows, and left at the top of its own
  // tile that dependent pair of latencies was 22 % of the kernel (in-kernel phase clocks, tools/phase_stamps.py tail: 3 900 of 18 150
  // cycles a tile).
  const int64_t tstride = int64_t(gridDim.x) * waves;
  int tgt_next = 0;                                       // (the geometry record too would be 4 registers more: 2 spilled at 256, measured slower)
  if (ATTN) {
    const int64_t e0 = (int64_t(blockIdx.x) * waves + wave) * 16 + L.n;
    tgt_next = ag.dst[e0 < E ? e0 : E - 1];
  }
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += tstride) {
    keep_lds_reads_here();
    ++tiles_done;
    const int64_t e = tile * 16 + L.n, ec = e < E ? e : E - 1;
    const f4 ge = *reinterpret_cast<const f4*>(geom + 4 * ec);
    const int tgt_now = tgt_next;
    if (ATTN) {
      const int64_t en = (tile + tstride) * 16 + L.n;
      tgt_next = ag.dst[en < E ? en : E - 1];
    }
    f4 xh[4], a0[4], b0[4], sp[4], s[4], ep[4], d[4];
    // ATTN: the target's q / dagg rows and the edge's scalars are requested HERE, ahead of the forward recompute that does not
    // need them (index -> row is two dependent latencies; left where they are used, they stalled every tile at two waves per SIMD)
    f4 qv[4], gv[4], d0, a0_, d1, a1_;
    if (!ATTN) load_row(d, demb, ec, L.g);
    if (ATTN) {
      const int tgt = tgt_now;
      load_row(qv, ag.q, tgt, L.g);
      load_row(gv, ag.dagg, tgt, L.g);
      // (the second halves by an address, not by `d1 = d0; if (8 heads) load`: copying a register that a load is still writing made the
      //  wave wait right here for every load it had just requested -- 24 % of the kernel in its phase clocks, round 4)
      const int half2 = ag.heads == 8 ? 4 : 0;
      d0 = *reinterpret_cast<const f4*>(ag.ED + ec * ag.heads);
      a0_ = *reinterpret_cast<const f4*>(ag.EA + ec * ag.heads);
      d1 = *reinterpret_cast<const f4*>(ag.ED + ec * ag.heads + half2);
      a1_ = *reinterpret_cast<const f4*>(ag.EA + ec * ag.heads + half2);
    }
    clk.mark(0);                                          // loads requested
    float r_;
    branch_fwd(xh, a0, r_, ge[0], ge[1], lds + EL::A_W0, lds + EL::A_B0, lds + EL::A_G, lds + EL::A_E, L);
    branch_fwd(xh, b0, r_, ge[2], ge[3], lds + EL::B_W0, lds + EL::B_B0, lds + EL::B_G, lds + EL::B_E, L);
    clk.mark(1);                                          // first layers of the two branches
    load_vec<4>(sp, lds + EL::B3, L.g);
    linear_acc_x6<4, 4>(sp, a0, lds + EL::WA3, L.lane);
    linear_acc_x6<4, 4>(sp, b0, lds + EL::WB3, L.lane);
    const float rs0 = ln_normalize(sp);                   // sp = s_hat
    bool pos[16];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 ga = *reinterpret_cast<const f4*>(lds + EL::AG0 + 16 * jt + 4 * L.g);
      const f4 be = *reinterpret_cast<const f4*>(lds + EL::AE0 + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pre = sp[jt][c] * ga[c] + be[c];
        pos[4 * jt + c] = pre > 0.f;
        s[jt][c] = fmaxf(pre, 0.f);
      }
    }
    clk.mark(2);                                          // two products, LayerNorm, ReLU
    linear_x6<4, 4>(ep, s, lds + EL::W2, lds + EL::B2, L);
    const float rs3 = ln_normalize(ep);                   // ep = e_hat
    clk.mark(3);                                          // third product, LayerNorm
    if (ATTN) {
      // lane group g holds the features of heads 2jt + (g >> 1) (8 heads) or jt (4 heads)
      const int odd = L.g >> 1;
      f4 dk[4], dv[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        float sd, sa;
        if (ag.heads == 8) {
          const f4& dd = jt < 2 ? d0 : d1;
          const f4& aa = jt < 2 ? a0_ : a1_;
          sd = odd ? dd[2 * (jt & 1) + 1] : dd[2 * (jt & 1)];
          sa = odd ? aa[2 * (jt & 1) + 1] : aa[2 * (jt & 1)];
        } else {
          sd = d0[jt];
          sa = a0_[jt];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dk[jt][c] = sd * qv[jt][c];
          dv[jt][c] = sa * gv[jt][c];
        }
      }
      linear_t(d, dk, lds + EdgeBwdL::WA3T, L);
      linear_adj<4, 4>(d, dv, lds + EdgeBwdL::WA3T + MAT64, L);
    }
    if (e >= E) zero4(d);
    clk.mark(4);                                          // d emb from the attention scalars: two adjoint products (ATTN) / the row load
    ln_backward(d, ep, rs3, lds + EL::AG3, L.g, dg3, db3);   // d := d ep
    f4 t[4];
    linear_t(t, d, lds + EdgeBwdL::W2T, L);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!pos[4 * jt + c]) t[jt][c] = 0.f;
    ln_backward(t, sp, rs0, lds + EL::AG0, L.g, dg0, db0);   // t := d sp
    clk.mark(5);                                          // LayerNorm backwards, W2^T adjoint, ReLU mask
    // the three slabs leave as whole rows (tile.hpp store_tile_rows; with store_row they were 0.36 of this kernel's 1.3 ms per step)
    float* stile = lds + stage_off + wave * ROWSTAGE;
    store_tile_rows(stile, s, S, tile * 16, E, L);
    store_tile_rows(stile, d, DEP, tile * 16, E, L);
    store_tile_rows(stile, t, DSP, tile * 16, E, L);
    clk.mark(6);                                          // three slabs out
  }
#ifdef TSDE_STAMPS
  if (ATTN && L.lane == 0 && (wave == 0 || wave == 5)) clk.flush(g_stamps_tail, tiles_done);
#endif
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 256;
  flush_vec(dg3, vp, L);
  flush_vec(db3, vp + 64, L);
  flush_vec(dg0, vp + 128, L);
  flush_vec(db0, vp + 192, L);
}

// branch BR (0: first input, 1: second): d act = W{A,B}3^T dsp * (act > 0) -> LN backward -> the 2-wide input linear.
// per-wave vectors (dgamma | dbeta | dW0[:,0] | dW0[:,1] | db0) -> vpart[wave][320]
template <int BR>
__global__ __launch_bounds__(256) void k_edge_embed_bwd_branch(const float* __restrict__ img, const float* __restrict__ geom,
                                                               const float* __restrict__ DSP, int64_t E,
                                                               float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL;
  constexpr int W0 = BR ? EL::B_W0 : EL::A_W0, B0_ = BR ? EL::B_B0 : EL::A_B0, G_ = BR ? EL::B_G : EL::A_G, E_ = BR ? EL::B_E : EL::A_E;
  // stage the input-embedding vectors and this branch's transposed matrix only
  static_assert(EL::WA3 % 4 == 0 && EdgeBwdL::WA3T % 4 == 0 && EdgeBwdL::WB3T % 4 == 0, "16-byte staging");
  stage_copy(lds, img, EL::WA3);
  stage_copy(lds + EL::WA3, img + (BR ? EdgeBwdL::WB3T : EdgeBwdL::WA3T), MAT64);
  __syncthreads();
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (E + 15) / 16;
  f4 dg[4], db[4], dwx[4], dwy[4], dbb[4];
  zero4(dg); zero4(db); zero4(dwx); zero4(dwy); zero4(dbb);
  // geometry record and gradient row of the NEXT tile are requested while this one is computed (the tail kernel's phase clocks showed
  // what a tile pays for loads requested at its own top)
  const int64_t tstride = int64_t(gridDim.x) * waves;
  f4 ge_next, d_next[4];
  {
    const int64_t t0 = int64_t(blockIdx.x) * waves + wave;
    const int64_t e0 = t0 * 16 + L.n, ec0 = e0 < E ? e0 : E - 1;
    ge_next = *reinterpret_cast<const f4*>(geom + 4 * ec0);
    load_row(d_next, DSP, ec0, L.g);
  }
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += tstride) {
    keep_lds_reads_here();
    const int64_t e = tile * 16 + L.n;
    const f4 ge = ge_next;
    f4 xh[4], act[4], d[4], t[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) d[jt] = d_next[jt];
    {
      const int64_t en = (tile + tstride) * 16 + L.n, enc = en < E ? en : E - 1;
      ge_next = *reinterpret_cast<const f4*>(geom + 4 * enc);
      load_row(d_next, DSP, enc, L.g);
    }
    const float i0 = BR ? ge[2] : ge[0], i1 = BR ? ge[3] : ge[1];
    float rstd;
    branch_fwd(xh, act, rstd, i0, i1, lds + W0, lds + B0_, lds + G_, lds + E_, L);
    if (e >= E) zero4(d);
    linear_t(t, d, lds + EL::WA3, L);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!(act[jt][c] > 0.f)) t[jt][c] = 0.f;
    ln_backward(t, xh, rstd, lds + G_, L.g, dg, db);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dwx[jt][c] = fmaf(t[jt][c], i0, dwx[jt][c]);
        dwy[jt][c] = fmaf(t[jt][c], i1, dwy[jt][c]);
        dbb[jt][c] += t[jt][c];
      }
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 320;
  flush_vec(dg, vp, L);
  flush_vec(db, vp + 64, L);
  flush_vec(dwx, vp + 128, L);
  flush_vec(dwy, vp + 192, L);
  flush_vec(dbb, vp + 256, L);
}
// Both branches in ONE pass over the d sp rows (round 5, an alternative: TRAJSDE_BRANCH_FUSED=1, measured slower -- see
// edge_embed_backward): the two kernels above read the same 256-byte row each, 1.15 GB twice per 64 x 128 training step.  Here a tile's
// row is loaded once and goes through branch A then branch B (the same device functions in the same order: per branch the arithmetic
// is that of k_edge_embed_bwd_branch<BR>); 160 accumulator registers per lane, so one wave per SIMD.  vpart: [branch][wave][320].
__global__ __launch_bounds__(256) void k_edge_embed_bwd_branch2(const float* __restrict__ img, const float* __restrict__ geom,
                                                                const float* __restrict__ DSP, int64_t E, float* __restrict__ vpartA,
                                                                float* __restrict__ vpartB) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using EL = EdgeL;
  stage_copy(lds, img, EL::WA3);
  stage_copy(lds + EL::WA3, img + EdgeBwdL::WA3T, MAT64);
  stage_copy(lds + EL::WA3 + MAT64, img + EdgeBwdL::WB3T, MAT64);
  __syncthreads();
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (E + 15) / 16;
  f4 dg[2][4], db[2][4], dwx[2][4], dwy[2][4], dbb[2][4];
#pragma unroll
  for (int br = 0; br < 2; ++br) { zero4(dg[br]); zero4(db[br]); zero4(dwx[br]); zero4(dwy[br]); zero4(dbb[br]); }
  const int64_t tstride = int64_t(gridDim.x) * waves;
  f4 ge_next, d_next[4];
  {
    const int64_t t0 = int64_t(blockIdx.x) * waves + wave;
    const int64_t e0 = t0 * 16 + L.n, ec0 = e0 < E ? e0 : E - 1;
    ge_next = *reinterpret_cast<const f4*>(geom + 4 * ec0);
    load_row(d_next, DSP, ec0, L.g);
  }
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += tstride) {
    keep_lds_reads_here();
    const int64_t e = tile * 16 + L.n;
    const f4 ge = ge_next;
    f4 d[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) d[jt] = d_next[jt];
    {
      const int64_t en = (tile + tstride) * 16 + L.n, enc = en < E ? en : E - 1;
      ge_next = *reinterpret_cast<const f4*>(geom + 4 * enc);
      load_row(d_next, DSP, enc, L.g);
    }
    if (e >= E) zero4(d);
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      const int W0 = br ? EL::B_W0 : EL::A_W0, B0_ = br ? EL::B_B0 : EL::A_B0, G_ = br ? EL::B_G : EL::A_G, E_ = br ? EL::B_E : EL::A_E;
      const float i0 = br ? ge[2] : ge[0], i1 = br ? ge[3] : ge[1];
      f4 xh[4], act[4], t[4];
      float rstd;
      branch_fwd(xh, act, rstd, i0, i1, lds + W0, lds + B0_, lds + G_, lds + E_, L);
      linear_t(t, d, lds + EL::WA3 + br * MAT64, L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (!(act[jt][c] > 0.f)) t[jt][c] = 0.f;
      ln_backward(t, xh, rstd, lds + G_, L.g, dg[br], db[br]);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dwx[br][jt][c] = fmaf(t[jt][c], i0, dwx[br][jt][c]);
          dwy[br][jt][c] = fmaf(t[jt][c], i1, dwy[br][jt][c]);
          dbb[br][jt][c] += t[jt][c];
        }
    }
  }
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    float* vp = (br ? vpartB : vpartA) + int64_t(blockIdx.x * waves + wave) * 320;
    flush_vec(dg[br], vp, L);
    flush_vec(db[br], vp + 64, L);
    flush_vec(dwx[br], vp + 128, L);
    flush_vec(dwy[br], vp + 192, L);
    flush_vec(dbb[br], vp + 256, L);
  }
}
template __global__ void k_edge_embed_bwd_branch<0>(const float*, const float*, const float*, int64_t, float*);
template __global__ void k_edge_embed_bwd_branch<1>(const float*, const float*, const float*, int64_t, float*);

// part[s][d][c] = sum over row slice s of X[i][d] * Y[i][head(d)][c]    (X [N,64], Y [N,heads,64]); grid (heads, S slices): a
// workgroup owns one head -- its 64 / heads output rows d share the Y rows, which are therefore read exactly once.  The four
// waves of a workgroup take every 4th row of the slice; the S partial 64x64 blocks are then summed by k_reduce_partials.
template <int HEADS>
__global__ __launch_bounds__(256) void k_headwise_outer_t(const float* __restrict__ X, const float* __restrict__ Y, int64_t N,
                                                          float* __restrict__ part) {
  constexpr int LPH = 64 / HEADS;
  __shared__ float red[4][LPH][64];
  const int h = blockIdx.x, c = threadIdx.x & 63, sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int S = gridDim.y, sl = blockIdx.y;
  const int64_t per = (N + S - 1) / S, lo = sl * per, hi = lo + per < N ? lo + per : N;
  float acc[LPH];
#pragma unroll
  for (int d = 0; d < LPH; ++d) acc[d] = 0.f;
  for (int64_t i0 = lo + sub; i0 < hi; i0 += 32) {          // eight rows of this wave in flight (the kernel waits on memory)
    float y[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + 4 * u < hi ? i0 + 4 * u : hi - 1;
      y[u] = Y[(i * HEADS + h) * 64 + c];
      x[u] = c < LPH ? X[i * 64 + LPH * h + c] : 0.f;        // the head's LPH inputs of the row, handed out as scalars below
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + 4 * u >= hi) break;
#pragma unroll
      for (int d = 0; d < LPH; ++d)
        acc[d] = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[u]), d)), y[u], acc[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < LPH; ++d) red[sub][d][c] = acc[d];
  __syncthreads();
  if (sub == 0) {
#pragma unroll
    for (int d = 0; d < LPH; ++d)
      part[int64_t(sl) * 4096 + (LPH * h + d) * 64 + c] = (red[0][d][c] + red[1][d][c]) + (red[2][d][c] + red[3][d][c]);
  }
}
int run_headwise_outer(const WgradCtx& wc, const float* X, const float* Y, int64_t N, float* W, int heads) {
  // slices: a workgroup keeps 32 rows in flight and waits on memory between rounds -- few rounds a workgroup (8 192 rows: 32 slices
  // 0.34 ms for the ten launches of a 64 x 128 step, 128 slices 0.28)
  static const int s_env = []() { const char* e = getenv("TRAJSDE_HEADWISE_SLICES"); return e ? atoi(e) : 0; }();
  const int S = s_env > 0 && N >= 4096 ? s_env : (N >= 65536 ? 256 : (N >= 4096 ? 128 : (N >= 256 ? 8 : 1)));
  if (int64_t(S) > wc.cap) return fail(TRAJSDE_ERR_WORKSPACE, "headwise_outer: partial buffer too small");
  ReduceQueue* rq = active_reduce_queue();
  if (rq && rq->part != wc.part) rq = nullptr;
  float* part = wc.part;
  if (rq) {                                             // deferred sums (bwd.hpp): the S partials keep their slots until finish()
    int rc = TRAJSDE_OK;
    const int64_t base = rq->take(S, &rc);
    if (rc) return rc;
    part += base * 4096;
    rq->jobs.push_back(ReduceJob{W, nullptr, base, S, 1, 64, 0, 0});
  }
  if (heads == 4) TS_LAUNCH_TAG("k_headwise_outer", false, k_headwise_outer_t<4>, dim3(4, S), 256, 0, wc.st, X, Y, N, part);
  else TS_LAUNCH_TAG("k_headwise_outer", false, k_headwise_outer_t<8>, dim3(8, S), 256, 0, wc.st, X, Y, N, part);
  if (rq) return TRAJSDE_OK;
  WgradJobs one;
  one.n = 1;
  one.j[0] = WgradJob{nullptr, nullptr, W, nullptr, 0, 0, 64, 0, 0};
  TS_LAUNCH(k_reduce_partials, dim3(cdiv(4096, 32), 1), 256, 0, wc.st, one, wc.part, wc.cs, S, 1, nullptr);
  return TRAJSDE_OK;
}

// ------------------------------------------------------------------ host drivers
// out = x1 + W2 relu(W1 norm2(x1) + b1) + b2 given dout: sc.dx1 = d x1 (residual + through norm2), FFN / norm2 gradients
int ffn_block_backward(const float* img_a, const float* img_b, const float* xn2, const float* x1, const float* dout, int64_t R,
                       const NodeBlockScratch& sc, const WgradCtx& wc, const NodeBlockGrads& gr, hipStream_t st, const DropArg& drop) {
  const int64_t ntiles = (R + 15) / 16;
  const int ga = tile_grid(ntiles, 256, FfnBwdAL::SIZE * 4), gb = vec_grid(ntiles, 256, FfnBwdBL::SIZE * 4);
  // dropout: the masked gradient rows entering mlp.3 live in sc.UPD until this function's weight gradients are enqueued
  // (k_upd_bwd, which owns that buffer, runs after them on the same stream)
  const float* dout2 = drop.p > 0.f ? sc.UPD : dout;
  TS_LAUNCH(k_ffn_bwd_a, ga, 256, FfnBwdAL::SIZE * 4, st, img_a, dout, xn2, R, sc.H, sc.DH, sc.UPD, drop);
  float* const vp = vpart_slab(sc.vpart, int64_t(gb) * 4, 128);
  TS_LAUNCH(k_ffn_bwd_b, gb, 256, FfnBwdBL::SIZE * 4, st, img_b, sc.DH, dout, x1, R, sc.dx1, vp);
  {
    ColsumBatch cb(st, gb * 4, 128);
    cb.add(vp, 64, gr.n2g);
    cb.add(vp + 64, 64, gr.n2b);
    if (int rc = cb.flush()) return rc;
  }
  // first layer [256,64] (four 64-row blocks) and second layer [64,256] (four 64-column blocks): eight problems, one launch pair
  WgradBatch wb(wc, R, R);
  for (int b = 0; b < 4; ++b) {
    if (int rc = wb.add(sc.DH + 64 * b, 256, xn2, 64, gr.w1 + b * MAT64, 64, 0, gr.b1 + 64 * b, 0)) return rc;
    if (int rc = wb.add(dout2, 64, sc.H + 64 * b, 256, gr.w2, 256, 64 * b, b == 0 ? gr.b2 : nullptr, 0)) return rc;
  }
  return wb.flush();
}

// `defer`: a batch of the caller over the same R rows that takes the gated update's four weight-gradient problems instead of a launch of
// their own.  Only for a caller whose later kernels leave sc.UPD / DGP / DS / dx1 / H alone until it flushes: the AGGREGATOR's layers
// do (seven N-row problems in one launch); the ENCODER's blocks do NOT -- their attention backward keeps its per-target sums in the
// same memory as sc.H (encoder_bwd.hip EncBwdWs: RL = nb.H), which under dropout holds the masked d x1 rows of the out_proj problem
// (tried: test_full_size_training_step_agrees_between_kernel_forms caught it at once)
int node_block_backward(const float* img, const NodeBlockTape& tp, const float* dout, int64_t R, const NodeBlockScratch& sc,
                        const WgradCtx& wc, const NodeBlockGrads& gr, float* dagg, float* dxn, hipStream_t st, const DropArg& drop,
                        WgradBatch* defer) {
  const int64_t ntiles = (R + 15) / 16;
  const int gu = tile_grid(ntiles, 256, UpdBwdL::SIZE * 4);
  if (int rc = ffn_block_backward(img + NodeBlockBwdL::FFN_A, img + NodeBlockBwdL::FFN_B, tp.xn2, tp.x1, dout, R, sc, wc, gr, st, drop)) return rc;
  // dropout: the masked gradient rows entering out_proj go to sc.H (the FFN's weight gradients, its last readers, are enqueued)
  const float* dx1m = drop.p > 0.f ? sc.H : sc.dx1;
  TS_LAUNCH(k_upd_bwd, gu, 256, UpdBwdL::SIZE * 4, st, img + NodeBlockBwdL::UPD, sc.dx1, tp.agg, tp.xn, R, sc.UPD, sc.DGP, sc.DS, dagg, dxn, sc.H,
            drop);
  WgradBatch own(wc, R, R);
  WgradBatch& wb = defer ? *defer : own;
  if (int rc = wb.add(dx1m, 64, sc.UPD, 64, gr.w_out, 64, 0, gr.b_out, 0)) return rc;
  if (int rc = wb.add(sc.DGP, 64, tp.agg, 64, gr.w_ih, 64, 0, gr.b_ih, 0)) return rc;
  if (int rc = wb.add(sc.DGP, 64, tp.xn, 64, gr.w_hh, 64, 0, gr.b_hh, 0)) return rc;
  if (int rc = wb.add(sc.DS, 64, tp.xn, 64, gr.w_self, 64, 0, gr.b_self, 0)) return rc;
  return defer ? TRAJSDE_OK : own.flush();
}

int edge_embed_backward(const float* img, const float* geom, const float* demb, int64_t E, const EdgeEmbedScratch& sc,
                        const WgradCtx& wc, const EdgeEmbedGrads& gr, hipStream_t st, const EdgeAttnGrad* ag) {
  if (E <= 0) return TRAJSDE_OK;
  const int64_t ntiles = (E + 15) / 16;
  const int tail_threads = 512;                            // one workgroup per CU (image + a store tile per wave: 113 / 145 KB)
  const int lds_tail = (EdgeBwdL::WA3T + (ag ? 2 * MAT64 : 0) + (tail_threads / 64) * ROWSTAGE) * 4, lds_br = (EdgeL::WA3 + MAT64) * 4;
  const int gt = vec_grid(ntiles, tail_threads, lds_tail), gb = vec_grid(ntiles, 256, lds_br);
  const int tail_waves = tail_threads / 64;
  float* vp = vpart_slab(sc.vpart, int64_t(gt) * tail_waves, 256);
  if (ag) TS_LAUNCH_TAG("k_edge_embed_bwd_tail<attn>", false, k_edge_embed_bwd_tail<true>, gt, tail_threads, lds_tail, st, img, geom, demb, *ag, E,
                        sc.S, sc.DEP, sc.DSP, vp);
  else TS_LAUNCH_TAG("k_edge_embed_bwd_tail", false, k_edge_embed_bwd_tail<false>, gt, tail_threads, lds_tail, st, img, geom, demb, EdgeAttnGrad{}, E,
                     sc.S, sc.DEP, sc.DSP, vp);
  float* const tail_vec[4] = {gr.ag3, gr.ae3, gr.ag0, gr.ae0};
  {
    ColsumBatch cb(st, gt * tail_waves, 256);
    for (int i = 0; i < 4; ++i) cb.add(vp + 64 * i, 64, tail_vec[i]);
    if (int rc = cb.flush()) return rc;
  }
  {
    // three problems over the E rows: (DEP, S) 512 B per row, 2 x (DSP, geometry) 272 B per row -- the HBM-bound launch whose
    // roofline tools/train_step_bench.py reports
    WgradBatch wb(wc, E, E, "k_wgrad[edge-embed]");
    if (int rc = wb.add(sc.DEP, 64, sc.S, 64, gr.w2, 64, 0, gr.b2, 0)) return rc;
    if (int rc = wb.add_in2(sc.DSP, 64, geom, 0, img + EdgeL6::A_C, img + EdgeL6::A_E, gr.wa3, 64, gr.ba3)) return rc;
    if (int rc = wb.add_in2(sc.DSP, 64, geom, 1, img + EdgeL6::B_C, img + EdgeL6::B_E, gr.wb3, 64, gr.bb3)) return rc;
    if (int rc = wb.flush_edge()) return rc;
  }
  {
    // TRAJSDE_BRANCH_FUSED=1: both branches in one pass over the d sp rows (k_edge_embed_bwd_branch2).  Measured (64 x 128 step, one
    // box, alternating): 1.047 ms against 0.379 + 0.367 = 0.746 ms for the two one-branch kernels -- at one wave per SIMD (296
    // registers) the pass is bound by its dependent chain, not by the rows it saves reading.  Off by default; kept as the record.
    static const bool fused = []() { const char* e = getenv("TRAJSDE_BRANCH_FUSED"); return e && atoi(e) == 1; }();
    const int lds_br2 = (EdgeL::WA3 + 2 * MAT64) * 4;
    const int g2 = fused ? vec_grid(ntiles, 256, lds_br2) : gb;
    auto sums = [&](int br, float* vp, int g) -> int {
      float* g_ = br ? gr.b_g : gr.a_g;
      float* e_ = br ? gr.b_e : gr.a_e;
      float* w0 = br ? gr.b_w0 : gr.a_w0;
      float* b0 = br ? gr.b_b0 : gr.a_b0;
      ColsumBatch cb(st, g * 4, 320);
      cb.add(vp, 64, g_);
      cb.add(vp + 64, 64, e_);
      cb.add(vp + 128, 64, w0, 2);        // [64,2] weight, column 0
      cb.add(vp + 192, 64, w0 + 1, 2);    // column 1
      cb.add(vp + 256, 64, b0);
      return cb.flush();
    };
    float* vps[2] = {nullptr, nullptr};
    if (fused) {
      for (int br = 0; br < 2; ++br) vps[br] = vpart_slab(sc.vpart, int64_t(g2) * 4, 320);
    }
    if (fused && vps[0] != vps[1]) {      // (two slabs of their own: not the shared fallback slab twice)
      TS_LAUNCH(k_edge_embed_bwd_branch2, g2, 256, lds_br2, st, img, geom, sc.DSP, E, vps[0], vps[1]);
      for (int br = 0; br < 2; ++br)
        if (int rc = sums(br, vps[br], g2)) return rc;
    } else {
      for (int br = 0; br < 2; ++br) {
        float* vp = vpart_slab(sc.vpart, int64_t(gb) * 4, 320);
        if (br == 0) TS_LAUNCH(k_edge_embed_bwd_branch<0>, gb, 256, lds_br, st, img, geom, sc.DSP, E, vp);
        else TS_LAUNCH(k_edge_embed_bwd_branch<1>, gb, 256, lds_br, st, img, geom, sc.DSP, E, vp);
        if (int rc = sums(br, vp, gb)) return rc;
      }
    }
  }
  return TRAJSDE_OK;
}

}  // namespace tsde
