// grid_bwd.hip -- backward of the vanilla HiVT variant's own pieces (SURVEY.md 8(f) rank 4 + rank 1): the MLPDecoder under
// the winner-takes-all L2 loss and (below) the TemporalEncoder; the attention families and the node blocks reuse the
// kernels of encoder_bwd.hip / node_bwd.hip / aggregator_bwd.hip with 4 heads.
//
// MLPDecoder (GDEC:47-63) + L2 (losses/L2.py:10-27): only the winning mode of each actor carries gradient, so the head and
// aggr_embed backward run on N rows.  The scale and pi heads get no gradient from this loss.
#include "attn_common.hpp"
#include "bwd.hpp"
#include "common.hpp"
#include "dropout.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "tile.hpp"
#include "tile_bwd.hpp"

namespace tsde {

// loc head forward + backward on the winning rows: out [N,64] -> l [2T]; dL/dl from (y, mask, 1/count).
// writes H (relu output, input of loc.3), DL [N,128] (d l, zero beyond 2T), DU (d of loc.0's output), DOUT (d out);
// per-wave (dgamma | dbeta) of loc.1 -> vpart[wave][128]
__global__ __launch_bounds__(256) void k_mlp_heads_bwd(const float* __restrict__ img, const float* __restrict__ out,
                                                       const float* __restrict__ y, const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ scal, int N, int T, float* __restrict__ H,
                                                       float* __restrict__ DL, float* __restrict__ DU, float* __restrict__ DOUT,
                                                       float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, MlpHeadBwdL::SIZE);
  using M = MlpHeadBwdL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  const float inv_count = scal[1];
  f4 dgam[4], dbet[4];
  zero4(dgam); zero4(dbet);
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    keep_lds_reads_here();
    const int row = tile * 16 + L.n, i = row < N ? row : N - 1;
    f4 a[4], u[4], h[4], l[8], dl[8];
    load_row(a, out, i, L.g);
    linear<4, 4>(u, a, lds + M::W0, lds + M::B0, L);
    const float rstd = ln_normalize(u);
    bool pos[16];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 ga = *reinterpret_cast<const f4*>(lds + M::G + 16 * jt + 4 * L.g);
      const f4 be = *reinterpret_cast<const f4*>(lds + M::E + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pre = u[jt][c] * ga[c] + be[c];
        pos[4 * jt + c] = pre > 0.f;
        h[jt][c] = fmaxf(pre, 0.f);
      }
    }
    linear<8, 4>(l, h, lds + M::W3, lds + M::B3, L);
    // lane (n, g) holds outputs 16jt + 4g + c: steps t0 = 8jt + 2g (c = 0,1 -> x,y) and t0 + 1 (c = 2,3)
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      dl[jt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2) {
        const int t = 8 * jt + 2 * L.g + u2;
        if (row < N && t < T && mask[int64_t(i) * T + t]) {
          const float dx = l[jt][2 * u2] - y[(int64_t(i) * T + t) * 2], dy = l[jt][2 * u2 + 1] - y[(int64_t(i) * T + t) * 2 + 1];
          const float nrm = sqrtf(dx * dx + dy * dy);
          if (nrm > 0.f) {
            dl[jt][2 * u2] = dx / nrm * inv_count;
            dl[jt][2 * u2 + 1] = dy / nrm * inv_count;
          }
        }
      }
    }
    f4 dh[4];
    zero4(dh);
    linear_adj<4, 8>(dh, dl, lds + M::W3T, L);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!pos[4 * jt + c]) dh[jt][c] = 0.f;
    ln_backward(dh, u, rstd, lds + M::G, L.g, dgam, dbet);       // dh := d u
    f4 dout[4];
    linear_t(dout, dh, lds + M::W0T, L);
    if (row < N) {
      store_row(h, H, row, L.g);
      store_row(dh, DU, row, L.g);
      store_row(dout, DOUT, row, L.g);
      float* p = DL + int64_t(row) * 128 + 4 * L.g;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) *reinterpret_cast<f4*>(p + 16 * jt) = dl[jt];
    }
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 128;
  flush_vec(dgam, vp, L);
  flush_vec(dbet, vp + 64, L);
}

// ------------------------------------------------------------------ TemporalEncoder backward (GENC:241-292)
constexpr int TRB_S = 22;

// d of transformer_encoder.norm on the cls rows: DX[n][21] = LN-backward(dtout[n]) (DX zeroed by the caller);
// per-wave (dgamma | dbeta) -> vpart[wave][128]
__global__ __launch_bounds__(256) void k_tr_final_bwd(const float* __restrict__ norm, const float* __restrict__ x,
                                                      const float* __restrict__ dtout, int N, float* __restrict__ DX,
                                                      float* __restrict__ vpart) {
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  f4 dgam[4], dbet[4];
  zero4(dgam); zero4(dbet);
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    const int row = tile * 16 + L.n, r = row < N ? row : N - 1;
    f4 xh[4], d[4];
    load_row(xh, x, int64_t(r) * TRB_S + (TRB_S - 1), L.g);
    const float rstd = ln_normalize(xh);
    load_row(d, dtout, r, L.g);
    if (row >= N) zero4(d);
    ln_backward(d, xh, rstd, norm, L.g, dgam, dbet);
    if (row < N) store_row(d, DX, int64_t(row) * TRB_S + (TRB_S - 1), L.g);
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 128;
  flush_vec(dgam, vp, L);
  flush_vec(dbet, vp + 64, L);
}

// causal self-attention backward, one wave per actor (lane = feature): given dO -> dQ, dK, dV (rows [n][s][64]).
// Recomputes the softmax per query; with alpha the weights, d alpha_j = dO_i . v_j (head-wise), d logit_j =
// alpha_j (d alpha_j - sum_j' alpha_j' d alpha_j'), and the logit is (q * dh^-0.5) . k.  DROP (train mode): the values were summed
// with alpha_j m_j (m the dropout factors of dropout.hpp drop_tr_attn8), so d alpha_j = m_j (dO_i . v_j) and d v_j += alpha_j m_j dO_i.
template <int HEADS, bool DROP>
__global__ __launch_bounds__(256) void k_tr_attention_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ dO, int N,
                                                          float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, DropArg drop) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  constexpr float SCALE = HEADS == 4 ? 0.25f : INV_SQRT_DH;
  const int head = lane / (64 / HEADS);
  const int64_t base = int64_t(n) * TRB_S * 64 + lane;
  float kr[TRB_S], vr[TRB_S], dkr[TRB_S], dvr[TRB_S];
#pragma unroll
  for (int j = 0; j < TRB_S; ++j) {
    kr[j] = k[base + j * 64];
    vr[j] = v[base + j * 64];
    dkr[j] = 0.f;
    dvr[j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < TRB_S; ++i) {
    const float qd = q[base + i * 64] * SCALE;
    const float go = dO[base + i * 64];
    float p[TRB_S];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      p[j] = head_sum_n<HEADS>(qd * kr[j]);
      m = fmaxf(m, p[j]);
    }
    float mk[24];
    if (DROP) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (8 * c <= i) drop_tr_attn8<HEADS>(mk + 8 * c, drop, uint32_t(n), head, i, c);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      p[j] = fast_exp(p[j] - m);
      s += p[j];
    }
    const float inv = 1.0f / s;
    float dal[TRB_S], dlt = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      p[j] *= inv;                                            // alpha_j
      dal[j] = head_sum_n<HEADS>(go * vr[j]);
      if (DROP) dal[j] *= mk[j];
      dlt = fmaf(p[j], dal[j], dlt);
    }
    float dqi = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const float dl = p[j] * (dal[j] - dlt);
      dqi = fmaf(dl, kr[j], dqi);
      dkr[j] = fmaf(dl, qd, dkr[j]);
      dvr[j] = fmaf(DROP ? p[j] * mk[j] : p[j], go, dvr[j]);
    }
    dq[base + i * 64] = dqi * SCALE;
  }
#pragma unroll
  for (int j = 0; j < TRB_S; ++j) {
    dk[base + j * 64] = dkr[j];
    dv[base + j * 64] = dvr[j];
  }
}
template __global__ void k_tr_attention_bwd<4, false>(const float*, const float*, const float*, const float*, int, float*, float*, float*, DropArg);
template __global__ void k_tr_attention_bwd<8, false>(const float*, const float*, const float*, const float*, int, float*, float*, float*, DropArg);
template __global__ void k_tr_attention_bwd<4, true>(const float*, const float*, const float*, const float*, int, float*, float*, float*, DropArg);
template __global__ void k_tr_attention_bwd<8, true>(const float*, const float*, const float*, const float*, int, float*, float*, float*, DropArg);

// dst[r] = src[r] * (dropout factors of site `kind` on row r): the gradient that enters a Linear whose output went through a
// node-site dropout (dropout1 of a TemporalEncoder layer: x1 = x + m (Wout o + b), so Wout sees dx1 m, the residual dx1)
__global__ __launch_bounds__(256) void k_drop_rows(const float* __restrict__ src, int64_t R, float* __restrict__ dst, DropArg drop, int kind) {
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    const int64_t row = tile * 16 + L.n;
    if (row >= R) continue;
    f4 a[4], mk[4];
    load_row(a, src, row, L.g);
    drop_feat16(mk, drop, kind, uint32_t(row), 0, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) a[jt] *= mk[jt];
    store_row(a, dst, row, L.g);
  }
}

// d aa_out[t][n] = padded ? 0 : dX0[n][t]
__global__ void k_tr_prep_bwd(const float* __restrict__ DX0, const uint8_t* __restrict__ pad, int N, int TT, float* __restrict__ DAA) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(N) * (TRB_S - 1) * 64) return;
  const int c = int(i & 63), n = int((i >> 6) % N), t = int((i >> 6) / N);
  DAA[i] = pad[int64_t(n) * TT + t] ? 0.f : DX0[(int64_t(n) * TRB_S + t) * 64 + c];
}

// token gradients, one workgroup per token s: dpos[s] = sum_n dX0[n][s]; dpad[s] = sum over padded n (s < 21); dcls = sum (s = 21)
__global__ __launch_bounds__(1024) void k_tr_tok_grad(const float* __restrict__ DX0, const uint8_t* __restrict__ pad, int N, int TT,
                                                      float* __restrict__ dpad, float* __restrict__ dcls, float* __restrict__ dpos) {
  __shared__ float red[2][16][64];
  const int s = blockIdx.x, c = threadIdx.x & 63, part = threadIdx.x >> 6;
  float all = 0.f, pd = 0.f;
  for (int n = part; n < N; n += 16) {
    const float v = DX0[(int64_t(n) * TRB_S + s) * 64 + c];
    all += v;
    if (s < TRB_S - 1 && pad[int64_t(n) * TT + s]) pd += v;
  }
  red[0][part][c] = all;
  red[1][part][c] = pd;
  __syncthreads();
  if (part == 0) {
    float a = 0.f, b = 0.f;
    for (int p = 0; p < 16; ++p) { a += red[0][p][c]; b += red[1][p][c]; }
    dpos[s * 64 + c] = a;
    if (s < TRB_S - 1) dpad[s * 64 + c] = b;
    else dcls[c] = a;
  }
}

}  // namespace tsde

using namespace tsde;

namespace {
enum MlpGradSlot { L0W = 0, L0B, L1W, L1B, L3W, L3B, A0W, A0B, A1W, A1B, N_MLP_GRADS };

struct MlpBwdWs {
  int32_t *best, *cnt;
  float *minsum, *scal, *out, *gsel, *H, *DL, *DU, *DOUT, *DA, *w3tmp, *b3tmp, *part, *cs, *vpart;
  int64_t bytes, parts;
  bool ok;
  MlpBwdWs(void* ws, int64_t n, int N) {
    Carver c(ws, n);
    const int64_t slab = int64_t(N) * 64;
    best = c.take<int32_t>(N); cnt = c.take<int32_t>(N); minsum = c.take<float>(N); scal = c.take<float>(4);
    out = c.take<float>(slab); gsel = c.take<float>(slab); H = c.take<float>(slab); DL = c.take<float>(slab * 2);
    DU = c.take<float>(slab); DOUT = c.take<float>(slab); DA = c.take<float>(slab);
    w3tmp = c.take<float>(128 * 64); b3tmp = c.take<float>(128);
    parts = wgrad_max_parts(N, 1);
    part = c.take<float>(parts * 4096); cs = c.take<float>(parts * 64);
    vpart = c.take<float>(int64_t(512) * 4 * 128);
    bytes = c.off + 256;
    ok = c.ok;
  }
};
}  // namespace

extern "C" {

int64_t trajsde_mlp_decoder_backward_ws_bytes(int32_t N) { return MlpBwdWs(nullptr, 0, N).bytes; }

int trajsde_mlp_decoder_l2_backward(int32_t N, int num_modes, int future_steps, const float* blob_bwd, const float* local_embed,
                                    const float* global_embed, const float* loc, const float* y, const uint8_t* reg_mask, void* ws,
                                    int64_t ws_bytes, float* loss, int32_t* best_mode, float* const* grads, int n_grads,
                                    float* d_local, float* d_global, void* stream_) {
  TS_REQUIRE(blob_bwd && local_embed && global_embed && loc && y && reg_mask && ws && loss && grads && d_local && d_global,
             "mlp_decoder_l2_backward: null pointer");
  TS_REQUIRE(N > 0 && num_modes > 0 && future_steps > 0 && future_steps <= 64, "mlp_decoder_l2_backward: need 0 < future_steps <= 64");
  TS_REQUIRE(n_grads == N_MLP_GRADS, "mlp_decoder_l2_backward: gradient count does not match trajsde_param_count(DECODER_MLP_BWD)");
  for (int i = 0; i < N_MLP_GRADS; ++i) TS_REQUIRE(grads[i] != nullptr, "mlp_decoder_l2_backward: null gradient buffer");
  MlpBwdWs w(ws, ws_bytes, N);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "mlp_decoder_l2_backward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int K = num_modes, T = future_steps;
  const int ntiles = (N + 15) / 16;
  const float* init_img = blob_bwd + MlpDecBwdBlob::INIT;
  const WgradCtx wc{st, w.part, w.cs, nullptr, w.parts};
  {
    TS_REQUIRE(K >= 1 && K <= 256, "mlp decoder backward: 1 <= num_modes <= 256");
    int KP = 1;
    while (KP < K) KP <<= 1;
    TS_LAUNCH(k_l2_wta, cdiv(N, 256 / KP), 256, 0, st, loc, y, reg_mask, N, K, T, w.best, w.minsum, w.cnt, KP);
  }
  TS_LAUNCH(k_l2_finalize, 1, 1024, 0, st, w.minsum, w.cnt, N, w.scal);
  TS_HIP(hipMemcpyAsync(loss, w.scal, sizeof(float), hipMemcpyDeviceToDevice, st));
  if (best_mode) TS_HIP(hipMemcpyAsync(best_mode, w.best, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
  const int g128 = vec_grid(ntiles, 128, InitBwdL::SIZE * 4);
  TS_LAUNCH(k_init_sel, g128, 128, InitBwdL::AE_END * 4, st, init_img, local_embed, global_embed, w.best, N, w.out, w.gsel);
  const int gh = vec_grid(ntiles, 256, MlpHeadBwdL::SIZE * 4);
  TS_LAUNCH(k_mlp_heads_bwd, gh, 256, MlpHeadBwdL::SIZE * 4, st, blob_bwd + MlpDecBwdBlob::HEAD, w.out, y, reg_mask, w.scal, N, T, w.H, w.DL,
            w.DU, w.DOUT, w.vpart);
  {
    ColsumBatch cb(st, gh * 4, 128);
    cb.add(w.vpart, 64, grads[L1W]);
    cb.add(w.vpart + 64, 64, grads[L1B]);
    if (int rc = cb.flush()) return rc;
  }
  // loc.3 [2T, 64]: two 64-row blocks into a 128-row scratch, the first 2T rows are the gradient
  for (int b = 0; b < 2; ++b)
    if (int rc = run_wgrad(wc, w.DL + 64 * b, 128, w.H, 64, N, N, w.w3tmp + b * MAT64, 64, 0, w.b3tmp + 64 * b, 0)) return rc;
  TS_HIP(hipMemcpyAsync(grads[L3W], w.w3tmp, size_t(2 * T) * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  TS_HIP(hipMemcpyAsync(grads[L3B], w.b3tmp, size_t(2 * T) * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (int rc = run_wgrad(wc, w.DU, 64, w.out, 64, N, N, grads[L0W], 64, 0, grads[L0B], 0)) return rc;
  TS_HIP(hipMemsetAsync(d_global, 0, size_t(K) * N * 64 * sizeof(float), st));
  TS_LAUNCH(k_dec_init_bwd, g128, 128, InitBwdL::SIZE * 4, st, init_img, local_embed, w.gsel, w.DOUT, w.best, N, w.DA, d_local, d_global,
            w.vpart);
  {
    ColsumBatch cb(st, g128 * 2, InitV::SIZE);
    cb.add(w.vpart + InitV::DGAM, 64, grads[A1W]);
    cb.add(w.vpart + InitV::DBET, 64, grads[A1B]);
    if (int rc = cb.flush()) return rc;
  }
  if (int rc = run_wgrad(wc, w.DA, 64, w.gsel, 64, N, N, grads[A0W], 128, 0, grads[A0B], 0)) return rc;      // cat(global, local)
  return run_wgrad(wc, w.DA, 64, local_embed, 64, N, N, grads[A0W], 128, 64, nullptr, 0);
}

}  // extern "C"
