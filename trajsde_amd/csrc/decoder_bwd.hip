// decoder_bwd.hip -- winner-takes-all L2 regression loss (reference losses/L2.py:10-27) and the backward pass of the
// SDEDecoder stage (DEC:77-105) for gfx950: gradients w.r.t. the decoder's parameters and w.r.t. its two inputs
// (local_embed, global_embed).  First family of SURVEY.md 8(f) rank 1.
//
// Only the winning mode of each actor carries gradient, so everything below runs on N rows (path r_i = best_i*N + i),
// not K*N.  The Euler-Maruyama solve is differentiated "discretise-then-optimise": the forward trajectory of the
// N winning paths is replayed once (same Philox counters, so the same noise) keeping every state and hidden
// activation, then a reverse sweep propagates dL/dy_k through the step map y' = y + f(y,t) dt + g(y,t) z sqrt(h).
//
//   k_l2_wta / k_l2_finalize   best mode per actor, loss value, 1/count
//   k_init_sel                 y0 of the winning paths                      (aggr_embed, DEC:82)
//   k_sde_replay               forward replay, keeps y_k, tanh activations and g  [step][row][64]
//   k_head_bwd                 loc head forward + backward per (row, output step): dL/ds_o, saves (s, du)
//   k_sde_bwd                  the reverse sweep; saves the pre-activation gradients of the five linears
//   k_dec_init_bwd             aggr_embed backward: d local_embed, d global_embed, saves (input, da)
//   k_wgrad / k_reduce_partials   dW = sum_rows delta^T a  as MFMA outer products over saved rows (deterministic
//                              two-stage reduction), bias = column sums, time-feature columns = step-weighted sums
//
// Matrix products run on transposed images (layouts.hpp SweepL / HeadBwdL / InitBwdL): dX^T = W^T dY^T has the same
// "row on lane" operand/result layout as the forward.  tile.hpp linear_adj: row-scaled split precision in the fp16x3
// build, the exact fp32 instruction in the bf16x6 build; the weight gradients (k_wgrad) are always exact fp32.
#include <cstdlib>

#include "common.hpp"
#include "layouts.hpp"
#include "philox.hpp"
#include "sde_funcs.hpp"
#include "tile.hpp"
#include "tile_bwd.hpp"
#include "bwd.hpp"
#include "kernels.hpp"

namespace tsde {

// ------------------------------------------------------------------ loss
// masked L2 per (actor, mode), first minimum wins (L2.py:19-22).  One thread per (actor, mode): the T steps' mask bytes, targets and
// predictions are requested eight steps at a time and summed in step order -- the same sums, bit for bit, as the one-thread-per-actor
// loop this replaces, which walked K x T dependent loads per thread (0.29 ms at 128 x 48 agents, K = 10, T = 60: 4 % of that step).
// The argmin over the modes of an actor goes through LDS: 256 / KP actors a workgroup (KP = K rounded up to a power of two).
__global__ __launch_bounds__(256) void k_l2_wta(const float* __restrict__ loc, const float* __restrict__ y, const uint8_t* __restrict__ mask, int N, int K,
                                                int T, int32_t* __restrict__ best, float* __restrict__ minsum, int32_t* __restrict__ cnt, int KP) {
  __shared__ float s_sum[256];
  __shared__ int s_cnt[256];
  const int per = 256 / KP, a = threadIdx.x / KP, k = threadIdx.x - a * KP;
  const int i = blockIdx.x * per + a;
  const bool live = a < per && i < N && k < K;
  float s = 0.f;
  int c = 0;
  if (live) {
    const uint8_t* mk = mask + int64_t(i) * T;
    const float2* yy = reinterpret_cast<const float2*>(y) + int64_t(i) * T;
    const f4* ll = reinterpret_cast<const f4*>(loc) + (int64_t(k) * N + i) * T;
    for (int t0 = 0; t0 < T; t0 += 8) {
      uint8_t m[8];
      float2 yv[8];
      f4 lv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u < T ? t0 + u : T - 1;
        m[u] = mk[t];
        yv[u] = yy[t];
        lv[u] = ll[t];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (t0 + u >= T || !m[u]) continue;
        const float dx = yv[u].x - lv[u][0], dy = yv[u].y - lv[u][1];
        s += sqrtf(dx * dx + dy * dy);
        ++c;
      }
    }
  }
  s_sum[threadIdx.x] = s;
  s_cnt[threadIdx.x] = c;
  __syncthreads();
  if (live && k == 0) {
    int bk = 0;
    float bs = s;
    for (int kk = 1; kk < K; ++kk) {
      const float v = s_sum[a * KP + kk];
      if (v < bs) {
        bs = v;
        bk = kk;
      }
    }
    best[i] = bk;
    minsum[i] = bs;
    cnt[i] = c;
  }
}

__device__ __forceinline__ void k_loss_finalize_body(const float* __restrict__ minsum, const int32_t* __restrict__ cnt, int N,
                                                      float* __restrict__ scal, int per_step);
// Laplace NLL of the winning mode (losses/laplace_nll_loss.py:29-44): per actor the sum over its valid steps of
// log(2 s) + |y - l| / s for both coordinates, s = max(scale, eps); overwrites minsum (the winner is already chosen)
__global__ void k_nll_value(const float* __restrict__ loc, const float* __restrict__ y, const uint8_t* __restrict__ mask,
                            const int32_t* __restrict__ best, int N, int T, float eps, float* __restrict__ minsum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float s = 0.f;
  const uint8_t* mk = mask + int64_t(i) * T;
  const float2* yy = reinterpret_cast<const float2*>(y) + int64_t(i) * T;
  const f4* ll = reinterpret_cast<const f4*>(loc) + (int64_t(best[i]) * N + i) * T;
  for (int t0 = 0; t0 < T; t0 += 8) {                    // eight steps' loads in flight, summed in step order
    uint8_t m[8];
    float2 yv[8];
    f4 lv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u < T ? t0 + u : T - 1;
      m[u] = mk[t];
      yv[u] = yy[t];
      lv[u] = ll[t];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (t0 + u >= T || !m[u]) continue;
      const f4 l = lv[u];
      const float sx = fmaxf(l[2], eps), sy = fmaxf(l[3], eps);
      s += logf(2.f * sx) + fabsf(yv[u].x - l[0]) / sx;
      s += logf(2.f * sy) + fabsf(yv[u].y - l[1]) / sy;
    }
  }
  minsum[i] = s;
}

// scal[0] = loss = sum(minsum) / count, scal[1] = 1/count (0 when nothing is valid); fixed summation order
__global__ __launch_bounds__(1024) void k_l2_finalize(const float* __restrict__ minsum, const int32_t* __restrict__ cnt, int N,
                                                      float* __restrict__ scal) {
  k_loss_finalize_body(minsum, cnt, N, scal, 1);
}
// the same with `per_step` loss elements per valid step (the Laplace NLL averages over the x and the y term: 2)
__global__ __launch_bounds__(1024) void k_nll_finalize(const float* __restrict__ minsum, const int32_t* __restrict__ cnt, int N,
                                                       float* __restrict__ scal) {
  k_loss_finalize_body(minsum, cnt, N, scal, 2);
}
__device__ __forceinline__ void k_loss_finalize_body(const float* __restrict__ minsum, const int32_t* __restrict__ cnt, int N,
                                                      float* __restrict__ scal, int per_step) {
  __shared__ double ssum[1024];
  __shared__ long long scnt[1024];
  double s = 0.0;
  long long c = 0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    s += double(minsum[i]);
    c += cnt[i];
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = c;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) {
      ssum[threadIdx.x] += ssum[threadIdx.x + w];
      scnt[threadIdx.x] += scnt[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double denom = double(scnt[0]) * per_step;
    scal[0] = scnt[0] > 0 ? float(ssum[0] / denom) : 0.f;
    scal[1] = scnt[0] > 0 ? float(1.0 / denom) : 0.f;
  }
}

// ------------------------------------------------------------------ forward replay of the winning paths
__global__ __launch_bounds__(128) void k_init_sel(const float* __restrict__ img, const float* __restrict__ local,
                                                  const float* __restrict__ global, const int32_t* __restrict__ best, int N,
                                                  float* __restrict__ y0, float* __restrict__ gsel) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, InitBwdL::AE_END);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    keep_lds_reads_here();
    const int row = tile * 16 + L.n;
    const int i = row < N ? row : N - 1;
    f4 gl[4], lo[4], a[4];
    load_row(gl, global, int64_t(best[i]) * N + i, L.g);
    load_row(lo, local, i, L.g);
    load_vec<4>(a, lds + InitBwdL::BA, L.g);
    linear_acc<4, 4>(a, gl, lds + InitBwdL::WA_G, L.lane);
    linear_acc<4, 4>(a, lo, lds + InitBwdL::WA_L, L.lane);
    layer_norm<4>(a, lds + InitBwdL::AG, lds + InitBwdL::AE, L.g);
    relu<4>(a);
    if (row < N) {
      store_row(a, y0, row, L.g);
      store_row(gl, gsel, row, L.g);
    }
  }
}

// states [n_euler+1][N][64] (slab 0 = y0 on entry); H1,H2,G1,G2 [n_euler][N][64]; GS [n_euler][N]
__global__ __launch_bounds__(128) void k_sde_replay(const float* __restrict__ img, const int32_t* __restrict__ best, int N, int K,
                                                    int n_euler, const float* __restrict__ step_tab, NoiseArg na,
                                                    float* __restrict__ states, float* __restrict__ H1, float* __restrict__ H2,
                                                    float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ GS) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, DecSdeL::LOC);                      // drift + diffusion images
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  const int64_t slab = int64_t(N) * D;
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    const int row = tile * 16 + L.n;
    const int i = row < N ? row : N - 1;
    const int64_t r = int64_t(best[i]) * N + i;
    f4 y[4];
    load_row(y, states, i, L.g);
    for (int k = 0; k < n_euler; ++k) {
      keep_lds_reads_here();
      const float dt = step_tab[k * 8 + 1], sq = step_tab[k * 8 + 2], sn = step_tab[k * 8 + 3], cs = step_tab[k * 8 + 4];
      const float* F = lds + DecSdeL::F;
      const float* G = lds + DecSdeL::G;
      f4 h1[4], h2[4], f[4], z[4];
      sde_layer0(h1, y, F, DriftL::W0, DriftL::WS, DriftL::WC, DriftL::B0, sn, cs, L);
      tanh_<4>(h1);
      linear<4, 4>(h2, h1, F + DriftL::W2, F + DriftL::B2, L);
      tanh_<4>(h2);
      linear<4, 4>(f, h2, F + DriftL::W4, F + DriftL::B4, L);
      if (row < N) {
        store_row(h1, H1 + k * slab, row, L.g);
        store_row(h2, H2 + k * slab, row, L.g);
      }
      sde_layer0(h1, y, G, DiffL::W0, DiffL::WS, DiffL::WC, DiffL::B0, sn, cs, L);
      tanh_<4>(h1);
      linear<4, 4>(h2, h1, G + DiffL::W2, G + DiffL::B2, L);
      tanh_<4>(h2);
      const float gs = fast_sigmoid(row_dot(h2, G + DiffL::W4, L.g) + G[DiffL::B4]);
      if (row < N) {
        store_row(h1, G1 + k * slab, row, L.g);
        store_row(h2, G2 + k * slab, row, L.g);
        if (L.g == 0) GS[int64_t(k) * N + row] = gs;
      }
      noise_row(z, na, STREAM_DECODER, k, r, int64_t(N) * K, L.g);
      em_update(y, f, gs, z, dt, sq);
      if (row < N) store_row(y, states + (k + 1) * slab, row, L.g);
    }
  }
}

// ------------------------------------------------------------------ loc head: forward + backward per (row, output)
// vector-gradient slots of one wave in `vpart` (floats)
struct HeadV { enum : int { DGAM = 0, DBET = 64, DW3X = 128, DW3Y = 192, DB3 = 256, SIZE = 264 }; };

// rows are (o, i): o = output step, i = actor.  S_in / DU / DS are [T][N][64]
// MODE 0: the loc head under the winner-takes-all L2 loss.  MODE 1 / 2: the loc / the scale head under the Laplace NLL
// (losses/laplace_nll_loss.py): `nll` carries the forward's outputs of the winning mode (the other head's value enters each
// head's upstream gradient); the scale head's launch ADDS its state gradient to the loc head's (DS) and writes its own delta rows.
struct NllArg {
  const float* loc;        // [K, N, T, 4] forward outputs
  const int32_t* best;     // winning mode per actor
  float eps, min_scale;
};
template <int MODE>
__global__ __launch_bounds__(128) void k_head_bwd(const float* __restrict__ img, const float* __restrict__ states,
                                                  const float* __restrict__ out_tab, const float* __restrict__ y,
                                                  const uint8_t* __restrict__ mask, const float* __restrict__ scal, int N, int T,
                                                  float* __restrict__ S_in, float* __restrict__ DU, float* __restrict__ DS,
                                                  float* __restrict__ vpart, NllArg nll) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, HeadBwdL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int tiles_per_o = (N + 15) / 16;
  const int ntiles = tiles_per_o * T;
  const int64_t slab = int64_t(N) * D;
  const float inv_count = scal[1];
  const float* H = lds + HeadBwdL::FWD;
  f4 dgam[4], dbet[4], dw3x[4], dw3y[4];
  zero4(dgam); zero4(dbet); zero4(dw3x); zero4(dw3y);
  float db3x = 0.f, db3y = 0.f;
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    keep_lds_reads_here();
    const int o = tile / tiles_per_o;
    const int row = (tile - o * tiles_per_o) * 16 + L.n;
    const int i = row < N ? row : N - 1;
    const int ko = int(out_tab[o * 4]);
    const float w0 = out_tab[o * 4 + 1], w1 = out_tab[o * 4 + 2];
    f4 s[4], u[4], v[4];
    {
      f4 a[4], b[4];
      load_row(a, states + (ko - 1) * slab, i, L.g);
      load_row(b, states + ko * slab, i, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[jt][c] = w0 * a[jt][c] + w1 * b[jt][c];
    }
    linear<4, 4>(u, s, H + HeadL::W0, H + HeadL::B0, L);
    const float rstd = ln_normalize(u);                     // u = x_hat
    bool pos[16];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 ga = *reinterpret_cast<const f4*>(H + HeadL::G + 16 * jt + 4 * L.g);
      const f4 be = *reinterpret_cast<const f4*>(H + HeadL::E + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pre = u[jt][c] * ga[c] + be[c];
        pos[jt * 4 + c] = pre > 0.f;
        v[jt][c] = fmaxf(pre, 0.f);
      }
    }
    const float lx = row_dot(v, H + HeadL::W3, L.g) + H[HeadL::B3];
    const float ly = row_dot(v, H + HeadL::W3 + 64, L.g) + H[HeadL::B3 + 1];
    float gx = 0.f, gy = 0.f;
    if (row < N && mask[int64_t(i) * T + o]) {
      const float yx = y[(int64_t(i) * T + o) * 2], yy = y[(int64_t(i) * T + o) * 2 + 1];
      if (MODE == 0) {
        // dL/dl = (l - y) / |l - y| / count on valid steps (L2.py:16,25)
        const float dx = lx - yx, dy = ly - yy;
        const float nrm = sqrtf(dx * dx + dy * dy);
        if (nrm > 0.f) {
          gx = dx / nrm * inv_count;
          gy = dy / nrm * inv_count;
        }
      } else {
        const f4 fw = *reinterpret_cast<const f4*>(nll.loc + ((int64_t(nll.best[i]) * N + i) * T + o) * 4);
        if (MODE == 1) {
          // d/dl [ |y - l| / s ] = -sign(y - l) / s, s = max(scale, eps) of the forward (no gradient through the clamp's value)
          const float sx = fmaxf(fw[2], nll.eps), sy = fmaxf(fw[3], nll.eps);
          const float ex = yx - lx, ey = yy - ly;
          gx = (ex > 0.f ? -1.f : ex < 0.f ? 1.f : 0.f) / sx * inv_count;
          gy = (ey > 0.f ? -1.f : ey < 0.f ? 1.f : 0.f) / sy * inv_count;
        } else {
          // this head's two outputs are the raw scales: s = ELU(raw) + 1 + min_scale (DEC:97-98), clamped at eps in place, the
          // gradient passing through (laplace_nll_loss.py:38-40); d/ds [ log 2s + |y - l| / s ] = 1/s - |y - l| / s^2
          const float sxr = (lx > 0.f ? lx : fast_exp(lx) - 1.0f) + 1.0f + nll.min_scale;
          const float syr = (ly > 0.f ? ly : fast_exp(ly) - 1.0f) + 1.0f + nll.min_scale;
          const float sx = fmaxf(sxr, nll.eps), sy = fmaxf(syr, nll.eps);
          const float ax = fabsf(yx - fw[0]), ay = fabsf(yy - fw[1]);
          gx = (1.0f / sx - ax / (sx * sx)) * inv_count * (lx > 0.f ? 1.0f : fast_exp(lx));
          gy = (1.0f / sy - ay / (sy * sy)) * inv_count * (ly > 0.f ? 1.0f : fast_exp(ly));
        }
      }
    }
    db3x += gx;
    db3y += gy;
    f4 dv[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 wx = *reinterpret_cast<const f4*>(H + HeadL::W3 + 16 * jt + 4 * L.g);
      const f4 wy = *reinterpret_cast<const f4*>(H + HeadL::W3 + 64 + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        dw3x[jt][c] = fmaf(gx, v[jt][c], dw3x[jt][c]);
        dw3y[jt][c] = fmaf(gy, v[jt][c], dw3y[jt][c]);
        dv[jt][c] = pos[jt * 4 + c] ? fmaf(gx, wx[c], gy * wy[c]) : 0.f;
      }
    }
    ln_backward(dv, u, rstd, H + HeadL::G, L.g, dgam, dbet);      // dv := du
    f4 ds[4];
    linear_t(ds, dv, lds + HeadBwdL::W0T, L);
    if (row < N) {
      if (MODE == 2) {                                       // the second head of the step: its state gradient joins the first's
        f4 prev[4];
        load_row(prev, DS + o * slab, row, L.g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) ds[jt] += prev[jt];
      } else {
        store_row(s, S_in + o * slab, row, L.g);
      }
      store_row(dv, DU + o * slab, row, L.g);
      store_row(ds, DS + o * slab, row, L.g);
    }
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * HeadV::SIZE;
  flush_vec(dgam, vp + HeadV::DGAM, L);
  flush_vec(dbet, vp + HeadV::DBET, L);
  flush_vec(dw3x, vp + HeadV::DW3X, L);
  flush_vec(dw3y, vp + HeadV::DW3Y, L);
  flush_scalar(db3x, vp + HeadV::DB3, L);
  flush_scalar(db3y, vp + HeadV::DB3 + 1, L);
}

// ------------------------------------------------------------------ reverse sweep through the Euler-Maruyama steps
struct SweepV { enum : int { DV4 = 0, DC4 = 64, SIZE = 68 }; };

__global__ __launch_bounds__(128, 2) void k_sde_bwd(const float* __restrict__ img, const int32_t* __restrict__ best, int N, int K, int T,
                                                 int n_euler, const float* __restrict__ step_tab, const float* __restrict__ out_tab,
                                                 NoiseArg na, const float* __restrict__ H1, const float* __restrict__ H2,
                                                 const float* __restrict__ G1, const float* __restrict__ G2,
                                                 const float* __restrict__ GS, const float* __restrict__ DS,
                                                 float* __restrict__ DH1, float* __restrict__ DH2, float* __restrict__ DF,
                                                 float* __restrict__ DG1, float* __restrict__ DG2, float* __restrict__ DY0,
                                                 float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, SweepL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  const int64_t slab = int64_t(N) * D;
  f4 dv4[4];
  zero4(dv4);
  float dc4 = 0.f;
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    const int row = tile * 16 + L.n;
    const int i = row < N ? row : N - 1;
    const bool live = row < N;
    const int64_t r = int64_t(best[i]) * N + i;
    f4 dy[4];                                             // dL/dy_{k+1} on entry of iteration k
    zero4(dy);
    int o = T - 1;
    // The saved activation tiles of an iteration used to be loaded right where they are consumed, behind a matrix product they do
    // not depend on: four exposed round trips to HBM per iteration of a kernel that runs one wave per SIMD on a third of the chip (384
    // tiles at 128 x 48 agents).  Now the two that are consumed FIRST (the last layers' activations) and the diffusion value are
    // requested one iteration ahead (32 registers), the other two at the top of their iteration, a matrix product ahead of their use.
    f4 nh2[4], ng2[4];
    float ngs;
    {
      const int k0 = n_euler - 1;
      load_row(nh2, H2 + k0 * slab, i, L.g);
      load_row(ng2, G2 + k0 * slab, i, L.g);
      ngs = GS[int64_t(k0) * N + i];
    }
    for (int k = n_euler - 1; k >= 0; --k) {
      keep_lds_reads_here();
      const float dt = step_tab[k * 8 + 1], sq = step_tab[k * 8 + 2];
      f4 ah2[4], ah1[4], ag2[4], ag1[4];
      load_row(ah1, H1 + k * slab, i, L.g);
      load_row(ag1, G1 + k * slab, i, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) { ah2[jt] = nh2[jt]; ag2[jt] = ng2[jt]; }
      const float gs = ngs;
      if (k > 0) {
        load_row(nh2, H2 + (k - 1) * slab, i, L.g);
        load_row(ng2, G2 + (k - 1) * slab, i, L.g);
        ngs = GS[int64_t(k - 1) * N + i];
      }
      // outputs interpolated between y_k and y_{k+1}: s_o = w0 y_k + w1 y_{k+1}
      f4 dprev[4];
      zero4(dprev);
      while (o >= 0 && int(out_tab[o * 4]) == k + 1) {
        const float w0 = out_tab[o * 4 + 1], w1 = out_tab[o * 4 + 2];
        f4 ds[4];
        load_row(ds, DS + o * slab, i, L.g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            dy[jt][c] = fmaf(w1, ds[jt][c], dy[jt][c]);
            dprev[jt][c] = fmaf(w0, ds[jt][c], dprev[jt][c]);
          }
        --o;
      }
      f4 d[4], t[4];
      // ---- drift net: y' gets f*dt
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = dt * dy[jt][c];
      if (live) store_row(d, DF + k * slab, row, L.g);
      linear_t(t, d, lds + SweepL::F_W4T, L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - ah2[jt][c] * ah2[jt][c]);
      if (live) store_row(d, DH2 + k * slab, row, L.g);
      linear_t(t, d, lds + SweepL::F_W2T, L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - ah1[jt][c] * ah1[jt][c]);
      if (live) store_row(d, DH1 + k * slab, row, L.g);
      f4 dyn[4];                                          // dL/dy_k being assembled
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) dyn[jt] = dy[jt] + dprev[jt];
      linear_adj<4, 4>(dyn, d, lds + SweepL::F_W0T, L);
      // ---- diffusion net: y' gets g * (z sqrt(h)), g one scalar per row
      f4 z[4];
      noise_row(z, na, STREAM_DECODER, k, r, int64_t(N) * K, L.g);
      float cdot = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) cdot = fmaf(z[jt][c] * sq, dy[jt][c], cdot);
      const float dgp = row_sum(cdot) * gs * (1.0f - gs);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f4 w4 = *reinterpret_cast<const f4*>(lds + SweepL::G_W4 + 16 * jt + 4 * L.g);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (live) dv4[jt][c] = fmaf(dgp, ag2[jt][c], dv4[jt][c]);
          d[jt][c] = dgp * w4[c] * (1.0f - ag2[jt][c] * ag2[jt][c]);
        }
      }
      if (live) {
        dc4 += dgp;
        store_row(d, DG2 + k * slab, row, L.g);
      }
      linear_t(t, d, lds + SweepL::G_W2T, L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - ag1[jt][c] * ag1[jt][c]);
      if (live) store_row(d, DG1 + k * slab, row, L.g);
      linear_adj<4, 4>(dyn, d, lds + SweepL::G_W0T, L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) dy[jt] = dyn[jt];
    }
    if (live) store_row(dy, DY0, row, L.g);
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * SweepV::SIZE;
  flush_vec(dv4, vp + SweepV::DV4, L);
  flush_scalar(dc4, vp + SweepV::DC4, L);
}

// ------------------------------------------------------------------ aggr_embed backward

__global__ __launch_bounds__(128) void k_dec_init_bwd(const float* __restrict__ img, const float* __restrict__ local,
                                                      const float* __restrict__ gsel, const float* __restrict__ DY0,
                                                      const int32_t* __restrict__ best, int N, float* __restrict__ DA,
                                                      float* __restrict__ d_local, float* __restrict__ d_global,
                                                      float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, InitBwdL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  f4 dgam[4], dbet[4];
  zero4(dgam); zero4(dbet);
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    keep_lds_reads_here();
    const int row = tile * 16 + L.n;
    const int i = row < N ? row : N - 1;
    f4 gl[4], lo[4], a[4], d[4];
    load_row(gl, gsel, i, L.g);
    load_row(lo, local, i, L.g);
    load_vec<4>(a, lds + InitBwdL::BA, L.g);
    linear_acc<4, 4>(a, gl, lds + InitBwdL::WA_G, L.lane);
    linear_acc<4, 4>(a, lo, lds + InitBwdL::WA_L, L.lane);
    const float rstd = ln_normalize(a);
    load_row(d, DY0, i, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 ga = *reinterpret_cast<const f4*>(lds + InitBwdL::AG + 16 * jt + 4 * L.g);
      const f4 be = *reinterpret_cast<const f4*>(lds + InitBwdL::AE + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!(a[jt][c] * ga[c] + be[c] > 0.f) || row >= N) d[jt][c] = 0.f;
    }
    ln_backward(d, a, rstd, lds + InitBwdL::AG, L.g, dgam, dbet);   // d := d a_pre
    f4 t[4];
    if (row < N) store_row(d, DA, row, L.g);
    linear_t(t, d, lds + InitBwdL::WA_GT, L);
    if (row < N) store_row(t, d_global, int64_t(best[i]) * N + i, L.g);
    linear_t(t, d, lds + InitBwdL::WA_LT, L);
    if (row < N) store_row(t, d_local, row, L.g);
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * InitV::SIZE;
  flush_vec(dgam, vp + InitV::DGAM, L);
  flush_vec(dbet, vp + InitV::DBET, L);
}

// ------------------------------------------------------------------ weight gradients from saved rows
// part[p] = sum_{rows of chunk p} delta[r][:]^T a[r][:]  (64x64, [o][i]),  cs[p][o] = sum delta[r][o].
// Chunks never straddle a group (= one Euler step of rows_per_group rows), so the reducer can weight them per step.
__global__ __launch_bounds__(256, 3) void k_wgrad(WgradJobs jobs, int64_t R, int64_t rows_per_group, int chunk, int chunks_per_group, int P,
                                               float* __restrict__ part, float* __restrict__ cs) {
  const WgradJob& job = jobs.j[blockIdx.y];
  const float* __restrict__ delta = job.delta;
  const float* __restrict__ a = job.a;
  const int ldd = job.ldd, lda = job.lda;
  part += int64_t(blockIdx.y) * P * 4096;
  cs += int64_t(blockIdx.y) * P * 64;
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int p = blockIdx.x;
  const int group = p / chunks_per_group, sub = p - group * chunks_per_group;
  const int64_t row0 = group * rows_per_group + int64_t(sub) * chunk;
  int64_t row1 = row0 + chunk;
  if (row1 > (group + 1) * rows_per_group) row1 = (group + 1) * rows_per_group;
  if (row1 > R) row1 = R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, idx = lane & 15, kg = lane >> 4;
  // Wave `ot` owns output rows 16 ot .. 16 ot + 15 of the 64 x 64 block and walks ALL 4-row k-steps of a staged block: 16
  // accumulator registers per lane instead of 64 (every wave holding the whole block and taking every 4th k-step), so four
  // workgroups fit a CU between the barriers instead of two, and no cross-wave reduction at the end.
  const int ot = wave;
  f4 acc[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) acc[it] = f4{0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  // 64-row blocks of delta and a are staged in LDS TRANSPOSED -- [feature][row], the row index XOR-ed with a per-feature multiple
  // of 4 -- so that a lane's 16 operand values of a block are 4 aligned 16-byte reads, all issued before the block's 64 matrix
  // instructions (reading them one k-step at a time put an LDS round trip in front of every pair of matrix instructions: the
  // pipe was 60 % busy).  The contraction index is permuted to make that possible: k-step j sums rows {j, 16+j, 32+j, 48+j} of the
  // block (lane group kg holds rows 16 kg .. 16 kg + 15), the same permutation for both operands.  The XOR term 4 ((f & 15) ^ (f >> 4))
  // spreads the 16 features a read instruction touches over all banks, and the 64 scalar writes of a wave over all 32 write banks.
  float* ds_ = dyn;                       // [64 features][64 rows]
  float* as_ = dyn + 4096;                // [64 features][64 rows]
  const int c4w = threadIdx.x & 15;
  int wofs[4];                            // this thread's four features: f * 64, with the swizzle term kept apart in wswz
  int wswz[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = 4 * c4w + j;
    wofs[j] = f * 64;
    wswz[j] = 4 * ((f & 15) ^ (f >> 4));
  }
  const float* a_rd = ds_ + (16 * ot + idx) * 64;           // operand reads: feature rows of this lane
  const int a_sw = 4 * (idx ^ ot);
  // software pipeline: the global loads (or the computed operand) of block k+1 are issued before the matrix work of block
  // k, so their latency hides behind it; registers -> LDS happens after the barrier that retires block k's reads
  f4 dreg[4], areg[4];
  const bool computed = job.in2 != nullptr;             // uniform per launch slice (blockIdx.y)
  const int pair = job.pair;
  // the computed operand is built from the row's geometry AFTER the barrier that opens the next block (`finish`), not where the
  // geometry is fetched: its consumer would otherwise wait out the load right in front of the matrix loop, every block
  auto fetch = [&](int64_t blk) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = threadIdx.x + 256 * u;            // float4 index within the 64 x 16 block
      const int r = f >> 4, c4 = f & 15;
      const int64_t row = blk + r;
      f4 dv = f4{0.f, 0.f, 0.f, 0.f}, av = dv;
      if (row < row1) {
        dv = *reinterpret_cast<const f4*>(delta + row * ldd + 4 * c4);
        av = computed ? *reinterpret_cast<const f4*>(a + row * 4) : *reinterpret_cast<const f4*>(a + row * lda + 4 * c4);
      }
      dreg[u] = dv;
      areg[u] = av;
    }
  };
  auto finish = [&](int64_t blk) {                      // in2_rstd / in2_ln_relu4 (tile.hpp)
    // a thread always produces the same four features (4 * (threadIdx.x & 15)); their closed-form constants are re-read per block
    // (cache hits) rather than held in 24 registers across the matrix loop
    const int f0 = 4 * (threadIdx.x & 15);
    const f4 kw0 = *reinterpret_cast<const f4*>(job.in2 + f0), kw1 = *reinterpret_cast<const f4*>(job.in2 + 64 + f0);
    const f4 kgb = *reinterpret_cast<const f4*>(job.in2 + 128 + f0), kbe = *reinterpret_cast<const f4*>(job.beta + f0);
    const f4 kc0 = *reinterpret_cast<const f4*>(job.in2 + 192), kc1 = *reinterpret_cast<const f4*>(job.in2 + 196);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = (threadIdx.x + 256 * u) >> 4;
      const f4 ge = areg[u];
      const float x0 = pair ? ge[2] : ge[0], x1 = pair ? ge[3] : ge[1];
      const float ca = fmaf(kc0[0], x0, fmaf(kc0[1], x1, kc0[2])), cb = fmaf(kc0[3], x1, kc1[0]);
      const float rstd = rsqrt_nr(fmaf(ca, ca, fmaf(cb, cb, kc1[1] * kc1[1])) + 1e-5f);
      const float x0r = x0 * rstd, x1r = x1 * rstd;
      f4 av;
#pragma unroll
      for (int k = 0; k < 4; ++k) av[k] = fmaxf(fmaf(kw0[k], x0r, fmaf(kw1[k], x1r, fmaf(kgb[k], rstd, kbe[k]))), 0.f);
      areg[u] = blk + r < row1 ? av : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (row0 < row1) fetch(row0);
  for (int64_t blk = row0; blk < row1; blk += 64) {
    __syncthreads();
    if (computed) finish(blk);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = (threadIdx.x + 256 * u) >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ds_[wofs[j] + (r ^ wswz[j])] = dreg[u][j];
        as_[wofs[j] + (r ^ wswz[j])] = areg[u][j];
      }
    }
    __syncthreads();
    if (blk + 64 < row1) fetch(blk + 64);
    f4 A[4], B[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) A[m] = *reinterpret_cast<const f4*>(a_rd + ((16 * kg + 4 * m) ^ a_sw));
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int m = 0; m < 4; ++m) B[q][m] = *reinterpret_cast<const f4*>(as_ + (16 * q + idx) * 64 + ((16 * kg + 4 * m) ^ (4 * (idx ^ q))));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        csum += A[m][c];
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][c], B[it][m][c], acc[it], 0, 0, 0);
      }
  }
  // D fragment: lane holds dW[16 ot + 4 kg + reg][16 it + idx]
  float* out = part + int64_t(p) * 4096;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out[(16 * ot + 4 * kg + reg) * 64 + 16 * it + idx] = acc[it][reg];
  csum += __shfl_xor(csum, 16);                             // the four k-groups of column 16 ot + idx
  csum += __shfl_xor(csum, 32);
  if (kg == 0) cs[int64_t(p) * 64 + 16 * ot + idx] = csum;
}

#if TSDE_SPLIT_H3
// ---- the same partial sums on the 16-bit matrix cores (fp16x3: a_h b_h + a_h b_l + a_l b_h, tile.hpp), the default in this build.
// The fp32 matrix instruction runs at 1/16 of the 16-bit rate: at 8 192 flops per row the exact kernel above is bound by it (0.85 ms
// of matrix pipe for the 4.55 M edge rows of a 64 x 128 step, the same as streaming their 4.8 GB); three 16-bit products take a fifth of
// that and leave the kernel to HBM.  Both operands of a block are scaled by a power of two taken from the block's largest magnitude (deltas
// are tiny, fp16 has 5 exponent bits: the same reason tile.hpp linear_adj scales rows), split into hi / lo halves when the block is
// staged, and the block's product is scaled back when it joins the fp32 accumulator.  The contraction runs over ROWS, which sit on the
// lanes' row index in global memory order: the planes are staged row-major ([64 rows][64 halves], 8-byte chunks XOR-swizzled) and read
// through ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of four rows -- an operand fragment, no transposing writes
// (lane map checked on the hardware: tools/microbench/trread.hip).
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ int wg6_off(int r, int c) {      // byte offset of chunk c (4 halves) of row r: conflict-free for the stores and the transposed reads
  return r * 128 + 8 * (c ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 2));
}
__device__ __forceinline__ h8 wg6_frag(const char* plane, int off0, int off1) {
  const s4v x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(plane + off0));
  const s4v y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(plane + off1));
  const uint2 a = __builtin_bit_cast(uint2, x), b = __builtin_bit_cast(uint2, y);
  return __builtin_bit_cast(h8, u4{a.x, a.y, b.x, b.y});
}
// 2^(14 - floor(log2 m)) and its inverse for a block whose largest magnitude is m (0 / 0 for an all-zero or sub-2^-113 block)
__device__ __forceinline__ void wg6_scale(float m, float& up, float& down) {
  const unsigned e = __float_as_uint(m) & 0x7F800000u;
  const bool ok = e >= (14u << 23);
  up = ok ? __uint_as_float(0x86000000u - e) : 0.f;
  down = ok ? __uint_as_float(e - (14u << 23)) : 0.f;
}
#ifndef TSDE_WG6_OCC
#define TSDE_WG6_OCC 3
#endif
// One workgroup's rows [row0, row1) of one problem -- or, DUAL, of the TWO computed-operand problems of an edge embedding (branch A from
// geometry columns 0-1, branch B from columns 2-3), which contract the same delta rows: the delta planes are staged once and every
// delta fragment feeds both products (the pair read the 256-byte delta row twice as separate problems: 24 % of the launch's bytes).
// Per problem the arithmetic is that of the single form: same block scales, same products, same order.
template <bool DUAL>
__device__ __forceinline__ void wgrad6_rows(const WgradJob& job, const WgradJob& jobB, int64_t row0, int64_t row1, float* __restrict__ out,
                                            float* __restrict__ outB, float* __restrict__ csout, float* __restrict__ csoutB, char* smem) {
  const float* __restrict__ delta = job.delta;
  const float* __restrict__ a = job.a;
  const int ldd = job.ldd, lda = job.lda;
  char* const dh = smem;                               // [64 rows][64 halves] planes: delta hi / lo, a hi / lo (, the second a hi / lo)
  char* const dl = smem + 8192;
  char* const ah = smem + 16384;
  char* const al = smem + 24576;
  char* const bh = smem + 32768;
  char* const bl = smem + 40960;
  constexpr int NM = DUAL ? 3 : 2;                     // block maxima per wave
  float* const slots = reinterpret_cast<float*>(smem + (DUAL ? 49152 : 32768));      // [parity][wave][NM]
  const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6, idx = lane & 15, kg = lane >> 4;
  f4 acc[4], acc2[4], csum4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 4; ++it) acc[it] = acc2[it] = f4{0.f, 0.f, 0.f, 0.f};
  f4 dreg[4], areg[4], breg[4];
  const bool computed = DUAL || job.in2 != nullptr;
  const int c4w = threadIdx.x & 15, r0w = threadIdx.x >> 4;
  auto fetch = [&](int64_t blk) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t row = blk + r0w + 16 * u;
      f4 dv = f4{0.f, 0.f, 0.f, 0.f}, av = dv;
      if (row < row1) {
        dv = *reinterpret_cast<const f4*>(delta + row * ldd + 4 * c4w);
        av = computed ? *reinterpret_cast<const f4*>(a + row * 4) : *reinterpret_cast<const f4*>(a + row * lda + 4 * c4w);
      }
      dreg[u] = dv;
      areg[u] = av;
    }
  };
  // the computed operand from the row's geometry record `ge` (see k_wgrad): ReLU(LN(Linear(2, 64))) in its closed form
  auto finish = [&](const WgradJob& jb, f4 (&dst)[4], int64_t blk) {
    const int f0 = 4 * c4w, pair = jb.pair;
    const f4 kw0 = *reinterpret_cast<const f4*>(jb.in2 + f0), kw1 = *reinterpret_cast<const f4*>(jb.in2 + 64 + f0);
    const f4 kgb = *reinterpret_cast<const f4*>(jb.in2 + 128 + f0), kbe = *reinterpret_cast<const f4*>(jb.beta + f0);
    const f4 kc0 = *reinterpret_cast<const f4*>(jb.in2 + 192), kc1 = *reinterpret_cast<const f4*>(jb.in2 + 196);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f4 ge = areg[u];
      const float x0 = pair ? ge[2] : ge[0], x1 = pair ? ge[3] : ge[1];
      const float ca = fmaf(kc0[0], x0, fmaf(kc0[1], x1, kc0[2])), cb = fmaf(kc0[3], x1, kc1[0]);
      const float rstd = rsqrt_nr(fmaf(ca, ca, fmaf(cb, cb, kc1[1] * kc1[1])) + 1e-5f);
      const float x0r = x0 * rstd, x1r = x1 * rstd;
      f4 av;
#pragma unroll
      for (int k = 0; k < 4; ++k) av[k] = fmaxf(fmaf(kw0[k], x0r, fmaf(kw1[k], x1r, fmaf(kgb[k], rstd, kbe[k]))), 0.f);
      dst[u] = blk + r0w + 16 * u < row1 ? av : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // this lane's fragment addresses: rows 32 ks + 8 kg + 4 half + q, chunk 4 * (16-column block) + p   (q = idx >> 2, p = idx & 3)
  const int fq = idx >> 2, fp = idx & 3;
  if (row0 < row1) fetch(row0);
  for (int64_t blk = row0; blk < row1; blk += 64) {
    if constexpr (DUAL) finish(jobB, breg, blk);        // (before areg's geometry is overwritten in place)
    if (computed) finish(job, areg, blk);
    float md = 0.f, ma = 0.f, mb = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        md = fmaxf(md, fabsf(dreg[u][c]));
        ma = fmaxf(ma, fabsf(areg[u][c]));
        if constexpr (DUAL) mb = fmaxf(mb, fabsf(breg[u][c]));
      }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      md = fmaxf(md, __shfl_xor(md, sft));
      ma = fmaxf(ma, __shfl_xor(ma, sft));
      if constexpr (DUAL) mb = fmaxf(mb, __shfl_xor(mb, sft));
    }
    float* const sl = slots + 4 * NM * (int((blk - row0) >> 6) & 1);  // two sets by block parity: a set is rewritten two barriers after its last read
    if (lane == 0) {
      sl[NM * ot] = md;
      sl[NM * ot + 1] = ma;
      if constexpr (DUAL) sl[NM * ot + 2] = mb;
    }
    __syncthreads();                                    // every wave is done with the previous block's planes; the maxima are visible
    md = fmaxf(fmaxf(sl[0], sl[NM]), fmaxf(sl[2 * NM], sl[3 * NM]));
    ma = fmaxf(fmaxf(sl[1], sl[NM + 1]), fmaxf(sl[2 * NM + 1], sl[3 * NM + 1]));
    if constexpr (DUAL) mb = fmaxf(fmaxf(sl[2], sl[NM + 2]), fmaxf(sl[2 * NM + 2], sl[3 * NM + 2]));
    float up_d, down_d, up_a, down_a, up_b = 0.f, down_b = 0.f;
    wg6_scale(md, up_d, down_d);
    wg6_scale(ma, up_a, down_a);
    if constexpr (DUAL) wg6_scale(mb, up_b, down_b);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int off = wg6_off(r0w + 16 * u, c4w);
      const f4 dv = dreg[u] * up_d, av = areg[u] * up_a;
      unsigned h0, l0, h1, l1;
      split_pair(dv[0], dv[1], h0, l0);
      split_pair(dv[2], dv[3], h1, l1);
      *reinterpret_cast<uint2*>(dh + off) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(dl + off) = uint2{l0, l1};
      split_pair(av[0], av[1], h0, l0);
      split_pair(av[2], av[3], h1, l1);
      *reinterpret_cast<uint2*>(ah + off) = uint2{h0, h1};
      *reinterpret_cast<uint2*>(al + off) = uint2{l0, l1};
      if constexpr (DUAL) {
        const f4 bv = breg[u] * up_b;
        split_pair(bv[0], bv[1], h0, l0);
        split_pair(bv[2], bv[3], h1, l1);
        *reinterpret_cast<uint2*>(bh + off) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(bl + off) = uint2{l0, l1};
      }
      csum4 += dreg[u];
    }
    __syncthreads();
    if (blk + 64 < row1) fetch(blk + 64);
    f4 accb[4], accb2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) accb[it] = accb2[it] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int rA = 32 * ks + 8 * kg + fq;
      const int oa0 = wg6_off(rA, 4 * ot + fp), oa1 = wg6_off(rA + 4, 4 * ot + fp);
      const h8 Ah = wg6_frag(dh, oa0, oa1), Al = wg6_frag(dl, oa0, oa1);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ob0 = wg6_off(rA, 4 * it + fp), ob1 = wg6_off(rA + 4, 4 * it + fp);
        const h8 Bh = wg6_frag(ah, ob0, ob1), Bl = wg6_frag(al, ob0, ob1);
        accb[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, accb[it], 0, 0, 0);
        accb[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, accb[it], 0, 0, 0);
        accb[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, accb[it], 0, 0, 0);
        if constexpr (DUAL) {
          const h8 Ch = wg6_frag(bh, ob0, ob1), Cl = wg6_frag(bl, ob0, ob1);
          accb2[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Ch, accb2[it], 0, 0, 0);
          accb2[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Cl, accb2[it], 0, 0, 0);
          accb2[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Ch, accb2[it], 0, 0, 0);
        }
      }
    }
    const float down = down_d * down_a, down2 = down_d * down_b;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[it][c] = fmaf(accb[it][c], down, acc[it][c]);
        if constexpr (DUAL) acc2[it][c] = fmaf(accb2[it][c], down2, acc2[it][c]);
      }
  }
  // D fragment: lane holds dW[16 ot + 4 kg + reg][16 it + idx]
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      out[(16 * ot + 4 * kg + reg) * 64 + 16 * it + idx] = acc[it][reg];
      if constexpr (DUAL) outB[(16 * ot + 4 * kg + reg) * 64 + 16 * it + idx] = acc2[it][reg];
    }
  // column sums of delta (exact fp32, from the rows as they were fetched): this thread's four features over its rows -> the block's
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    csum4[c] += __shfl_xor(csum4[c], 16);
    csum4[c] += __shfl_xor(csum4[c], 32);
  }
  __syncthreads();
  f4* red = reinterpret_cast<f4*>(smem);
  if (lane < 16) red[ot * 16 + lane] = csum4;
  __syncthreads();
  if (threadIdx.x < 16) {
    const f4 t = (red[threadIdx.x] + red[16 + threadIdx.x]) + (red[32 + threadIdx.x] + red[48 + threadIdx.x]);
    *reinterpret_cast<f4*>(csout + 4 * threadIdx.x) = t;
    if constexpr (DUAL) *reinterpret_cast<f4*>(csoutB + 4 * threadIdx.x) = t;
  }
}
__global__ __launch_bounds__(256, TSDE_WG6_OCC) void k_wgrad6(WgradJobs jobs, int64_t R, int64_t rows_per_group, int chunk, int chunks_per_group, int P,
                                                   float* __restrict__ part, float* __restrict__ cs) {
  const WgradJob& job = jobs.j[blockIdx.y];
  part += int64_t(blockIdx.y) * P * 4096;
  cs += int64_t(blockIdx.y) * P * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = blockIdx.x;
  const int group = p / chunks_per_group, sub = p - group * chunks_per_group;
  const int64_t row0 = group * rows_per_group + int64_t(sub) * chunk;
  int64_t row1 = row0 + chunk;
  if (row1 > (group + 1) * rows_per_group) row1 = (group + 1) * rows_per_group;
  if (row1 > R) row1 = R;
  wgrad6_rows<false>(job, job, row0, row1, part + int64_t(p) * 4096, nullptr, cs + int64_t(p) * 64, nullptr, smem);
}
// The three problems of an edge embedding's weight gradients over its R rows (node_bwd.hip edge_embed_backward) as one launch of two kinds
// of workgroup: the first P0 reduce `chunk0` rows of problem 0 (two stored operands: 512 bytes per row), the next P1 reduce `chunk1` rows
// of problems 1 AND 2 (wgrad6_rows<true>: 272 bytes per row).  Partials: problem 0 in slots [0, P0), 1 in [P0, P0 + P1), 2 behind.
__global__ __launch_bounds__(256, 3) void k_wgrad6_edge(WgradJobs jobs, int64_t R, int chunk0, int P0, int chunk1, int P1,
                                                        float* __restrict__ part, float* __restrict__ cs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = blockIdx.x;
  if (p < P0) {
    const int64_t row0 = int64_t(p) * chunk0, row1 = row0 + chunk0 < R ? row0 + chunk0 : R;
    wgrad6_rows<false>(jobs.j[0], jobs.j[0], row0, row1, part + int64_t(p) * 4096, nullptr, cs + int64_t(p) * 64, nullptr, smem);
  } else {
    const int q = p - P0;
    const int64_t row0 = int64_t(q) * chunk1, row1 = row0 + chunk1 < R ? row0 + chunk1 : R;
    wgrad6_rows<true>(jobs.j[1], jobs.j[2], row0, row1, part + int64_t(P0 + q) * 4096, part + int64_t(P0 + P1 + q) * 4096,
                      cs + int64_t(P0 + q) * 64, cs + int64_t(P0 + P1 + q) * 64, smem);
  }
}
static bool wgrad_f32() {          // TRAJSDE_WGRAD_F32=1: the exact fp32 kernel (A/B runs)
  static const bool v = []() { const char* e = getenv("TRAJSDE_WGRAD_F32"); return e && atoi(e) != 0; }();
  return v;
}
#else
static bool wgrad_f32() { return true; }
#endif

// W[o*ldw + col0 + i] = sum_p part[p][o][i];  bias[o] = sum_p cs[p][o];  with time_cols the (sin t, cos t) input
// columns 64 / 65 of the 66-wide first SDE layer: W[o*ldw + 64] = sum_p sin(t_group(p)) cs[p][o], likewise cos.
// A workgroup owns 32 outputs; its 8 thread groups each sum every 8th partial, then combine in a fixed order.
__global__ __launch_bounds__(256) void k_reduce_partials(WgradJobs jobs, const float* __restrict__ part, const float* __restrict__ cs, int P,
                                                         int chunks_per_group, const float* __restrict__ step_tab) {
  const WgradJob& job = jobs.j[blockIdx.y];
  float* __restrict__ W = job.W;
  float* __restrict__ bias = job.bias;
  const int ldw = job.ldw, col0 = job.col0, time_cols = job.time_cols;
  part += int64_t(blockIdx.y) * P * 4096;
  cs += int64_t(blockIdx.y) * P * 64;
  __shared__ float red[3][8][32];
  const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + lane;
  float s = 0.f, ws = 0.f, wc = 0.f;
  if (j < 4096) {
    int p = sl;                                         // sixteen loads in flight; the additions in the order of the plain loop
    for (; p + 120 < P; p += 128) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[int64_t(p + 8 * u) * 4096 + j];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; p + 24 < P; p += 32) {
      const float v0 = part[int64_t(p) * 4096 + j], v1 = part[int64_t(p + 8) * 4096 + j];
      const float v2 = part[int64_t(p + 16) * 4096 + j], v3 = part[int64_t(p + 24) * 4096 + j];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; p < P; p += 8) s += part[int64_t(p) * 4096 + j];
  } else if (j < 4096 + 64) {
    const int o = j - 4096;
    // (eight partials in flight, consumed in the order of the plain loop: these 64 threads a problem walked P / 8 dependent loads
    //  and were the launch's critical path)
    for (int p0 = sl; p0 < P; p0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p0 + 8 * u < P ? cs[int64_t(p0 + 8 * u) * 64 + o] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 8 * u;
        if (p >= P) break;
        s += v[u];
        if (time_cols) {
          const int k = p / chunks_per_group;
          ws = fmaf(step_tab[k * 8 + 3], v[u], ws);
          wc = fmaf(step_tab[k * 8 + 4], v[u], wc);
        }
      }
    }
  }
  red[0][sl][lane] = s;
  red[1][sl][lane] = ws;
  red[2][sl][lane] = wc;
  __syncthreads();
  if (sl != 0) return;
  float t[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int g = 0; g < 8; ++g) t[q] += red[q][g][lane];
  if (j < 4096) {
    W[(j >> 6) * ldw + col0 + (j & 63)] = t[0];
  } else if (j < 4096 + 64) {
    const int o = j - 4096;
    if (bias) bias[o] = t[0];
    if (time_cols) {
      W[o * ldw + 64] = t[1];
      W[o * ldw + 65] = t[2];
    }
  }
}

// dst[j*dst_stride] = sum_w src[w*stride + j], j < n   (per-wave vector partials -> one vector), for up to COLSUM_MAX_JOBS
// vectors cut from the same slab of partials in one launch (grid.y = job).  One workgroup per 64 columns, 16 row slices per
// workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void k_colsum(ColsumJobs jobs, int64_t rows, int stride) {
  __shared__ float red[16][64];
  const ColsumJob& job = jobs.j[blockIdx.y];
  const float* __restrict__ src = job.src;
  const int n = job.n;
  if (blockIdx.x * 64 >= n) return;                   // (uniform) a narrower job of the same launch
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float s = 0.f;
  if (j < n) {
    // eight independent partial sums keep several loads in flight (a single workgroup streams the whole slab)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t w = part;
    for (; w + 112 < rows; w += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += src[(w + 16 * u) * stride + j];
    }
    for (; w < rows; w += 16) a[0] += src[w * stride + j];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][c];
    job.dst[int64_t(j) * job.dst_stride] = t;
  }
}


// ------------------------------------------------------------------ deferred sums (bwd.hpp)
// k_reduce_partials with per-problem partial runs: grid.y = problem, slots [base, base + P) of the shared partial buffer
__global__ __launch_bounds__(256) void k_reduce_partials_q(ReduceJobs jobs, const float* __restrict__ part, const float* __restrict__ cs,
                                                           const float* __restrict__ step_tab) {
  const ReduceJob& job = jobs.j[blockIdx.y];
  float* __restrict__ W = job.W;
  float* __restrict__ bias = job.bias;
  const int ldw = job.ldw, col0 = job.col0, time_cols = job.time_cols, P = job.P, chunks_per_group = job.cpg;
  part += job.base * 4096;
  cs += job.base * 64;
  __shared__ float red[3][8][32];
  const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + lane;
  float s = 0.f, ws = 0.f, wc = 0.f;
  if (j < 4096) {
    int p = sl;                                         // sixteen loads in flight; the additions in the order of the plain loop
    for (; p + 120 < P; p += 128) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[int64_t(p + 8 * u) * 4096 + j];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; p + 24 < P; p += 32) {
      const float v0 = part[int64_t(p) * 4096 + j], v1 = part[int64_t(p + 8) * 4096 + j];
      const float v2 = part[int64_t(p + 16) * 4096 + j], v3 = part[int64_t(p + 24) * 4096 + j];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; p < P; p += 8) s += part[int64_t(p) * 4096 + j];
  } else if (j < 4096 + 64 && (bias || time_cols)) {
    const int o = j - 4096;
    // (eight partials in flight, consumed in the order of the plain loop: these 64 threads a problem walked P / 8 dependent loads
    //  and were the launch's critical path)
    for (int p0 = sl; p0 < P; p0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p0 + 8 * u < P ? cs[int64_t(p0 + 8 * u) * 64 + o] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 8 * u;
        if (p >= P) break;
        s += v[u];
        if (time_cols) {
          const int k = p / chunks_per_group;
          ws = fmaf(step_tab[k * 8 + 3], v[u], ws);
          wc = fmaf(step_tab[k * 8 + 4], v[u], wc);
        }
      }
    }
  }
  red[0][sl][lane] = s;
  red[1][sl][lane] = ws;
  red[2][sl][lane] = wc;
  __syncthreads();
  if (sl != 0) return;
  float t[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int g = 0; g < 8; ++g) t[q] += red[q][g][lane];
  if (j < 4096) {
    W[(j >> 6) * ldw + col0 + (j & 63)] = t[0];
  } else if (j < 4096 + 64) {
    const int o = j - 4096;
    if (bias) bias[o] = t[0];
    if (time_cols) {
      W[o * ldw + 64] = t[1];
      W[o * ldw + 65] = t[2];
    }
  }
}
// k_colsum with per-vector slabs (rows, stride): grid.y = vector
__global__ __launch_bounds__(1024) void k_colsum_q(ColsumQJobs jobs) {
  __shared__ float red[16][64];
  const ColsumQJob& job = jobs.j[blockIdx.y];
  const float* __restrict__ src = job.src;
  const int n = job.n, stride = job.stride;
  const int64_t rows = job.rows;
  if (blockIdx.x * 64 >= n) return;                   // (uniform) a narrower vector of the same launch
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float s = 0.f;
  if (j < n) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t w = part;
    for (; w + 112 < rows; w += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += src[(w + 16 * u) * stride + j];
    }
    for (; w < rows; w += 16) a[0] += src[w * stride + j];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][c];
    job.dst[int64_t(j) * job.dst_stride] = t;
  }
}

ReduceQueue*& active_reduce_queue() {
  static thread_local ReduceQueue* q = nullptr;
  return q;
}
ColsumQueue*& active_colsum_queue() {
  static thread_local ColsumQueue* q = nullptr;
  return q;
}
int ReduceQueue::drain() {
  for (size_t first = 0; first < jobs.size(); first += REDUCE_MAX_JOBS) {
    ReduceJobs sub;
    sub.n = int(jobs.size() - first < size_t(REDUCE_MAX_JOBS) ? jobs.size() - first : size_t(REDUCE_MAX_JOBS));
    for (int i = 0; i < sub.n; ++i) sub.j[i] = jobs[first + i];
    TS_LAUNCH(k_reduce_partials_q, dim3(cdiv(4096 + 64, 32), sub.n), 256, 0, st, sub, part, cs, step_tab);
  }
  jobs.clear();
  used = 0;
  return TRAJSDE_OK;
}
int64_t ReduceQueue::take(int64_t slots, int* rc) {
  *rc = TRAJSDE_OK;
  if (used + slots > cap) *rc = drain();
  const int64_t base = used;
  used += slots;
  return base;
}
int ColsumQueue::drain() {
  for (size_t first = 0; first < jobs.size(); first += COLSUMQ_MAX_JOBS) {
    ColsumQJobs sub;
    sub.n = int(jobs.size() - first < size_t(COLSUMQ_MAX_JOBS) ? jobs.size() - first : size_t(COLSUMQ_MAX_JOBS));
    int widest = 0;
    for (int i = 0; i < sub.n; ++i) {
      sub.j[i] = jobs[first + i];
      widest = sub.j[i].n > widest ? sub.j[i].n : widest;
    }
    TS_LAUNCH(k_colsum_q, dim3(cdiv(widest, 64), sub.n), 1024, 0, st, sub);
  }
  jobs.clear();
  used = 0;
  return TRAJSDE_OK;
}
float* ColsumQueue::take(int64_t floats) {
  floats = (floats + 63) / 64 * 64;
  if (floats > cap) return nullptr;
  if (used + floats > cap && drain() != TRAJSDE_OK) return nullptr;
  float* p = arena + used;
  used += floats;
  return p;
}
DeferredSums::DeferredSums(hipStream_t st, float* part, float* cs, int64_t cap, const float* step_tab, float* arena, int64_t arena_floats) {
  // TRAJSDE_REDUCE_CAP / TRAJSDE_VPART_ARENA (partial slots / arena floats): smaller areas than the workspace has, so that tests reach
  // the sum-early-when-full paths at sizes where the real areas never fill
  static const int64_t cap_env = []() { const char* e = getenv("TRAJSDE_REDUCE_CAP"); return e ? atoll(e) : 0; }();
  static const int64_t arena_env = []() { const char* e = getenv("TRAJSDE_VPART_ARENA"); return e ? atoll(e) : 0; }();
  if (cap_env > 0 && cap_env < cap) cap = cap_env;
  if (arena_env > 0 && arena_env < arena_floats) arena_floats = arena_env;
  rq.st = st; rq.part = part; rq.cs = cs; rq.step_tab = step_tab; rq.cap = cap; rq.used = 0;
  cq.st = st; cq.arena = arena; cq.cap = arena_floats; cq.used = 0;
  static const bool off = []() { const char* e = getenv("TRAJSDE_IMMEDIATE_SUMS"); return e && e[0] == '1'; }();   // A/B switch
  if (off) return;
  active_reduce_queue() = &rq;
  active_colsum_queue() = arena ? &cq : nullptr;
}
DeferredSums::~DeferredSums() {
  if (active_reduce_queue() == &rq) active_reduce_queue() = nullptr;
  if (active_colsum_queue() == &cq) active_colsum_queue() = nullptr;
}
int DeferredSums::finish() {
  if (int rc = cq.drain()) return rc;
  return rq.drain();
}
float* vpart_slab(float* shared_slab, int64_t rows, int stride) {
  ColsumQueue* q = active_colsum_queue();
  if (!q) return shared_slab;
  float* p = q->take(rows * stride);
  if (p) return p;
  // larger than the arena: sum what is queued (the shared slab may be one of its sources) and fall back to the shared slab, whose
  // own sum then runs immediately (ColsumBatch::flush sees a source outside the arena)
  q->drain();
  return shared_slab;
}

int ColsumBatch::add(const float* src, int n, float* dst, int dst_stride) {
  if (jobs.n == COLSUM_MAX_JOBS)
    if (int rc = flush()) return rc;
  jobs.j[jobs.n++] = ColsumJob{src, dst, n, dst_stride};
  return TRAJSDE_OK;
}
int ColsumBatch::flush() {
  if (jobs.n == 0) return TRAJSDE_OK;
  if (ColsumQueue* q = active_colsum_queue()) {
    bool inside = true;                               // deferred only for slabs that live in the queue's arena
    for (int i = 0; i < jobs.n; ++i) inside = inside && jobs.j[i].src >= q->arena && jobs.j[i].src < q->arena + q->cap;
    if (inside) {
      for (int i = 0; i < jobs.n; ++i) q->jobs.push_back(ColsumQJob{jobs.j[i].src, jobs.j[i].dst, rows, jobs.j[i].n, stride, jobs.j[i].dst_stride});
      jobs.n = 0;
      return TRAJSDE_OK;
    }
  }
  int widest = 0;
  for (int i = 0; i < jobs.n; ++i) widest = jobs.j[i].n > widest ? jobs.j[i].n : widest;
  TS_LAUNCH(k_colsum, dim3(cdiv(widest, 64), jobs.n), 1024, 0, st, jobs, rows, stride);
  jobs.n = 0;
  return TRAJSDE_OK;
}

int WgradBatch::add(const float* delta, int ldd, const float* a, int lda, float* W, int ldw, int col0, float* bias, int time_cols) {
  if (jobs.n == WGRAD_MAX_JOBS)
    if (int rc = flush()) return rc;
  jobs.j[jobs.n++] = WgradJob{delta, a, W, bias, ldd, lda, ldw, col0, time_cols, nullptr, nullptr, 0};
  return TRAJSDE_OK;
}
int WgradBatch::add_in2(const float* delta, int ldd, const float* geom, int pair, const float* in2, const float* beta, float* W, int ldw,
                        float* bias) {
  if (jobs.n == WGRAD_MAX_JOBS)
    if (int rc = flush()) return rc;
  jobs.j[jobs.n++] = WgradJob{delta, geom, W, bias, ldd, 4, ldw, 0, 0, in2, beta, pair};
  return TRAJSDE_OK;
}

#if TSDE_SPLIT_H3 && defined(TSDE_WG6_OCC)
#define TSDE_WG6_OCC_HOST TSDE_WG6_OCC
#else
#define TSDE_WG6_OCC_HOST 3
#endif
static int launch_wgrad(const char* tag, const WgradJobs& sub, int64_t R, int64_t rows_per_group, int chunk, int cpg, int P, float* part, float* cs,
                        hipStream_t st) {
  static const bool trace = getenv("TRAJSDE_WGRAD_TRACE") != nullptr;       // one line per launch: its shape
  if (trace) fprintf(stderr, "wgrad %s: R=%lld jobs=%d P=%d chunk=%d groups=%lld\n", tag, (long long)R, sub.n, P, chunk, (long long)((R + rows_per_group - 1) / rows_per_group));
#if TSDE_SPLIT_H3
  if (!wgrad_f32()) {
    TS_LAUNCH_TAG(tag, false, k_wgrad6, dim3(P, sub.n), 256, 32768 + 64, st, sub, R, rows_per_group, chunk, cpg, P, part, cs);
    return TRAJSDE_OK;
  }
#endif
  TS_LAUNCH_TAG(tag, false, k_wgrad, dim3(P, sub.n), 256, 2 * 4096 * 4, st, sub, R, rows_per_group, chunk, cpg, P, part, cs);
  return TRAJSDE_OK;
}

int WgradBatch::flush() {
  if (jobs.n == 0) return TRAJSDE_OK;
  const int n = jobs.n;
  jobs.n = 0;
  if (R <= 0) {   // nothing to sum: the gradient blocks are zero
    WgradJobs all = jobs;
    all.n = n;
    TS_LAUNCH(k_reduce_partials, dim3(cdiv(4096 + 64, 32), n), 256, 0, c.st, all, c.part, c.cs, 0, 1, c.step_tab);
    return TRAJSDE_OK;
  }
  // rows per workgroup: at least WGRAD_CHUNK; enough partials to fill the chip several times over (a workgroup walks its
  // rows 64 at a time with a barrier in between), few enough (<= ~1024) that the second stage stays short
  const int groups = int((R + rows_per_group - 1) / rows_per_group);
  int64_t chunk = WGRAD_CHUNK;
  static const int parts_env = []() { const char* e = getenv("TRAJSDE_WGRAD_PARTS"); return e ? atoi(e) : 0; }();
  // One resident round of the chip: 3 workgroups of this kernel fit a CU (40 KB of LDS each), and a workgroup streams its
  // rows at the same rate however many it has, so 768 workgroups over the launch's problems leave no partial last round and
  // the fewest partials to reduce (3 problems: 1024 partials each 1.41 ms, 384 1.54 ms, 256 1.35 ms, 128 1.93 ms)
  const int64_t one_round = ((wgrad_f32() ? 768 : 256 * TSDE_WG6_OCC_HOST) + n - 1) / n;
  const int64_t base_parts = parts_env > 0 ? parts_env : (one_round > 32 ? one_round : 32);
  const int64_t want_parts = groups > base_parts ? groups : base_parts;
  if ((rows_per_group + chunk - 1) / chunk * groups > want_parts) {
    // the smallest multiple of 64 rows that stays within the partial budget: P lands just under it (1024 = one full round of
    // the chip's 4 x 256 resident workgroups per problem), not at whatever a doubling of the chunk happens to give
    const int64_t per_group = want_parts / groups > 0 ? want_parts / groups : 1;
    chunk = ((rows_per_group + per_group - 1) / per_group + 63) / 64 * 64;
  } else if (groups == 1) {
    // a short problem (the aggregator's node-level layers: 8 192 rows at 64 x 128): 512-row chunks leave 16 workgroups a problem, each
    // walking 8 blocks one after the other on a mostly idle chip -- shorter chunks, up to one resident round of workgroups
    static const int min_chunk = []() { const char* e = getenv("TRAJSDE_WGRAD_MIN_CHUNK"); const int v = e ? atoi(e) : 128; return v < 64 ? 64 : (v + 63) / 64 * 64; }();
    const int64_t fill = ((R + one_round - 1) / one_round + 63) / 64 * 64;
    const int64_t shorter = fill > min_chunk ? fill : min_chunk;
    if (shorter < chunk) chunk = shorter;
  }
  const int cpg = int((rows_per_group + chunk - 1) / chunk);
  const int P = cpg * groups;
  int per_launch = int(c.cap / P);
  if (per_launch < 1) return fail(TRAJSDE_ERR_WORKSPACE, "wgrad: partial buffer too small");
  ReduceQueue* rq = active_reduce_queue();
  if (rq && rq->part != c.part) rq = nullptr;           // (a context over another partial buffer: immediate)
  if (rq && rq->cap < c.cap) {                          // a queue over a smaller area than the context's
    per_launch = int(rq->cap / P);
    if (per_launch < 1) {                               // one problem's partials do not fit it: this batch is summed immediately,
      if (int rc = rq->drain()) return rc;              // from slot 0 of the same buffer -- after what is queued there
      rq = nullptr;
      per_launch = int(c.cap / P);
    }
  }
  for (int first = 0; first < n; first += per_launch) {
    WgradJobs sub;
    sub.n = n - first < per_launch ? n - first : per_launch;
    for (int i = 0; i < sub.n; ++i) sub.j[i] = jobs.j[first + i];
    if (rq) {                                           // partials stay in their slots; summed at DeferredSums::finish (or when full)
      int rc = TRAJSDE_OK;
      const int64_t base = rq->take(int64_t(sub.n) * P, &rc);
      if (rc) return rc;
      if (int rc2 = launch_wgrad(tag, sub, R, rows_per_group, int(chunk), cpg, P, c.part + base * 4096, c.cs + base * 64, c.st)) return rc2;
      for (int i = 0; i < sub.n; ++i)
        rq->jobs.push_back(ReduceJob{sub.j[i].W, sub.j[i].bias, base + int64_t(i) * P, P, cpg, sub.j[i].ldw, sub.j[i].col0, sub.j[i].time_cols});
      continue;
    }
    if (int rc2 = launch_wgrad(tag, sub, R, rows_per_group, int(chunk), cpg, P, c.part, c.cs, c.st)) return rc2;
    TS_LAUNCH(k_reduce_partials, dim3(cdiv(4096 + 64, 32), sub.n), 256, 0, c.st, sub, c.part, c.cs, P, cpg, c.step_tab);
  }
  return TRAJSDE_OK;
}

// The edge embedding's batch -- problem 0 over two stored operands, problems 1 and 2 over the same delta and geometry records -- through
// k_wgrad6_edge; anything else (the exact fp32 kernel, another shape of batch, TRAJSDE_WGRAD_EDGE_PAIR=0) through flush().
int WgradBatch::flush_edge() {
#if TSDE_SPLIT_H3
  static const bool on = []() { const char* e = getenv("TRAJSDE_WGRAD_EDGE_PAIR"); return !(e && e[0] == '0'); }();
  static const int p0_env = []() { const char* e = getenv("TRAJSDE_WGRAD_EDGE_P0"); return e ? atoi(e) : 0; }();
  static const int p1_env = []() { const char* e = getenv("TRAJSDE_WGRAD_EDGE_P1"); return e ? atoi(e) : 0; }();
  const WgradJob &j0 = jobs.j[0], &j1 = jobs.j[1], &j2 = jobs.j[2];
  const bool shape = jobs.n == 3 && R >= 64 * 64 && rows_per_group >= R && !j0.in2 && j1.in2 && j2.in2 && j1.delta == j2.delta &&
                     j1.ldd == j2.ldd && j1.a == j2.a && !j0.time_cols && !j1.time_cols && !j2.time_cols;
  if (!on || wgrad_f32() || !shape) return flush();
  // One resident round of the chip (3 workgroups a CU): a CU gets one workgroup of the first kind and two of the second.  The pair
  // workgroups are bound by vector arithmetic (two closed-form operands, three splits a block), not by their 272 bytes a row: measured
  // at 64 x 128, 4.55 M rows -- 500 + 268 workgroups 1.30 ms, 384 + 384 0.99, 256 + 512 0.85-0.88, 200 + 568 0.86, 256 + 1024 0.89;
  // the three separate problems (flush) 1.03 ms.
  const int want0 = p0_env > 0 ? p0_env : 256, want1 = p1_env > 0 ? p1_env : 512;
  const int64_t chunk0 = ((R + want0 - 1) / want0 + 63) / 64 * 64, chunk1 = ((R + want1 - 1) / want1 + 63) / 64 * 64;
  const int P0 = int((R + chunk0 - 1) / chunk0), P1 = int((R + chunk1 - 1) / chunk1);
  const int64_t slots = int64_t(P0) + 2 * int64_t(P1);
  ReduceQueue* rq = active_reduce_queue();
  if (rq && (rq->part != c.part || rq->cap < slots)) rq = nullptr;
  if (slots > c.cap) return flush();
  WgradJobs sub = jobs;
  jobs.n = 0;
  int64_t base = 0;
  if (rq) {
    int rc = TRAJSDE_OK;
    base = rq->take(slots, &rc);
    if (rc) return rc;
  } else if (ReduceQueue* other = active_reduce_queue()) {
    if (other->part == c.part)
      if (int rc = other->drain()) return rc;            // summed immediately from slot 0 of the same buffer: after what is queued there
  }
  TS_LAUNCH_TAG(tag, false, k_wgrad6_edge, P0 + P1, 256, 49152 + 128, c.st, sub, R, int(chunk0), P0, int(chunk1), P1, c.part + base * 4096,
                c.cs + base * 64);
  ReduceJobs rj;
  rj.n = 3;
  rj.j[0] = ReduceJob{j0.W, j0.bias, base, P0, P0, j0.ldw, j0.col0, 0};
  rj.j[1] = ReduceJob{sub.j[1].W, sub.j[1].bias, base + P0, P1, P1, sub.j[1].ldw, sub.j[1].col0, 0};
  rj.j[2] = ReduceJob{sub.j[2].W, sub.j[2].bias, base + P0 + P1, P1, P1, sub.j[2].ldw, sub.j[2].col0, 0};
  if (rq) {
    for (int i = 0; i < 3; ++i) rq->jobs.push_back(rj.j[i]);
    return TRAJSDE_OK;
  }
  TS_LAUNCH(k_reduce_partials_q, dim3(cdiv(4096 + 64, 32), 3), 256, 0, c.st, rj, c.part, c.cs, c.step_tab);
  return TRAJSDE_OK;
#else
  return flush();
#endif
}

int run_wgrad(const WgradCtx& c, const float* delta, int ldd, const float* a, int lda, int64_t R, int64_t rows_per_group, float* W,
              int ldw, int col0, float* bias, int time_cols) {
  WgradBatch b(c, R, rows_per_group);
  if (int rc = b.add(delta, ldd, a, lda, W, ldw, col0, bias, time_cols)) return rc;
  return b.flush();
}
// a tall slab (rows >> the 16 row slices of one workgroup): `slices` workgroups per 64 columns each sum a contiguous share of the
// rows into scratch[slice][n], a second launch sums the slices -- fixed order, no atomics
__global__ __launch_bounds__(1024) void k_colsum_slices(const float* __restrict__ src, int64_t rows, int stride, int n, float* __restrict__ scratch) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  float s = 0.f;
  if (j < n) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t w = lo + part;
    for (; w + 48 < hi; w += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += src[(w + 16 * u) * stride + j];
    }
    for (; w < hi; w += 16) a[0] += src[w * stride + j];
    s = (a[0] + a[1]) + (a[2] + a[3]);
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][c];
    scratch[int64_t(blockIdx.y) * n + j] = t;
  }
}
int run_colsum_tall(hipStream_t st, const float* src, int64_t rows, int stride, int n, float* dst, float* scratch /* 256 x n floats */) {
  if (rows < 2048) return run_colsum(st, src, rows, stride, n, dst, 1);      // (a single workgroup streams a short slab fast enough)
  const int slices = 256;
  TS_LAUNCH(k_colsum_slices, dim3(cdiv(n, 64), slices), 1024, 0, st, src, rows, stride, n, scratch);
  return run_colsum(st, scratch, slices, n, n, dst, 1);
}
int run_colsum(hipStream_t st, const float* src, int64_t rows, int stride, int n, float* dst, int dst_stride) {
  ColsumBatch b(st, rows, stride);
  if (int rc = b.add(src, n, dst, dst_stride)) return rc;
  return b.flush();
}

}  // namespace tsde

using namespace tsde;

namespace {

// gradient slots, in the order of trajsde_param_name(TRAJSDE_STAGE_DECODER_BWD, i)  (pack.hip recipe_decoder_bwd)
enum GradSlot {
  F0W = 0, F2W, F4W, G0W, G2W, G4W, D0W, D0B, D1W, D1B, D3W, D3B, A0W, A0B, A1W, A1B, F0B, F2B, F4B, G0B, G2B, G4B, N_GRADS,
  // TRAJSDE_STAGE_DECODER_NLL_BWD: the same table followed by the scale head (pack.hip recipe_decoder_nll_bwd)
  S0W = N_GRADS, S0B, S1W, S1B, S3W, S3B, N_GRADS_NLL
};
constexpr int BWD_THREADS = 128;

struct BwdWs {
  int32_t *best, *cnt;
  float *minsum, *scal, *states, *H1, *H2, *G1, *G2, *GS, *DH1, *DH2, *DF, *DG1, *DG2, *S_in, *DU, *DS, *gsel, *DA, *DY0, *part, *cs,
      *vpart, *DU2, *varena;
  int64_t bytes, parts, varena_floats;
};

BwdWs carve_bwd(void* ws, int64_t ws_bytes, int N, int T, int n_euler, bool& ok, bool nll = false) {
  Carver cv(ws, ws_bytes);
  BwdWs w;
  const int64_t slab = int64_t(N) * 64;
  w.best = cv.take<int32_t>(N);
  w.cnt = cv.take<int32_t>(N);
  w.minsum = cv.take<float>(N);
  w.scal = cv.take<float>(4);
  w.states = cv.take<float>(slab * (n_euler + 1));
  w.H1 = cv.take<float>(slab * n_euler);
  w.H2 = cv.take<float>(slab * n_euler);
  w.G1 = cv.take<float>(slab * n_euler);
  w.G2 = cv.take<float>(slab * n_euler);
  w.GS = cv.take<float>(int64_t(N) * n_euler);
  w.DH1 = cv.take<float>(slab * n_euler);
  w.DH2 = cv.take<float>(slab * n_euler);
  w.DF = cv.take<float>(slab * n_euler);
  w.DG1 = cv.take<float>(slab * n_euler);
  w.DG2 = cv.take<float>(slab * n_euler);
  w.S_in = cv.take<float>(slab * T);
  w.DU = cv.take<float>(slab * T);
  w.DS = cv.take<float>(slab * T);
  w.gsel = cv.take<float>(slab);
  w.DA = cv.take<float>(slab);
  w.DY0 = cv.take<float>(slab);
  const int64_t max_rows = int64_t(N) * (n_euler > T ? n_euler : T);
  const int64_t max_parts = w.parts = wgrad_max_parts(max_rows, n_euler > T ? n_euler : T);
  w.part = cv.take<float>(max_parts * 4096);
  w.cs = cv.take<float>(max_parts * 64);
  w.vpart = cv.take<float>(int64_t(256) * (BWD_THREADS / 64) * 512);
  w.varena_floats = VPART_ARENA_SLABS * int64_t(256) * (BWD_THREADS / 64) * 512;
  w.varena = cv.take<float>(w.varena_floats);
  w.DU2 = nll ? cv.take<float>(slab * T) : nullptr;        // the scale head's delta rows (Laplace NLL)
  w.bytes = cv.off + 256;
  ok = cv.ok;
  return w;
}

int bwd_grid(int ntiles) {
  const int waves = BWD_THREADS / 64;
  const int g = (ntiles + waves - 1) / waves;
  return g < 1 ? 1 : (g > 256 ? 256 : g);
}

}  // namespace

extern "C" {

int64_t trajsde_decoder_backward_ws_bytes(int32_t N, int num_modes, int future_steps, int n_euler) {
  (void)num_modes;
  bool ok;
  return carve_bwd(nullptr, 0, N, future_steps, n_euler, ok).bytes;
}

int64_t trajsde_decoder_nll_backward_ws_bytes(int32_t N, int num_modes, int future_steps, int n_euler) {
  (void)num_modes;
  bool ok;
  return carve_bwd(nullptr, 0, N, future_steps, n_euler, ok, true).bytes;
}

static int decoder_backward_impl(bool nll, float eps, float min_scale, int32_t N, int num_modes, int future_steps, const float* blob_fwd,
                                 const float* blob_bwd, const float* local_embed, const float* global_embed, const float* step_table,
                                 int n_euler, const float* out_table, const trajsde_noise* noise, const float* loc, const float* y,
                                 const uint8_t* reg_mask, void* ws, int64_t ws_bytes, float* loss, int32_t* best_mode, float* const* grads,
                                 int n_grads, float* d_local, float* d_global, void* stream_);

int trajsde_decoder_l2_backward(int32_t N, int num_modes, int future_steps, const float* blob_fwd, const float* blob_bwd,
                                const float* local_embed, const float* global_embed, const float* step_table, int n_euler,
                                const float* out_table, const trajsde_noise* noise, const float* loc, const float* y,
                                const uint8_t* reg_mask, void* ws, int64_t ws_bytes, float* loss, int32_t* best_mode,
                                float* const* grads, int n_grads, float* d_local, float* d_global, void* stream_) {
  return decoder_backward_impl(false, 0.f, 0.f, N, num_modes, future_steps, blob_fwd, blob_bwd, local_embed, global_embed, step_table, n_euler,
                               out_table, noise, loc, y, reg_mask, ws, ws_bytes, loss, best_mode, grads, n_grads, d_local, d_global, stream_);
}

int trajsde_decoder_nll_backward(int32_t N, int num_modes, int future_steps, const float* blob_fwd, const float* blob_bwd,
                                 const float* local_embed, const float* global_embed, const float* step_table, int n_euler,
                                 const float* out_table, const trajsde_noise* noise, const float* loc, const float* y,
                                 const uint8_t* reg_mask, float eps, float min_scale, void* ws, int64_t ws_bytes, float* loss,
                                 int32_t* best_mode, float* const* grads, int n_grads, float* d_local, float* d_global, void* stream_) {
  TS_REQUIRE(eps > 0.f, "decoder_nll_backward: eps must be positive");
  return decoder_backward_impl(true, eps, min_scale, N, num_modes, future_steps, blob_fwd, blob_bwd, local_embed, global_embed, step_table,
                               n_euler, out_table, noise, loc, y, reg_mask, ws, ws_bytes, loss, best_mode, grads, n_grads, d_local, d_global,
                               stream_);
}

static int decoder_backward_impl(bool nll, float eps, float min_scale, int32_t N, int num_modes, int future_steps, const float* blob_fwd,
                                 const float* blob_bwd, const float* local_embed, const float* global_embed, const float* step_table,
                                 int n_euler, const float* out_table, const trajsde_noise* noise, const float* loc, const float* y,
                                 const uint8_t* reg_mask, void* ws, int64_t ws_bytes, float* loss, int32_t* best_mode, float* const* grads,
                                 int n_grads, float* d_local, float* d_global, void* stream_) {
  TS_REQUIRE(blob_fwd && blob_bwd && local_embed && global_embed && step_table && out_table && loc && y && reg_mask && ws && loss &&
                 grads && d_local && d_global,
             "decoder backward: null pointer");
  TS_REQUIRE(N > 0 && num_modes > 0 && future_steps > 0 && n_euler > 0, "decoder backward: empty problem");
  const int want_grads = nll ? int(N_GRADS_NLL) : int(N_GRADS);
  TS_REQUIRE(n_grads == want_grads, "decoder backward: gradient count does not match trajsde_param_count of the backward stage");
  for (int i = 0; i < want_grads; ++i) TS_REQUIRE(grads[i] != nullptr, "decoder backward: null gradient buffer");
  if (ws_bytes < (nll ? trajsde_decoder_nll_backward_ws_bytes(N, num_modes, future_steps, n_euler)
                      : trajsde_decoder_backward_ws_bytes(N, num_modes, future_steps, n_euler)))
    return fail(TRAJSDE_ERR_WORKSPACE, "decoder backward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  bool ok;
  const int K = num_modes, T = future_steps;
  BwdWs w = carve_bwd(ws, ws_bytes, N, T, n_euler, ok, nll);
  DeferredSums sums(st, w.part, w.cs, w.parts, step_table, w.varena, w.varena_floats);      // every reduction of this call: at the end
  NoiseArg na{0, nullptr, nullptr};
  if (noise) { na.seed = noise->seed; na.z = noise->z; na.row_ids = noise->row_ids; na.seed_dev = noise->seed_dev; }
  const int ntiles = (N + 15) / 16;
  const int waves = BWD_THREADS / 64;
  const int64_t slab = int64_t(N) * 64;

  // ---- loss, winner per actor
  {
    TS_REQUIRE(K >= 1 && K <= 256, "decoder backward: 1 <= num_modes <= 256");
    int KP = 1;
    while (KP < K) KP <<= 1;
    TS_LAUNCH(k_l2_wta, cdiv(N, 256 / KP), 256, 0, st, loc, y, reg_mask, N, K, T, w.best, w.minsum, w.cnt, KP);     // the winner is the L2 one in both losses
  }
  if (nll) {
    TS_LAUNCH(k_nll_value, cdiv(N, 256), 256, 0, st, loc, y, reg_mask, w.best, N, T, eps, w.minsum);
    TS_LAUNCH(k_nll_finalize, 1, 1024, 0, st, w.minsum, w.cnt, N, w.scal);
  } else {
    TS_LAUNCH(k_l2_finalize, 1, 1024, 0, st, w.minsum, w.cnt, N, w.scal);
  }
  TS_HIP(hipMemcpyAsync(loss, w.scal, sizeof(float), hipMemcpyDeviceToDevice, st));
  if (best_mode) TS_HIP(hipMemcpyAsync(best_mode, w.best, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));

  // ---- replay of the winning paths
  const float* init_img = blob_bwd + DecBwdBlob::INIT;
  TS_LAUNCH(k_init_sel, bwd_grid(ntiles), BWD_THREADS, InitBwdL::AE_END * 4, st, init_img, local_embed, global_embed, w.best, N,
            w.states, w.gsel);
#if TSDE_SPLIT_H3
  // the cooperative form (recur.hip k_sde_replay_coop: four waves a tile, the fused forward kernel's own image); TRAJSDE_REPLAY_COOP=0:
  // the one-wave kernel of this file
  static const bool replay_coop = []() { const char* e = getenv("TRAJSDE_REPLAY_COOP"); return !(e && e[0] == '0'); }();
  if (replay_coop)
    TS_LAUNCH_TAG("k_sde_replay", false, k_sde_replay_coop, ntiles < 8192 ? ntiles : 8192, 256, SDE_REPLAY_COOP_LDS_BYTES, st,
                  blob_fwd + DecBlob::SDE6, w.best, N, K, n_euler, step_table, na, w.states, w.H1, w.H2, w.G1, w.G2, w.GS);
  else
#endif
  TS_LAUNCH(k_sde_replay, bwd_grid(ntiles), BWD_THREADS, DecSdeL::LOC * 4, st, blob_fwd + DecBlob::SDE, w.best, N, K, n_euler,
            step_table, na, w.states, w.H1, w.H2, w.G1, w.G2, w.GS);

  // ---- backward: head, sweep, init
  const int head_grid = bwd_grid(ntiles * T);
  const NllArg na_nll{loc, w.best, eps, min_scale};
  const int head_waves = head_grid * waves;
  float* vp = vpart_slab(w.vpart, head_waves, HeadV::SIZE);
  if (nll)
    TS_LAUNCH(k_head_bwd<1>, head_grid, BWD_THREADS, HeadBwdL::SIZE * 4, st, blob_bwd + DecBwdBlob::HEAD, w.states, out_table, y, reg_mask,
              w.scal, N, T, w.S_in, w.DU, w.DS, vp, na_nll);
  else
    TS_LAUNCH(k_head_bwd<0>, head_grid, BWD_THREADS, HeadBwdL::SIZE * 4, st, blob_bwd + DecBwdBlob::HEAD, w.states, out_table, y, reg_mask,
              w.scal, N, T, w.S_in, w.DU, w.DS, vp, na_nll);
  {
    ColsumBatch cb(st, head_waves, HeadV::SIZE);
    cb.add(vp + HeadV::DGAM, 64, grads[D1W]);
    cb.add(vp + HeadV::DBET, 64, grads[D1B]);
    cb.add(vp + HeadV::DW3X, 128, grads[D3W]);      // rows x, y of decoder.3.weight [2,64]
    cb.add(vp + HeadV::DB3, 2, grads[D3B]);
    if (int rc = cb.flush()) return rc;
  }
  if (nll) {                                               // the scale head (its images follow the L2 blob: DecNllBwdBlob)
    vp = vpart_slab(w.vpart, head_waves, HeadV::SIZE);
    TS_LAUNCH(k_head_bwd<2>, head_grid, BWD_THREADS, HeadBwdL::SIZE * 4, st, blob_bwd + DecNllBwdBlob::HEAD_SC, w.states, out_table, y,
              reg_mask, w.scal, N, T, w.S_in, w.DU2, w.DS, vp, na_nll);
    ColsumBatch cb(st, head_waves, HeadV::SIZE);
    cb.add(vp + HeadV::DGAM, 64, grads[S1W]);
    cb.add(vp + HeadV::DBET, 64, grads[S1B]);
    cb.add(vp + HeadV::DW3X, 128, grads[S3W]);
    cb.add(vp + HeadV::DB3, 2, grads[S3B]);
    if (int rc = cb.flush()) return rc;
  }

  const int sweep_grid = bwd_grid(ntiles);
  static_assert(SweepV::SIZE == SDE_SWEEP_V_FLOATS && SweepV::DV4 == 0 && SweepV::DC4 == 64, "recur.hip k_sde_bwd_coop writes this row");
#if TSDE_SPLIT_H3
  // the cooperative form (recur.hip k_sde_bwd_coop: four waves a tile); TRAJSDE_SWEEP_COOP=0: the one-wave kernel of this file
  static const bool sweep_coop_env = []() { const char* e = getenv("TRAJSDE_SWEEP_COOP"); return !(e && e[0] == '0'); }();
  bool sweep_coop = sweep_coop_env;
#else
  bool sweep_coop = false;
#endif
  // (the cooperative kernel writes one SweepV row per workgroup: when vpart_slab() falls back to the workspace's shared slab -- no
  //  deferred sums active, or the arena full -- that slab bounds the rows; and its tables live in dynamic LDS, so a schedule too long
  //  for it takes the one-wave kernel, whose tables stay in global memory)
  constexpr int64_t SHARED_VPART_FLOATS = int64_t(256) * (BWD_THREADS / 64) * 512;       // DecBwdWs: w.vpart
  constexpr int64_t SWEEP_ROWS_MAX = SHARED_VPART_FLOATS / SweepV::SIZE < 8192 ? SHARED_VPART_FLOATS / SweepV::SIZE : 8192;
  const int64_t coop_lds = int64_t(SDE_BWD_COOP_LDS_BYTES) + int64_t(8 * n_euler + 4 * T) * 4;
  if (sweep_coop && coop_lds > 150 * 1024) sweep_coop = false;
  const int sweep_rows = sweep_coop ? int(ntiles < SWEEP_ROWS_MAX ? ntiles : SWEEP_ROWS_MAX) : sweep_grid * waves;
  TS_REQUIRE(int64_t(sweep_rows) * SweepV::SIZE <= SHARED_VPART_FLOATS, "decoder backward: the sweep's partial rows exceed the shared slab");
  vp = vpart_slab(w.vpart, sweep_rows, SweepV::SIZE);
  if (sweep_coop) {
    const SdeBwdCoopArgs ca{blob_bwd + DecBwdBlob::SWEEP, w.best, N, K, T, n_euler, step_table, out_table, na, w.H1, w.H2, w.G1, w.G2, w.GS,
                            w.DS, w.DH1, w.DH2, w.DF, w.DG1, w.DG2, w.DY0, vp};
    TS_LAUNCH_TAG("k_sde_bwd", false, k_sde_bwd_coop, sweep_rows, 256, int(coop_lds), st, ca);
  } else {
    TS_LAUNCH(k_sde_bwd, sweep_grid, BWD_THREADS, SweepL::SIZE * 4, st, blob_bwd + DecBwdBlob::SWEEP, w.best, N, K, T, n_euler, step_table,
              out_table, na, w.H1, w.H2, w.G1, w.G2, w.GS, w.DS, w.DH1, w.DH2, w.DF, w.DG1, w.DG2, w.DY0, vp);
  }
  {
    ColsumBatch cb(st, sweep_rows, SweepV::SIZE);
    cb.add(vp + SweepV::DV4, 64, grads[G4W]);
    cb.add(vp + SweepV::DC4, 1, grads[G4B]);
    if (int rc = cb.flush()) return rc;
  }

  TS_HIP(hipMemsetAsync(d_global, 0, size_t(K) * N * 64 * sizeof(float), st));
  vp = vpart_slab(w.vpart, int64_t(sweep_grid) * waves, InitV::SIZE);
  TS_LAUNCH(k_dec_init_bwd, sweep_grid, BWD_THREADS, InitBwdL::SIZE * 4, st, init_img, local_embed, w.gsel, w.DY0, w.best, N, w.DA, d_local,
            d_global, vp);
  {
    ColsumBatch cb(st, sweep_grid * waves, InitV::SIZE);
    cb.add(vp + InitV::DGAM, 64, grads[A1W]);
    cb.add(vp + InitV::DBET, 64, grads[A1B]);
    if (int rc = cb.flush()) return rc;
  }

  // ---- weight gradients: (delta rows, input rows, rows, rows per step) -> W (+ column offset), bias, time columns
  const WgradCtx wc{st, w.part, w.cs, step_table, w.parts};
  auto wgrad = [&](const float* delta, const float* a, int64_t R, int64_t rows_per_group, float* W, int ldw, int col0, float* bias,
                   int time_cols) -> int { return run_wgrad(wc, delta, 64, a, 64, R, rows_per_group, W, ldw, col0, bias, time_cols); };
  const int64_t RS = slab / 64 * n_euler, RT = slab / 64 * T;
  int rc;
  {
    WgradBatch sde(wc, RS, N);                              // the five SDE matrices over the same (step, path) rows: one launch pair
    if ((rc = sde.add(w.DH1, 64, w.states, 64, grads[F0W], 66, 0, grads[F0B], 1))) return rc;
    if ((rc = sde.add(w.DH2, 64, w.H1, 64, grads[F2W], 64, 0, grads[F2B], 0))) return rc;
    if ((rc = sde.add(w.DF, 64, w.H2, 64, grads[F4W], 64, 0, grads[F4B], 0))) return rc;
    if ((rc = sde.add(w.DG1, 64, w.states, 64, grads[G0W], 66, 0, grads[G0B], 1))) return rc;
    if ((rc = sde.add(w.DG2, 64, w.G1, 64, grads[G2W], 64, 0, grads[G2B], 0))) return rc;
    if ((rc = sde.flush())) return rc;
  }
  if ((rc = wgrad(w.DU, w.S_in, RT, RT, grads[D0W], 64, 0, grads[D0B], 0))) return rc;
  if (nll && (rc = wgrad(w.DU2, w.S_in, RT, RT, grads[S0W], 64, 0, grads[S0B], 0))) return rc;
  {
    WgradBatch init(wc, N, N);                              // aggr_embed.0 [64,128] = cat(global, local): DEC:82
    if ((rc = init.add(w.DA, 64, w.gsel, 64, grads[A0W], 128, 0, grads[A0B], 0))) return rc;
    if ((rc = init.add(w.DA, 64, local_embed, 64, grads[A0W], 128, 64, nullptr, 0))) return rc;
    if ((rc = init.flush())) return rc;
  }
  return sums.finish();
}

}  // extern "C"
