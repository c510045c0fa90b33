// recur.hip -- the encoder's latent SDE + GRU recurrence (ENC:128-182): 21 iterations of
//   h' = h + f(t0,h)*dt + g(t0,h,nus_mask) * dW          one Euler-Maruyama step (SDEINT:477-485, App. D)
//   h  = GRU_Unit(aa_out[t], h', valid[:,t])             (ODEU:136-152)
// fp32 weights of one iteration are 263 KB, more than the 160 KB of LDS, so an iteration is two launches:
// k_enc_sde_step (f, g_nus, g_argo images: 118 KB) and k_enc_gru_step (GRU image: 149 KB).
// The dual diffusion nets are selected per row; a tile whose 16 rows share the source evaluates one net only.
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "philox.hpp"
#include "sde_funcs.hpp"
#include "tile.hpp"

namespace tsde {

__global__ __launch_bounds__(512) void k_enc_sde_step(const float* __restrict__ img_g, const float* __restrict__ h_in,
                                                      const float* __restrict__ hidden0, int Nt, float dt, float sq, float sn,
                                                      float cs, int idx, int noise_step0, NoiseArg na, const uint8_t* __restrict__ nus,
                                                      const int32_t* __restrict__ eos, const int32_t* __restrict__ pick_slot,
                                                      float* __restrict__ h_ode, float* __restrict__ diff_pick) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, EncSdeL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    f4 y[4], f[4], z[4];
    if (h_in != nullptr) load_row(y, h_in, r, L.g);
    else load_vec<4>(y, hidden0, L.g);                               // ENC:78 learned initial state
    drift_eval(f, y, lds + EncSdeL::F, sn, cs, L);
    const bool is_nus = nus[r] != 0;
    const unsigned long long m = __ballot(is_nus);
    float gs;
    if (m == ~0ull) gs = diff_eval(y, lds + EncSdeL::GN, sn, cs, L);             // ENC:470-482
    else if (m == 0ull) gs = diff_eval(y, lds + EncSdeL::GA, sn, cs, L);
    else {
      const float a = diff_eval(y, lds + EncSdeL::GN, sn, cs, L);
      const float b = diff_eval(y, lds + EncSdeL::GA, sn, cs, L);
      gs = is_nus ? a : b;
    }
    noise_row(z, na, STREAM_ENCODER, noise_step0 + idx, r, Nt, L.g);
    em_update(y, f, gs, z, dt, sq);
    if (row < Nt) {
      store_row(y, h_ode, row, L.g);
      const int slot = pick_slot[r];
      if (diff_pick != nullptr && slot >= 0 && eos[r] == idx) {                               // ENC:171,190-191 diffusion of the kept step
        f4 gv[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) gv[jt] = f4{gs, gs, gs, gs};
        store_row(gv, diff_pick, slot, L.g);
      }
    }
  }
}

__global__ __launch_bounds__(512) void k_enc_gru_step(const float* __restrict__ img_g, const float* __restrict__ h_ode,
                                                      const float* __restrict__ x_t, int Nt, int N, int t, int TT, int idx,
                                                      const uint8_t* __restrict__ pad, const int32_t* __restrict__ orig,
                                                      const int32_t* __restrict__ eos, float* __restrict__ h_out,
                                                      float* __restrict__ local_out, float* __restrict__ latent_t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, EncGruL::SIZE);
  using G = EncGruL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    f4 h[4], x[4], ur[8];
    load_row(h, h_ode, r, L.g);
    load_row(x, x_t, r, L.g);
    load_vec<8>(ur, lds + G::BUR, L.g);
    linear_acc<8, 4>(ur, h, lds + G::WUR_H, L.lane);                  // y_concat = [h, x]
    linear_acc<8, 4>(ur, x, lds + G::WUR_X, L.lane);
    tanh_<8>(ur);
    f4 u1[4] = {ur[0], ur[1], ur[2], ur[3]}, r1[4] = {ur[4], ur[5], ur[6], ur[7]};
    f4 u[4], rg[4], n1[4], nw[4];
    linear<4, 4>(u, u1, lds + G::WU2, lds + G::BU2, L);
    sigmoid_<4>(u);
    linear<4, 4>(rg, r1, lds + G::WR2, lds + G::BR2, L);
    sigmoid_<4>(rg);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) rg[jt] *= h[jt];                   // reset_gate * h_cur
    load_vec<4>(n1, lds + G::BN0, L.g);
    linear_acc<4, 4>(n1, x, lds + G::WN_X, L.lane);                   // combined = [x, r*h]
    linear_acc<4, 4>(n1, rg, lds + G::WN_H, L.lane);
    tanh_<4>(n1);
    linear<4, 4>(nw, n1, lds + G::WN2, lds + G::BN2, L);
    const bool valid = !pad[int64_t(orig[r]) * TT + t];               // maski = actors_mask[:, t]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hn = (1.0f - u[jt][c]) * nw[jt][c] + u[jt][c] * h[jt][c];
        h[jt][c] = valid ? hn : h[jt][c];
      }
    if (row < Nt) {
      store_row(h, h_out, row, L.g);
      if (row < N) {
        if (eos[r] == idx) store_row(h, local_out, row, L.g);         // ENC:187-188 latent_ys[eos_idcs, arange]
        if (latent_t != nullptr) store_row(h, latent_t, row, L.g);
      }
    }
  }
}

// forward_ood (ENC:311-313): outs [S,N,64] -> mean over samples [N,64] and std(0).mean(-1) [N] (unbiased std)
__global__ __launch_bounds__(256) void k_ood_stats(const float* __restrict__ samples, int S, int N, float* __restrict__ mean,
                                                   float* __restrict__ stds) {
  const int lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  float m = 0.f;
  for (int s = 0; s < S; ++s) m += samples[(int64_t(s) * N + row) * 64 + lane];
  m /= float(S);
  float v = 0.f;
  for (int s = 0; s < S; ++s) {
    const float d = samples[(int64_t(s) * N + row) * 64 + lane] - m;
    v += d * d;
  }
  float sd = S > 1 ? sqrtf(v / float(S - 1)) : 0.f;
  mean[row * 64 + lane] = m;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sd += __shfl_xor(sd, o);
  if (lane == 0) stds[row] = sd * (1.0f / 64.0f);
}

}  // namespace tsde
