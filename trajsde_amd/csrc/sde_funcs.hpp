// sde_funcs.hpp -- drift / diffusion networks and the Euler-Maruyama update shared by the encoder
// recurrence (ENC:372-482 via SDEINT:477-485) and the decoder solve (DEC:107-195 via stock torchsde Euler).
#pragma once
#include "layouts.hpp"
#include "tile.hpp"

namespace tsde {

// first SDE layer: 66 -> 64 with the (sin t, cos t) columns folded into the bias
__device__ __forceinline__ void sde_layer0(f4 (&out)[4], const f4 (&y)[4], const float* img, int W0, int WS, int WC, int B0,
                                           float sn, float cs, const Lane& L) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int f0 = 16 * jt + 4 * L.g;
    const f4 b = *reinterpret_cast<const f4*>(img + B0 + f0);
    const f4 s = *reinterpret_cast<const f4*>(img + WS + f0);
    const f4 c = *reinterpret_cast<const f4*>(img + WC + f0);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[jt][e] = fmaf(c[e], cs, fmaf(s[e], sn, b[e]));
  }
  linear_acc<4, 4>(out, y, img + W0, L.lane);
}

// drift f(t, y) (FFunc) -> f[4];  img points at a DriftL image
__device__ __forceinline__ void drift_eval(f4 (&f)[4], const f4 (&y)[4], const float* img, float sn, float cs, const Lane& L) {
  f4 h1[4], h2[4];
  sde_layer0(h1, y, img, DriftL::W0, DriftL::WS, DriftL::WC, DriftL::B0, sn, cs, L);
  tanh_<4>(h1);
  linear<4, 4>(h2, h1, img + DriftL::W2, img + DriftL::B2, L);
  tanh_<4>(h2);
  linear<4, 4>(f, h2, img + DriftL::W4, img + DriftL::B4, L);
}

// diffusion g(t, y) (GFunc): one sigmoid scalar per row, broadcast over the 64 state channels
__device__ __forceinline__ float diff_eval(const f4 (&y)[4], const float* img, float sn, float cs, const Lane& L) {
  f4 h1[4], h2[4];
  sde_layer0(h1, y, img, DiffL::W0, DiffL::WS, DiffL::WC, DiffL::B0, sn, cs, L);
  tanh_<4>(h1);
  linear<4, 4>(h2, h1, img + DiffL::W2, img + DiffL::B2, L);
  tanh_<4>(h2);
  return fast_sigmoid(row_dot(h2, img + DiffL::W4, L.g) + img[DiffL::B4]);
}

// y1 = (y0 + f*dt) + g*(z*sqrt_h)      (Euler.step: y0 + f*dt + g_prod)
__device__ __forceinline__ void em_update(f4 (&y)[4], const f4 (&f)[4], float gs, const f4 (&z)[4], float dt, float sq) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) y[jt][c] = (y[jt][c] + f[jt][c] * dt) + gs * (z[jt][c] * sq);
}

// ---- split-precision (bf16x6, tile.hpp) twins used by the fused decoder kernel
__device__ __forceinline__ void sde_layer0_x6(f4 (&out)[4], const f4 (&y)[4], const float* img, int W0, int WS, int WC, int B0,
                                              float sn, float cs, const Lane& L) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int f0 = 16 * jt + 4 * L.g;
    const f4 b = *reinterpret_cast<const f4*>(img + B0 + f0);
    const f4 s = *reinterpret_cast<const f4*>(img + WS + f0);
    const f4 c = *reinterpret_cast<const f4*>(img + WC + f0);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[jt][e] = fmaf(c[e], cs, fmaf(s[e], sn, b[e]));
  }
  linear_acc_x6<4, 4>(out, y, img + W0, L.lane);
}
__device__ __forceinline__ void drift_eval_x6(f4 (&f)[4], const f4 (&y)[4], const float* img, float sn, float cs, const Lane& L) {
  f4 h1[4], h2[4];
  sde_layer0_x6(h1, y, img, DriftL6::W0, DriftL6::WS, DriftL6::WC, DriftL6::B0, sn, cs, L);
  tanh_<4>(h1);
  linear_x6<4, 4>(h2, h1, img + DriftL6::W2, img + DriftL6::B2, L);
  tanh_<4>(h2);
  linear_x6<4, 4>(f, h2, img + DriftL6::W4, img + DriftL6::B4, L);
}
__device__ __forceinline__ float diff_eval_x6(const f4 (&y)[4], const float* img, float sn, float cs, const Lane& L) {
  f4 h1[4], h2[4];
  sde_layer0_x6(h1, y, img, DiffL6::W0, DiffL6::WS, DiffL6::WC, DiffL6::B0, sn, cs, L);
  tanh_<4>(h1);
  linear_x6<4, 4>(h2, h1, img + DiffL6::W2, img + DiffL6::B2, L);
  tanh_<4>(h2);
  return fast_sigmoid(row_dot(h2, img + DiffL6::W4, L.g) + img[DiffL6::B4]);
}

#if TSDE_SPLIT_H3
// Drift and diffusion on the fp16x3 decoder image (layouts.hpp DecSdeL6).  The two first layers are ONE 128x64 product on one
// split of the state (the heads do the same, decoder.hip head_pair_eval); `tb` = their 128 time-conditioned biases
// b + s sin t + c cos t, which are constants of an Euler step: computed once per step into LDS (sde_time_bias), not once per tile.
// Per output the products and their order are those of drift_eval_x6 / diff_eval_x6, but NOT the same bits since round 4: the
// image's layers in front of a tanh / sigmoid are packed times 2 / ln 2 (-1 / ln 2), which moves the activation's last place.
__device__ __forceinline__ void sde_time_bias(float* tb, const float* img, float sn, float cs, int i /* 0..127 */) {
  using DD = DecSdeL6;
  tb[i] = fmaf(img[DD::WCFG + i], cs, fmaf(img[DD::WSFG + i], sn, img[DD::B0FG + i]));
}
__device__ __forceinline__ void sde_fg_eval(f4 (&f)[4], float& gs, const f4 (&y)[4], const float* img, const float* tb, const Lane& L) {
  using DD = DecSdeL6;
  // (the image's four layers in front of a tanh are packed times 2 / ln 2: tanh_prescaled)
  f4 h[8];
#pragma unroll
  for (int jo = 0; jo < 8; ++jo) h[jo] = *reinterpret_cast<const f4*>(tb + 16 * jo + 4 * L.g);
  linear_acc_x6<8, 4>(h, y, img + DD::W0FG, L.lane);
  tanh_prescaled_<8>(h);
  const f4 hf[4] = {h[0], h[1], h[2], h[3]}, hg[4] = {h[4], h[5], h[6], h[7]};
  f4 h2[4];
  linear_x6<4, 4>(h2, hf, img + DD::F_W2, img + DD::F_B2, L);
  tanh_prescaled_<4>(h2);
  linear_x6<4, 4>(f, h2, img + DD::F_W4, img + DD::F_B4, L);
  linear_x6<4, 4>(h2, hg, img + DD::G_W2, img + DD::G_B2, L);
  tanh_prescaled_<4>(h2);
  gs = fast_sigmoid(row_dot(h2, img + DD::G_W4, L.g) + img[DD::G_B4]);
}
#endif

}  // namespace tsde
