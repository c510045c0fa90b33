// tile_bwd.hpp -- wave-level pieces shared by the backward kernels (decoder_bwd.hip, node_bwd.hip, aggregator_bwd.hip):
// LayerNorm backward on a row tile, transposed-image products, per-wave flushing of vector-gradient accumulators.
#pragma once
#include "tile.hpp"

namespace tsde {

// x -> x_hat in place (the normalisation of tile.hpp layer_norm without the affine part); returns 1/std
__device__ __forceinline__ float ln_normalize(f4 (&a)[4]) {
  float s = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) s += (a[jt][0] + a[jt][1]) + (a[jt][2] + a[jt][3]);
  const float mean = row_sum(s) * (1.0f / 64);
  float v = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = a[jt][c] - mean;
      a[jt][c] = d;
      v += d * d;
    }
  const float rstd = rsqrt_nr(row_sum(v) * (1.0f / 64) + 1e-5f);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] *= rstd;
  return rstd;
}

// dy (grad w.r.t. gamma*x_hat+beta) -> grad w.r.t. the LayerNorm input, in place; accumulates dgamma, dbeta
__device__ __forceinline__ void ln_backward(f4 (&dy)[4], const f4 (&xh)[4], float rstd, const float* gamma, int g,
                                            f4 (&dgam)[4], f4 (&dbet)[4]) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const f4 ga = *reinterpret_cast<const f4*>(gamma + 16 * jt + 4 * g);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dgam[jt][c] = fmaf(dy[jt][c], xh[jt][c], dgam[jt][c]);
      dbet[jt][c] += dy[jt][c];
      const float gm = ga[c] * dy[jt][c];
      dy[jt][c] = gm;
      s1 += gm;
      s2 = fmaf(gm, xh[jt][c], s2);
    }
  }
  const float m1 = row_sum(s1) * (1.0f / 64), m2 = row_sum(s2) * (1.0f / 64);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) dy[jt][c] = rstd * (dy[jt][c] - m1 - xh[jt][c] * m2);
}

__device__ __forceinline__ void zero4(f4 (&a)[4]) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) a[jt] = f4{0.f, 0.f, 0.f, 0.f};
}

// out = W^T-image * in  (no bias)
__device__ __forceinline__ void linear_t(f4 (&out)[4], const f4 (&in)[4], const float* wt, const Lane& L) {
  zero4(out);
  linear_adj<4, 4>(out, in, wt, L);
}

// sum a per-lane accumulator over the 16 rows of the wave's tiles (lanes with equal g) -> 64 floats at dst
__device__ __forceinline__ void flush_vec(const f4 (&acc)[4], float* dst, const Lane& L) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    f4 v = acc[jt];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = v[c];
      x += __shfl_xor(x, 1);
      x += __shfl_xor(x, 2);
      x += __shfl_xor(x, 4);
      x += __shfl_xor(x, 8);
      v[c] = x;
    }
    if (L.n == 0) *reinterpret_cast<f4*>(dst + 16 * jt + 4 * L.g) = v;
  }
}
// the same for a value that is already equal on the 4 lanes of a row
__device__ __forceinline__ void flush_scalar(float x, float* dst, const Lane& L) {
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  if (L.lane == 0) *dst = x;
}


}  // namespace tsde
