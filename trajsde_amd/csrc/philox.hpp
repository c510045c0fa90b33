// philox.hpp -- in-kernel Philox4x32 + Box-Muller (curand-free).  Bit-exact uint32 stream vs the host
// twin trajsde_amd/philox.py; counter = (row id, step, stream, column/4), key = 64-bit seed.
// ROUNDS = 7: Random123's Philox4x32-7, the smallest round count its authors report as passing BigCrush (10 is their default
// with a safety margin).  The stream is defined by this repo (the reference draws from torch's generator, which no kernel can
// reproduce: parity is on injected normals), so the round count is a cost choice: the two 32x32->64 multiplies of a round are
// quarter-rate instructions, a block of four normals spent 10 x 2 of them, and the noise was a quarter of the SDE step's
// vector time.  Known answers of both round counts (Random123 kat_vectors) are in tests/test_oracle_golden.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tile.hpp"

namespace tsde {

constexpr uint32_t STREAM_FAKE_AGENT = 1, STREAM_ENCODER = 2, STREAM_DECODER = 3;

constexpr int PHILOX_ROUNDS = 7;
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    // one 32x32->64 multiply per constant (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
    const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
    const uint32_t hi0 = uint32_t(p0 >> 32), lo0 = uint32_t(p0);
    const uint32_t hi1 = uint32_t(p1 >> 32), lo1 = uint32_t(p1);
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the four raw words of counter (row, step, stream, quad): what the dropout masks are cut from (dropout.hpp)
__device__ __forceinline__ void philox_words(uint64_t seed, uint32_t stream, uint32_t step, uint32_t row, uint32_t quad, uint32_t (&w)[4]) {
  philox4x32(row, step, stream, quad, uint32_t(seed), uint32_t(seed >> 32), w);
}

__device__ __forceinline__ float u01(uint32_t x) { return (float(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four standard normals for columns 4*quad .. 4*quad+3 of (stream, step, row)
__device__ __forceinline__ f4 philox_normal4(uint64_t seed, uint32_t stream, uint32_t step, uint32_t row, uint32_t quad) {
  uint32_t w[4];
  philox4x32(row, step, stream, quad, uint32_t(seed), uint32_t(seed >> 32), w);
  f4 z;
#pragma unroll
  for (int a = 0; a < 4; a += 2) {
    // v_sqrt_f32 directly (1 ulp): sqrtf() expands to the IEEE sequence, ~13 instructions, for an argument in (6e-8, 35)
    const float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u01(w[a])));
    const float u = u01(w[a + 1]);                       // angle in revolutions: v_sin/v_cos take x/(2*pi)
    z[a] = r * __builtin_amdgcn_cosf(u);
    z[a + 1] = r * __builtin_amdgcn_sinf(u);
  }
  return z;
}

struct NoiseArg {
  uint64_t seed;
  const float* z;          // injected normals or nullptr
  const int32_t* row_ids;  // global ids or nullptr
  const uint64_t* seed_dev = nullptr;   // the key lives in device memory (graph replays): read when the kernel runs
};
__device__ __forceinline__ uint64_t noise_key(const NoiseArg& na) { return na.seed_dev ? *na.seed_dev : na.seed; }

// z for the 16 features (4 quads: jt*4+g) of `row` at `step`; injected layout [steps][rows][64]
__device__ __forceinline__ void noise_row(f4 (&z)[4], const NoiseArg& na, uint32_t stream, int step, int64_t row,
                                          int64_t rows_total, int g) {
  if (na.z != nullptr) {
    const float* p = na.z + (int64_t(step) * rows_total + row) * D + 4 * g;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) z[jt] = *reinterpret_cast<const f4*>(p + 16 * jt);
  } else {
    const uint32_t rid = na.row_ids ? uint32_t(na.row_ids[row]) : uint32_t(row);
    const uint64_t key = noise_key(na);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) z[jt] = philox_normal4(key, stream, uint32_t(step), rid, uint32_t(4 * jt + g));
  }
}

}  // namespace tsde
