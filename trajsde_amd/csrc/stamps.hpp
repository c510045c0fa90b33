// stamps.hpp -- in-kernel phase clocks of DIAGNOSTIC builds (-DTSDE_STAMPS; tools/build_variant.sh stamps "-DTSDE_STAMPS").
// A stamped kernel keeps, in the registers of ONE lane per stamped wave, the s_memtime cycles it spent between consecutive marks,
// summed per phase, and adds them to a per-kernel table in device memory when it ends; the host reads the table through
// trajsde_debug_stamps_<name>.  Nothing of this exists in the shipped library (PhaseClock<N> is then an empty struct whose calls
// compile to nothing), no output value is ever computed from a stamp, and the tables are read by no kernel.
// Reading s_memtime waits for the scalar-memory counter, which LDS instructions share: a stamped build runs a little slower than
// the shipped kernel, so the SHARES are what the tables are for (MI355X_MICROARCH.md: +11 % wave cycles per-segment stamping).
#pragma once
#include <hip/hip_runtime.h>

namespace tsde {

#ifdef TSDE_STAMPS
template <int N>
struct PhaseClock {
  unsigned long long last, real0, acc[N];
  __device__ __forceinline__ void start() {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0;
    real0 = __builtin_amdgcn_s_memrealtime();
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    acc[i] += now - last;
    last = now;
  }
  // table layout: [0, N) cycles per phase | N: stamped waves | N+1: units (caller's count) | N+2: 100 MHz ticks | N+3: spare
  __device__ __forceinline__ void flush(unsigned long long* table, unsigned long long units) {
#pragma unroll
    for (int i = 0; i < N; ++i) atomicAdd(&table[i], acc[i]);
    atomicAdd(&table[N], 1ull);
    atomicAdd(&table[N + 1], units);
    atomicAdd(&table[N + 2], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - real0));
  }
};
#define TSDE_STAMP_TABLE(name, N)                                                                                            \
  namespace tsde { __device__ unsigned long long g_stamps_##name[N + 4]; }                                                   \
  extern "C" int trajsde_debug_stamps_##name(unsigned long long* host, int reset) {                                          \
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(tsde::g_stamps_##name), (N + 4) * sizeof(unsigned long long)) != hipSuccess) return -1; \
    if (reset) {                                                                                                             \
      unsigned long long z[N + 4] = {};                                                                                      \
      if (hipMemcpyToSymbol(HIP_SYMBOL(tsde::g_stamps_##name), z, sizeof(z)) != hipSuccess) return -1;                       \
    }                                                                                                                        \
    return N;                                                                                                                \
  }
#else
template <int N>
struct PhaseClock {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush(unsigned long long*, unsigned long long) {}
};
#define TSDE_STAMP_TABLE(name, N)
#endif

}  // namespace tsde
