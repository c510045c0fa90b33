// Issue rate of transcendental vector instructions (v_exp_f32, v_rcp_f32, v_rsq_f32, v_sin_f32, v_log_f32) against v_fma_f32 on gfx950,
// with 1, 2 and 4 waves per SIMD, and of a mix (1 transcendental per 4 fma) -- what a tanh / Box-Muller heavy kernel (k_sde_step) pays.
//   hipcc --offload-arch=gfx950 -O3 trans_rate.hip -o trans_rate && ./trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f + 0.5f, b = a + 0.1f, c = a + 0.2f, d = a + 0.3f, e = a + 0.4f, f = a + 0.5f, g = a + 0.6f, h = a + 0.7f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(REP8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if (MODE == 1) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if (MODE == 2) asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                     "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if (MODE == 3) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                     "v_fma_f32 %4, %4, %4, %4\n v_rcp_f32 %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if (MODE == 4) asm volatile(REP8("v_sin_f32 %0, %0\n v_log_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_cos_f32 %3, %3\n"
                                     "v_rsq_f32 %4, %4\n v_sin_f32 %5, %5\n v_log_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const char* names[5] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "2 transcendental + 6 fma", "sin/log/sqrt/cos/rsq"};
  const int iters = 2000;
  for (int waves = 1; waves <= 4; waves *= 2)
    for (int m = 0; m < 5; ++m) {
      const dim3 g(256), b(256 * waves);
      if (m == 0) k<0><<<g, b>>>(out, cyc, iters);
      if (m == 1) k<1><<<g, b>>>(out, cyc, iters);
      if (m == 2) k<2><<<g, b>>>(out, cyc, iters);
      if (m == 3) k<3><<<g, b>>>(out, cyc, iters);
      if (m == 4) k<4><<<g, b>>>(out, cyc, iters);
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      // s_memtime: shader clock cycles.  One wave's view: what 64 instructions (8 independent chains) cost it
      std::printf("%d wave(s)/SIMD  %-26s %8.3f cycles per 64 instructions of a wave\n", waves, names[m], double(c) / iters);
    }
  return 0;
}
