// Is v_dot2c_f32_f16 with the constant (-1, 0) / (0, -1) the exact residual x - fp16_rtz(x), also where the fp16 piece is subnormal?
// (csrc/tile.hpp split_pair forms the lo piece of the fp16x3 operand split this way.)   hipcc --offload-arch=gfx950 -O3 dot2.hip -o dot2 && ./dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* ref, float* dot, int n) {
  const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 1 >= n) return;
  const float x0 = x[i], x1 = x[i + 1];
  const hp2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const hf2 hh = __builtin_bit_cast(hf2, h);
  ref[i] = x0 - float(hh[0]);
  ref[i + 1] = x1 - float(hh[1]);
  const hf2 e0 = __builtin_bit_cast(hf2, 0x0000BC00u), e1 = __builtin_bit_cast(hf2, 0xBC000000u);
  dot[i] = __builtin_amdgcn_fdot2(hh, e0, x0, false);
  dot[i + 1] = __builtin_amdgcn_fdot2(hh, e1, x1, false);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const int e = int((s >> 40) % 56) - 40;                     // 2^-40 .. 2^15: below, inside and above the fp16 subnormal range
    const float m = 1.0f + float((s >> 8) & 0x7FFFFF) / 8388608.0f;
    x[i] = ((s & 1) ? -1.f : 1.f) * std::ldexp(m, e);
  }
  x[0] = 0.f; x[1] = -0.f; x[2] = 65504.f; x[3] = 6.1e-5f; x[4] = 5.96e-8f; x[5] = 2.9e-8f; x[6] = 1e-30f; x[7] = -3e-6f;
  float *dx, *dr, *dd;
  hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dd, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, dr, dd, n);
  std::vector<float> r(n), d(n);
  hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
  int bad = 0, sub = 0, bad_normal = 0;
  double worst = 0.0;                                   // largest |dot2 - (x - h)| relative to |x|
  for (int i = 0; i < n; ++i) {
    const bool small = std::fabs(x[i]) < 6.1035e-5f;    // fp16 piece subnormal (or zero)
    if (small && x[i] != 0.f) ++sub;
    if (std::memcmp(&r[i], &d[i], 4) != 0 && !(r[i] == 0.f && d[i] == 0.f)) {
      if (bad < 6 || (!small && bad_normal < 6)) std::printf("x %.9g: x - h = %.9g, dot2 = %.9g\n", x[i], r[i], d[i]);
      ++bad;
      bad_normal += !small;
      const double e = std::fabs(double(r[i]) - double(d[i])) / std::fabs(double(x[i]));
      worst = e > worst ? e : worst;
    }
  }
  std::printf("%d values (%d with a subnormal fp16 piece): %d differ, %d of them with a normal fp16 piece; worst error %.3g of |x| (2^-22 = 2.4e-7)\n",
              n, sub, bad, bad_normal, worst);
  return bad_normal != 0;
}
