// coexec.hip -- can the vector instructions of one wave execute beside the matrix instructions of ANOTHER wave of the same SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coexec tools/microbench/coexec.hip && /tmp/coexec
//
// 512-thread workgroups, one per CU: waves w and w + 4 share a SIMD (MI355X_MICROARCH.md, "Two waves per SIMD").  Each role is a
// loop of inline-assembly blocks: M = 48 independent-accumulator v_mfma_f32_16x16x32_f16 (8 accumulators, the dependent
// distance of the edge kernel), V = 192 v_fma_f32 on 16 independent registers.  Modes:
//   0  every wave runs M only                       (matrix pipe shared by the two waves of a SIMD)
//   1  every wave runs V only
//   2  waves 0-3 run M only, waves 4-7 run V only   (role split: does V hide behind M?)
//   3  every wave alternates M, V; waves 4-7 start with V (opposite phases)
//   4  every wave alternates M, V; all start with M (same phases)
//   5  waves 0-3 only (4-7 exit): M then V alternating (one wave per SIMD: the serial reference)
//   6  as 2 with the matrix accumulators in AGPRs
//   7  as 4 with the M phase fed from LDS (2 ds_read_b128 per 6 matrix instructions, one step ahead: the edge kernel's form)
//   8 / 10  as 4 with the V phase as 2 / 4 dependent chains (a LayerNorm reduction, a softmax update);  9 = 7 + 8
//   11 / 12  V only, 2 / 4 dependent chains;  13  M only, fed from LDS;  14 / 15  one wave per SIMD, M only
//   16 / 17  every matrix instruction followed by 4 independent vector instructions, one / two waves per SIMD;  18 / 19  by 2
// Printed: microseconds per launch and SIMD cycles per loop iteration at the clock implied by mode 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define M6(a, b0, b1) \
  "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\t"  "v_mfma_f32_16x16x32_f16 %1, %8, %10, %1\n\t" \
  "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"  "v_mfma_f32_16x16x32_f16 %3, %8, %10, %3\n\t" \
  "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\t"  "v_mfma_f32_16x16x32_f16 %5, %8, %10, %5\n\t" \
  "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\t"  "v_mfma_f32_16x16x32_f16 %7, %8, %10, %7\n\t"

__device__ __forceinline__ void phase_m(f4 (&acc)[8], const h8& a, const h8& b0, const h8& b1) {
  // 48 matrix instructions: 6 x (8 accumulators)
  asm volatile(M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
               : "v"(a), "v"(b0), "v"(b1));
}
__device__ __forceinline__ void phase_m_agpr(f4 (&acc)[8], const h8& a, const h8& b0, const h8& b1) {
  asm volatile(M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1) M6(a, b0, b1)
               : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7])
               : "v"(a), "v"(b0), "v"(b1));
}
#define V16 \
  "v_fma_f32 %0, %0, %16, %17\n\t" "v_fma_f32 %1, %1, %16, %17\n\t" "v_fma_f32 %2, %2, %16, %17\n\t" "v_fma_f32 %3, %3, %16, %17\n\t" \
  "v_fma_f32 %4, %4, %16, %17\n\t" "v_fma_f32 %5, %5, %16, %17\n\t" "v_fma_f32 %6, %6, %16, %17\n\t" "v_fma_f32 %7, %7, %16, %17\n\t" \
  "v_fma_f32 %8, %8, %16, %17\n\t" "v_fma_f32 %9, %9, %16, %17\n\t" "v_fma_f32 %10, %10, %16, %17\n\t" "v_fma_f32 %11, %11, %16, %17\n\t" \
  "v_fma_f32 %12, %12, %16, %17\n\t" "v_fma_f32 %13, %13, %16, %17\n\t" "v_fma_f32 %14, %14, %16, %17\n\t" "v_fma_f32 %15, %15, %16, %17\n\t"
__device__ __forceinline__ void phase_v(float (&x)[16], float m, float c) {
  // 192 vector instructions: 12 x 16 independent registers
  asm volatile(V16 V16 V16 V16 V16 V16 V16 V16 V16 V16 V16 V16
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                 "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
               : "v"(m), "v"(c));
}

// M phase fed from LDS like the edge kernel: 8 steps of {6 matrix instructions on 2 accumulators, the two ds_read_b128 of the NEXT
// step's weight fragments}; the fragments of a step are waited for (lgkmcnt) right before its first matrix instruction
__device__ __forceinline__ void phase_m_lds(f4 (&acc)[8], const h8& b0, const h8& b1, const float* lds, int lane) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 f1[2], f2[2];
  f1[0] = *reinterpret_cast<const u4*>(lds + lane * 4);
  f2[0] = *reinterpret_cast<const u4*>(lds + lane * 4 + 256);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i + 1 < 8) {
      f1[(i + 1) & 1] = *reinterpret_cast<const u4*>(lds + (i + 1) * 512 + lane * 4);
      f2[(i + 1) & 1] = *reinterpret_cast<const u4*>(lds + (i + 1) * 512 + lane * 4 + 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    const h8 a1 = __builtin_bit_cast(h8, f1[i & 1]), a2 = __builtin_bit_cast(h8, f2[i & 1]);
    f4& x = acc[(2 * i) & 7];
    f4& y = acc[(2 * i + 1) & 7];
    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, x, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, y, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, x, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, y, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b0, x, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, y, 0, 0, 0);
  }
}
// V phase as two dependent chains (a LayerNorm's sum of squares, a softmax update): 192 instructions, 2 x 96 deep
__device__ __forceinline__ void phase_v_dep(float (&x)[16], float m, float c) {
#define D2 "v_fma_f32 %0, %0, %2, %3\n\t" "v_fma_f32 %1, %1, %2, %3\n\t"
#define D16 D2 D2 D2 D2 D2 D2 D2 D2
  asm volatile(D16 D16 D16 D16 D16 D16 D16 D16 D16 D16 D16 D16 : "+v"(x[0]), "+v"(x[1]) : "v"(m), "v"(c));
}
// ... four chains
__device__ __forceinline__ void phase_v_dep4(float (&x)[16], float m, float c) {
#define E4 "v_fma_f32 %0, %0, %4, %5\n\t" "v_fma_f32 %1, %1, %4, %5\n\t" "v_fma_f32 %2, %2, %4, %5\n\t" "v_fma_f32 %3, %3, %4, %5\n\t"
#define E16 E4 E4 E4 E4
  asm volatile(E16 E16 E16 E16 E16 E16 E16 E16 E16 E16 E16 E16 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(m), "v"(c));
}

// one stream with the two kinds interleaved: every matrix instruction followed by NV independent vector instructions
// (48 matrix + 48 NV vector instructions per call): what in-wave software pipelining of two row tiles would issue
template <int NV>
__device__ __forceinline__ void phase_mix(f4 (&acc)[8], float (&x)[16], const h8& a, const h8& b0, float m, float c) {
#define MX(A_, X0, X1, X2, X3) "v_mfma_f32_16x16x32_f16 %" #A_ ", %24, %25, %" #A_ "\n\t" \
  "v_fma_f32 %" #X0 ", %" #X0 ", %26, %27\n\t" "v_fma_f32 %" #X1 ", %" #X1 ", %26, %27\n\t" \
  "v_fma_f32 %" #X2 ", %" #X2 ", %26, %27\n\t" "v_fma_f32 %" #X3 ", %" #X3 ", %26, %27\n\t"
#define MX2(A_, X0, X1) "v_mfma_f32_16x16x32_f16 %" #A_ ", %24, %25, %" #A_ "\n\t" \
  "v_fma_f32 %" #X0 ", %" #X0 ", %26, %27\n\t" "v_fma_f32 %" #X1 ", %" #X1 ", %26, %27\n\t"
#define MX8 MX(0, 8, 9, 10, 11) MX(1, 12, 13, 14, 15) MX(2, 16, 17, 18, 19) MX(3, 20, 21, 22, 23) MX(4, 8, 9, 10, 11) MX(5, 12, 13, 14, 15) MX(6, 16, 17, 18, 19) MX(7, 20, 21, 22, 23)
#define MX28 MX2(0, 8, 9) MX2(1, 10, 11) MX2(2, 12, 13) MX2(3, 14, 15) MX2(4, 16, 17) MX2(5, 18, 19) MX2(6, 20, 21) MX2(7, 22, 23)
  if (NV == 4)
    asm volatile(MX8 MX8 MX8 MX8 MX8 MX8
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                   "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                   "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                 : "v"(a), "v"(b0), "v"(m), "v"(c));
  else
    asm volatile(MX28 MX28 MX28 MX28 MX28 MX28
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                   "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                   "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                 : "v"(a), "v"(b0), "v"(m), "v"(c));
}

// the pipelined edge kernel's step, compiled (not assembly): the next step's two fragments read from LDS, three matrix instructions
// on one accumulator pair, NV independent vector instructions, a scheduling barrier -- 16 steps = 48 matrix instructions
template <int NV>
__device__ __forceinline__ void phase_pipe(f4 (&acc)[8], float (&x)[16], const h8& b0, const h8& b1, const float* lds, int lane, float m, float c) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 f1[2], f2[2];
  f1[0] = *reinterpret_cast<const u4*>(lds + lane * 4);
  f2[0] = *reinterpret_cast<const u4*>(lds + lane * 4 + 256);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i + 1 < 16) {
      f1[(i + 1) & 1] = *reinterpret_cast<const u4*>(lds + ((i + 1) & 7) * 512 + lane * 4);
      f2[(i + 1) & 1] = *reinterpret_cast<const u4*>(lds + ((i + 1) & 7) * 512 + lane * 4 + 256);
    }
    const h8 a1 = __builtin_bit_cast(h8, f1[i & 1]), a2 = __builtin_bit_cast(h8, f2[i & 1]);
    f4& y = acc[i & 7];
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b0, y, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < NV; ++v) x[(v + 4 * i) & 15] = __builtin_fmaf(x[(v + 4 * i) & 15], m, c);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 512];
  for (int i = threadIdx.x; i < 8 * 512; i += 512) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  f4 acc[8];
  float x[16];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) x[i] = float(threadIdx.x + i);
  h8 a, b0, b1;
  for (int i = 0; i < 8; ++i) { a[i] = _Float16(0.001f * (threadIdx.x & 7)); b0[i] = _Float16(0.5f); b1[i] = _Float16(0.25f); }
  const float m = 0.999f, c = 0.001f;
  const bool young = wave >= 4;
  const bool idle = (MODE == 5 || MODE == 14 || MODE == 15 || MODE == 16 || MODE == 18 || MODE == 20 || MODE == 22) && young;    // one wave per SIMD: waves 4-7 do nothing
  for (int it = 0; it < (idle ? 0 : iters); ++it) {
    if (MODE == 0) phase_m(acc, a, b0, b1);
    else if (MODE == 1) phase_v(x, m, c);
    else if (MODE == 2) { if (young) phase_v(x, m, c); else phase_m(acc, a, b0, b1); }
    else if (MODE == 6) { if (young) phase_v(x, m, c); else phase_m_agpr(acc, a, b0, b1); }
    else if (MODE == 3) { if (young) { phase_v(x, m, c); phase_m(acc, a, b0, b1); } else { phase_m(acc, a, b0, b1); phase_v(x, m, c); } }
    else if (MODE == 4) { phase_m(acc, a, b0, b1); phase_v(x, m, c); }
    else if (MODE == 7) { phase_m_lds(acc, b0, b1, lds, threadIdx.x & 63); phase_v(x, m, c); }
    else if (MODE == 8) { phase_m(acc, a, b0, b1); phase_v_dep(x, m, c); }
    else if (MODE == 9) { phase_m_lds(acc, b0, b1, lds, threadIdx.x & 63); phase_v_dep(x, m, c); }
    else if (MODE == 10) { phase_m(acc, a, b0, b1); phase_v_dep4(x, m, c); }
    else if (MODE == 11) { phase_v_dep(x, m, c); }
    else if (MODE == 12) { phase_v_dep4(x, m, c); }
    else if (MODE == 13) { phase_m_lds(acc, b0, b1, lds, threadIdx.x & 63); }
    else if (MODE == 5) { phase_m(acc, a, b0, b1); phase_v(x, m, c); }
    else if (MODE == 14) { phase_m_lds(acc, b0, b1, lds, threadIdx.x & 63); }
    else if (MODE == 15) { phase_m(acc, a, b0, b1); }
    else if (MODE == 16 || MODE == 17) { phase_mix<4>(acc, x, a, b0, m, c); }
    else if (MODE == 18 || MODE == 19) { phase_mix<2>(acc, x, a, b0, m, c); }
    else if (MODE == 20 || MODE == 21) { phase_pipe<12>(acc, x, b0, b1, lds, threadIdx.x & 63, m, c); }
    else if (MODE == 22 || MODE == 23) { phase_pipe<6>(acc, x, b0, b1, lds, threadIdx.x & 63, m, c); }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<256, 512>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / 5;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 2000;
  const char* what[] = {"all M (48 mfma / iter / wave)", "all V (192 fma / iter / wave)", "waves 0-3 M, waves 4-7 V", "alternate, opposite phases",
                        "alternate, same phases", "one wave per SIMD, alternate", "waves 0-3 M (AGPR acc), waves 4-7 V",
                        "alternate, M fed from LDS", "alternate, V = 2 dependent chains", "alternate, LDS-fed M + 2-chain V",
                        "alternate, V = 4 dependent chains", "all V, 2 dependent chains", "all V, 4 dependent chains", "all M fed from LDS",
                        "ONE wave per SIMD: M fed from LDS (edge-kernel chain)", "ONE wave per SIMD: M, 8 accumulators",
                        "ONE wave: interleaved 48 x (mfma + 4 fma)", "two waves: interleaved 48 x (mfma + 4 fma)",
                        "ONE wave: interleaved 48 x (mfma + 2 fma)", "two waves: interleaved 48 x (mfma + 2 fma)",
                        "ONE wave: 16 compiled steps (2 ds_read, 3 mfma, 12 fma)", "two waves: 16 compiled steps (.., 12 fma)",
                        "ONE wave: 16 compiled steps (2 ds_read, 3 mfma, 6 fma)", "two waves: 16 compiled steps (.., 6 fma)"};
  float us[24];
  us[0] = run<0>(out, iters); us[1] = run<1>(out, iters); us[2] = run<2>(out, iters); us[3] = run<3>(out, iters);
  us[4] = run<4>(out, iters); us[5] = run<5>(out, iters); us[6] = run<6>(out, iters); us[7] = run<7>(out, iters);
  us[8] = run<8>(out, iters); us[9] = run<9>(out, iters); us[10] = run<10>(out, iters); us[11] = run<11>(out, iters);
  us[12] = run<12>(out, iters); us[13] = run<13>(out, iters); us[14] = run<14>(out, iters); us[15] = run<15>(out, iters); us[16] = run<16>(out, iters); us[17] = run<17>(out, iters);
  us[18] = run<18>(out, iters); us[19] = run<19>(out, iters); us[20] = run<20>(out, iters); us[21] = run<21>(out, iters);
  us[22] = run<22>(out, iters); us[23] = run<23>(out, iters);
  // mode 0: two waves x 48 mfma x 16 cycles per SIMD and iteration
  const double ghz = 2.0 * 48 * 16 * iters / (us[0] * 1e3);
  printf("clock implied by mode 0 (matrix pipe saturated): %.2f GHz\n", ghz);
  for (int mde = 0; mde < 24; ++mde)
    printf("mode %d  %-40s %9.1f us   %7.0f SIMD cycles / iteration\n", mde, what[mde], us[mde], us[mde] * 1e3 * ghz / iters);
  return 0;
}
