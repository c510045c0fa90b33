// v_mfma_f32_16x16x32_f16 with its destination ON TOP of SrcA (or SrcB): the register allocator does this for the second product of
//   D1 = A x B1;   A <- A x B2 + D1
// when registers are tight (csrc/node_bwd.hip k_edge_embed_bwd_tail at a 128-register cap is the only kernel of the library where it
// happens -- and the one whose results were not reproducible, HISTORY.md section 5 item 8).  Is the overlap safe on this hardware?
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap && ./mfma_overlap
// Each variant: the pair above with a separate destination (reference) and with the overlap, all four result registers of all lanes compared,
// 2 and 3 waves per SIMD all doing the same, 200 rounds per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define LOAD_AB                                                                                              \
  "v_mov_b32 v40, %4\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v42, %6\n\tv_mov_b32 v43, %7\n\t"                      \
  "v_mov_b32 v44, %8\n\tv_mov_b32 v45, %9\n\tv_mov_b32 v46, %10\n\tv_mov_b32 v47, %11\n\t"                    \
  "v_mov_b32 v56, %9\n\tv_mov_b32 v57, %8\n\tv_mov_b32 v58, %11\n\tv_mov_b32 v59, %10\n\t"                    \
  "s_nop 7\n\t"
#define OUT4(R0, R1, R2, R3) "s_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 "\n\tv_mov_b32 %2, " R2 "\n\tv_mov_b32 %3, " R3 "\n\t"
#define IO                                                                                                   \
  : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3])                                                          \
  : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)                            \
  : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59"

template <int V>
__global__ __launch_bounds__(768) void k(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n[4] = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r) {
    const u4 a = A[(r * 64 + lane + 64 * (threadIdx.x >> 6)) & 4095], b = B[(r * 97 + lane) & 4095];
    float w[4], o[4];
    // reference: separate destination v[48:51]
    asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[52:55], v[40:43], v[56:59], 0\n\t"
                         "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[52:55]\n\t" OUT4("v48", "v49", "v50", "v51") IO);
    for (int c = 0; c < 4; ++c) w[c] = o[c];
    if (V == 0)      // destination = SrcA
      asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[52:55], v[40:43], v[56:59], 0\n\t"
                           "v_mfma_f32_16x16x32_f16 v[40:43], v[40:43], v[44:47], v[52:55]\n\t" OUT4("v40", "v41", "v42", "v43") IO);
    if (V == 1)      // destination = SrcB
      asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[52:55], v[40:43], v[56:59], 0\n\t"
                           "v_mfma_f32_16x16x32_f16 v[44:47], v[40:43], v[44:47], v[52:55]\n\t" OUT4("v44", "v45", "v46", "v47") IO);
    if (V == 2)      // destination = SrcA, and a third product queued right behind (the matrix pipe stays busy)
      asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[52:55], v[40:43], v[56:59], 0\n\t"
                           "v_mfma_f32_16x16x32_f16 v[40:43], v[40:43], v[44:47], v[52:55]\n\t"
                           "v_mfma_f32_16x16x32_f16 v[48:51], v[56:59], v[44:47], 0\n\t" OUT4("v40", "v41", "v42", "v43") IO);
    if (V == 3)      // destination = SrcA, single product with a constant accumulator (no producer in front)
    {
      asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], 0\n\t" OUT4("v48", "v49", "v50", "v51") IO);
      for (int c = 0; c < 4; ++c) w[c] = o[c];
      asm volatile(LOAD_AB "v_mfma_f32_16x16x32_f16 v[40:43], v[40:43], v[44:47], 0\n\t" OUT4("v40", "v41", "v42", "v43") IO);
    }
    for (int c = 0; c < 4; ++c) n[c] += __float_as_uint(w[c]) != __float_as_uint(o[c]);
  }
  for (int c = 0; c < 4; ++c)
    if (n[c]) atomicAdd(&bad[(V * 4 + c) * 64 + lane], n[c]);
}

int main() {
  std::vector<u4> a(4096), b(4096);
  uint64_t s = 88172645463325252ull;
  auto half = [&]() -> unsigned { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return unsigned(0x3000 + ((s >> 20) & 0x0FFF)) | unsigned((s & 1) << 15); };
  for (auto* v : {&a, &b})
    for (auto& q : *v) q = u4{half() | (half() << 16), half() | (half() << 16), half() | (half() << 16), half() | (half() << 16)};
  u4 *dA, *dB; unsigned* dbad;
  (void)hipMalloc(&dA, 65536); (void)hipMalloc(&dB, 65536); (void)hipMalloc(&dbad, 16 * 64 * 4);
  (void)hipMemcpy(dA, a.data(), 65536, hipMemcpyHostToDevice); (void)hipMemcpy(dB, b.data(), 65536, hipMemcpyHostToDevice);
  const char* names[4] = {"destination = SrcA", "destination = SrcB", "destination = SrcA, third product behind", "destination = SrcA, constant accumulator"};
  for (int threads = 512; threads <= 768; threads += 256) {
    (void)hipMemset(dbad, 0, 16 * 64 * 4);
    const int rounds = 200;
    k<0><<<1024, threads>>>(dA, dB, dbad, rounds); k<1><<<1024, threads>>>(dA, dB, dbad, rounds);
    k<2><<<1024, threads>>>(dA, dB, dbad, rounds); k<3><<<1024, threads>>>(dA, dB, dbad, rounds);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> bad(16 * 64);
    (void)hipMemcpy(bad.data(), dbad, 16 * 64 * 4, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v) {
      std::printf("%d waves/SIMD  %-44s wrong words per result register:", threads / 256, names[v]);
      for (int c = 0; c < 4; ++c) {
        unsigned t = 0; int lo = 64, hi = -1;
        for (int l = 0; l < 64; ++l) if (bad[(v * 4 + c) * 64 + l]) { t += bad[(v * 4 + c) * 64 + l]; lo = l < lo ? l : lo; hi = l; }
        std::printf("  D%d %u", c, t);
        if (t) std::printf(" (lanes %d..%d)", lo, hi);
      }
      std::printf("   of %.0f each\n", 1024.0 * (threads / 64) * rounds * 64);
    }
  }
  return 0;
}
