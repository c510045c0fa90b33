// Does a vector instruction that overwrites a SrcA (or SrcB) register of v_mfma_f32_16x16x32_f16 right behind it change the product?
// The compiler schedules such writes with no wait states (tools/isa_mfma_war_scan.py); csrc/tile.hpp split_pair describes what was seen.
//   hipcc --offload-arch=gfx950 -O3 mfma_war.hip -o mfma_war && ./mfma_war
// Per variant: a matrix instruction on fixed registers, W wait states, then v_mov_b32 of junk into the first (or last) register of A or B;
// the accumulator is compared with the same product left alone.  2 waves per SIMD, 200 rounds per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define BODY(NOPS, VICTIM)                                                                                   \
  asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"        \
               "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\t"        \
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"            \
               "s_nop 7\n\ts_nop 7\n\t"                                                                     \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n\t" NOPS                     \
               "v_mov_b32 " VICTIM ", 0x7bff7bff\n\t"                                                       \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               "v_mov_b32 %0, v48\n\t"                                                                      \
               : "=v"(got)                                                                                   \
               : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)              \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");

template <int V>
__global__ __launch_bounds__(512) void k(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 a = A[(r * 64 + lane) & 4095], b = B[(r * 97 + lane) & 4095];
    float want, got;
    asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
                 "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\t"
                 "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n\t"
                 "s_nop 15\n\ts_nop 15\n\t"
                 "v_mov_b32 %0, v48\n\t"
                 : "=v"(want)
                 : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    if (V == 0) { BODY("", "v40") }
    if (V == 1) { BODY("s_nop 0\n\t", "v40") }
    if (V == 2) { BODY("s_nop 1\n\t", "v40") }
    if (V == 3) { BODY("s_nop 3\n\t", "v40") }
    if (V == 4) { BODY("s_nop 7\n\t", "v40") }
    if (V == 5) { BODY("", "v43") }
    if (V == 6) { BODY("", "v44") }
    if (V == 7) { BODY("", "v47") }
    if (V == 8) { BODY("s_nop 3\n\t", "v47") }
    if (V == 9) { BODY("s_nop 15\n\t", "v40") }
    n += __float_as_uint(want) != __float_as_uint(got);
  }
  if (n) atomicAdd(&bad[V * 64 + lane], n);
}

// ... and the other direction: how soon behind the matrix instruction may a vector instruction READ its result?  (the compiler leaves
// 7 wait states for this instruction.)  The accumulator starts from a known value; a stale read returns it.
#define RBODY(NOPS)                                                                                          \
  asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"        \
               "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\t"        \
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"            \
               "s_nop 7\n\ts_nop 7\n\t"                                                                     \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n\t"                            \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[44:47], v[40:43], v[48:51]\n\t" NOPS                       \
               "v_mov_b32 %0, v48\n\t"                                                                      \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               : "=v"(got)                                                                                   \
               : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)              \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
template <int V>
__global__ __launch_bounds__(512) void kr(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 a = A[(r * 64 + lane) & 4095], b = B[(r * 97 + lane) & 4095];
    float want, got;
    RBODY("s_nop 15\n\ts_nop 15\n\t")
    want = got;
    if (V == 0) { RBODY("s_nop 2\n\t") }
    if (V == 1) { RBODY("s_nop 4\n\t") }
    if (V == 2) { RBODY("s_nop 5\n\t") }
    if (V == 3) { RBODY("s_nop 6\n\t") }
    if (V == 4) { RBODY("s_nop 7\n\t") }
    if (V == 5) { RBODY("s_nop 9\n\t") }
    if (V == 6) { RBODY("s_nop 11\n\t") }
    if (V == 7) { RBODY("s_nop 15\n\t") }
    n += __float_as_uint(want) != __float_as_uint(got);
  }
  if (n) atomicAdd(&bad[V * 64 + lane], n);
}

// ... the same read by a PACKED fp32 instruction (v_pk_mul_f32 by (1, 1) of the pair D0:D1): low and high element checked separately
#define PBODY(NOPS)                                                                                          \
  asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t"        \
               "v_mov_b32 v44, %6\n\tv_mov_b32 v45, %7\n\tv_mov_b32 v46, %8\n\tv_mov_b32 v47, %9\n\t"        \
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"            \
               "v_mov_b32 v54, 1.0\n\tv_mov_b32 v55, 1.0\n\t"                                                \
               "s_nop 7\n\ts_nop 7\n\t"                                                                     \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n\t"                            \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[44:47], v[40:43], v[48:51]\n\t" NOPS                       \
               "v_pk_mul_f32 v[52:53], v[48:49], v[54:55]\n\t"                                                \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               "v_mov_b32 %0, v52\n\tv_mov_b32 %1, v53\n\t"                                                  \
               : "=v"(g0), "=v"(g1)                                                                          \
               : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)              \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
template <int V>
__global__ __launch_bounds__(512) void kp(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n0 = 0, n1 = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 a = A[(r * 64 + lane) & 4095], b = B[(r * 97 + lane) & 4095];
    float g0, g1, w0, w1;
    PBODY("s_nop 15\n\ts_nop 15\n\t")
    w0 = g0; w1 = g1;
    if (V == 0) { PBODY("s_nop 5\n\t") }
    if (V == 1) { PBODY("s_nop 6\n\t") }
    if (V == 2) { PBODY("s_nop 7\n\t") }
    if (V == 3) { PBODY("s_nop 8\n\t") }
    if (V == 4) { PBODY("s_nop 10\n\t") }
    n0 += __float_as_uint(w0) != __float_as_uint(g0);
    n1 += __float_as_uint(w1) != __float_as_uint(g1);
  }
  if (n0) atomicAdd(&bad[V * 128 + lane], n0);
  if (n1) atomicAdd(&bad[V * 128 + 64 + lane], n1);
}

// ... write-after-read on SrcC (distinct from vDst) and write-after-write on vDst: a vector write W wait states behind the instruction
#define CBODY(NOPS, VICTIM, JUNK)                                                                            \
  asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"        \
               "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\t"        \
               "v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 2.0\n\tv_mov_b32 v54, 4.0\n\tv_mov_b32 v55, 0.5\n\t"    \
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"            \
               "s_nop 7\n\ts_nop 7\n\t"                                                                     \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[52:55]\n\t" NOPS                       \
               "v_mov_b32 " VICTIM ", " JUNK "\n\t"                                                          \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               "v_mov_b32 %0, v48\n\t"                                                                      \
               : "=v"(got)                                                                                   \
               : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)              \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56");
template <int V>
__global__ __launch_bounds__(512) void kc(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 a = A[(r * 64 + lane) & 4095], b = B[(r * 97 + lane) & 4095];
    float want, got;
    CBODY("s_nop 15\n\ts_nop 15\n\t", "v56", "0")                    // (a bystander register: the clean product)
    want = got;
    if (V == 0) { CBODY("", "v52", "0x42c80000") }                    // SrcC register 0, junk 100.0
    if (V == 1) { CBODY("s_nop 1\n\t", "v52", "0x42c80000") }
    if (V == 2) { CBODY("s_nop 3\n\t", "v52", "0x42c80000") }
    if (V == 3) { CBODY("s_nop 7\n\t", "v52", "0x42c80000") }
    if (V == 4) { CBODY("s_nop 11\n\t", "v52", "0x42c80000") }
    if (V == 5) { CBODY("", "v48", "0x42c80000") }                    // vDst register 0: the product must still win
    if (V == 6) { CBODY("s_nop 3\n\t", "v48", "0x42c80000") }
    if (V == 7) { CBODY("s_nop 7\n\t", "v48", "0x42c80000") }
    if (V == 8) { CBODY("s_nop 11\n\t", "v48", "0x42c80000") }
    n += __float_as_uint(want) != __float_as_uint(got);
  }
  if (n) atomicAdd(&bad[V * 64 + lane], n);
}

// ... read-after-write INTO the instruction: SrcC registers 0:1 written by a packed (or a plain) multiply W wait states in front of it.
// A = 0, so D = C: a stale read returns the old C (3.0) instead of the product (lane-dependent).
#define SBODY(PROD, NOPS)                                                                                    \
  asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"            \
               "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %3\n\tv_mov_b32 v46, %4\n\tv_mov_b32 v47, %5\n\t"        \
               "v_mov_b32 v52, 0x40400000\n\tv_mov_b32 v53, 0x40400000\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t" \
               "v_mov_b32 v56, %6\n\tv_mov_b32 v57, %7\n\tv_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 2.0\n\t"      \
               "s_nop 7\n\ts_nop 7\n\t" PROD NOPS                                                           \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[52:55]\n\t"                            \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               "v_mov_b32 %0, v48\n\tv_mov_b32 %1, v49\n\t"                                                  \
               : "=v"(g0), "=v"(g1)                                                                          \
               : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(x0), "v"(x1)                                    \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
#define PK "v_pk_mul_f32 v[52:53], v[56:57], v[58:59]\n\t"
#define PL "v_mul_f32 v52, v56, v58\n\tv_mul_f32 v53, v57, v59\n\t"
template <int V>
__global__ __launch_bounds__(512) void ks(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n0 = 0, n1 = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 b = B[(r * 97 + lane) & 4095];
    const float x0 = float(lane + r % 7) + 0.25f, x1 = float(2 * lane + r % 5) + 0.5f;
    float g0, g1;
    if (V == 0) { SBODY(PK, "") }
    if (V == 1) { SBODY(PK, "s_nop 0\n\t") }
    if (V == 2) { SBODY(PK, "s_nop 1\n\t") }
    if (V == 3) { SBODY(PK, "s_nop 2\n\t") }
    if (V == 4) { SBODY(PK, "s_nop 4\n\t") }
    if (V == 5) { SBODY(PL, "") }
    if (V == 6) { SBODY(PL, "s_nop 0\n\t") }
    if (V == 7) { SBODY(PL, "s_nop 1\n\t") }
    if (V == 8) { SBODY(PL, "s_nop 2\n\t") }
    n0 += g0 != 2.0f * x0;
    n1 += g1 != 2.0f * x1;
  }
  if (n0) atomicAdd(&bad[V * 64 + lane], n0);
  if (n1) atomicAdd(&bad[V * 64 + lane], n1 << 16);
}

// ... packed fp32 instructions among themselves and next to plain ones, no wait states, behind a matrix instruction in flight:
// producer -> consumer pairs (plain -> packed, packed -> plain, packed -> packed), every lane checked against the exact product
#define VBODY(SEQ)                                                                                           \
  asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t"        \
               "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %3\n\tv_mov_b32 v46, %4\n\tv_mov_b32 v47, %5\n\t"        \
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"            \
               "v_mov_b32 v56, %6\n\tv_mov_b32 v57, %7\n\tv_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 4.0\n\t"      \
               "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"            \
               "s_nop 7\n\t"                                                                                \
               "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n\t" SEQ                       \
               "s_nop 15\n\ts_nop 15\n\t"                                                                   \
               "v_mov_b32 %0, v62\n\tv_mov_b32 %1, v63\n\t"                                                  \
               : "=v"(g0), "=v"(g1)                                                                          \
               : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(x0), "v"(x1)                                    \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
template <int V>
__global__ __launch_bounds__(512) void kv(const u4* A, const u4* B, unsigned* bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned n0 = 0, n1 = 0;
  for (int r = 0; r < rounds; ++r) {
    const u4 b = B[(r * 97 + lane) & 4095];
    const float x0 = float(lane + r % 7) + 0.25f, x1 = float(2 * lane + r % 5) + 0.5f;
    float g0, g1, w0, w1;
    if (V == 0) {   // plain -> packed: (x0 * 2, x1 * 4) then packed (.., ..) * (2, 4)
      VBODY("v_mul_f32 v60, v56, v58\n\tv_mul_f32 v61, v57, v59\n\tv_pk_mul_f32 v[62:63], v[60:61], v[58:59]\n\t")
      w0 = x0 * 4.f; w1 = x1 * 16.f;
    }
    if (V == 1) {   // packed -> plain
      VBODY("v_pk_mul_f32 v[60:61], v[56:57], v[58:59]\n\tv_mul_f32 v62, v60, v58\n\tv_mul_f32 v63, v61, v59\n\t")
      w0 = x0 * 4.f; w1 = x1 * 16.f;
    }
    if (V == 2) {   // packed -> packed
      VBODY("v_pk_mul_f32 v[60:61], v[56:57], v[58:59]\n\tv_pk_mul_f32 v[62:63], v[60:61], v[58:59]\n\t")
      w0 = x0 * 4.f; w1 = x1 * 16.f;
    }
    if (V == 3) {   // packed -> packed fma with op_sel_hi broadcast (what the adjoint scaling compiles to)
      VBODY("v_pk_mul_f32 v[60:61], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[62:63], v[60:61], v[58:59], 0 op_sel_hi:[1,0,0]\n\t")
      w0 = x0 * 4.f; w1 = x1 * 8.f;
    }
    if (V == 4) {   // packed mov shuffle of a packed result: v62 = v61, v63 = v60
      VBODY("v_pk_mul_f32 v[60:61], v[56:57], v[58:59]\n\tv_pk_mov_b32 v[62:63], v[60:61], v[60:61] op_sel:[1,0]\n\t")
      w0 = x1 * 4.f; w1 = x0 * 2.f;
    }
    n0 += g0 != w0;
    n1 += g1 != w1;
  }
  if (n0) atomicAdd(&bad[V * 64 + lane], n0);
  if (n1) atomicAdd(&bad[V * 64 + lane], n1 << 16);
}

int main() {
  std::vector<u4> a(4096), b(4096);
  uint64_t s = 88172645463325252ull;
  auto half = [&]() -> unsigned { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return unsigned(0x3000 + ((s >> 20) & 0x0FFF)) | unsigned((s & 1) << 15); };   // |x| in [0.125, 1)
  for (auto* v : {&a, &b})
    for (auto& q : *v) q = u4{half() | (half() << 16), half() | (half() << 16), half() | (half() << 16), half() | (half() << 16)};
  u4 *dA, *dB; unsigned* dbad;
  hipMalloc(&dA, 65536); hipMalloc(&dB, 65536); hipMalloc(&dbad, 10 * 64 * 4);
  hipMemcpy(dA, a.data(), 65536, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), 65536, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 10 * 64 * 4);
  const int rounds = 200;
#define RUN(V) k<V><<<1024, 512>>>(dA, dB, dbad, rounds);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  hipDeviceSynchronize();
  std::vector<unsigned> bad(640);
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const char* names[10] = {"A reg 0, +0 wait", "A reg 0, +1 wait", "A reg 0, +2 wait", "A reg 0, +4 wait", "A reg 0, +8 wait", "A reg 3, +0 wait",
                           "B reg 0, +0 wait", "B reg 3, +0 wait", "B reg 3, +4 wait", "A reg 0, +16 wait"};
  const double per_lane = 1024.0 * 8 * rounds;
  for (int v = 0; v < 10; ++v) {
    unsigned tot = 0; int lanes = 0, first = -1, last = -1;
    for (int l = 0; l < 64; ++l) if (bad[v * 64 + l]) { tot += bad[v * 64 + l]; ++lanes; if (first < 0) first = l; last = l; }
    std::printf("%-18s: %10u wrong accumulator words (D register 0) of %.0f, in %d lanes (%d..%d) -> rate %.2e per lane-product\n", names[v], tot,
                per_lane * 64, lanes, first, last, tot / (per_lane * 64));
  }
  hipMemset(dbad, 0, 10 * 64 * 4);
#define RUNR(V) kr<V><<<1024, 512>>>(dA, dB, dbad, rounds);
  RUNR(0) RUNR(1) RUNR(2) RUNR(3) RUNR(4) RUNR(5) RUNR(6) RUNR(7)
  hipDeviceSynchronize();
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const int waits[8] = {3, 5, 6, 7, 8, 10, 12, 16};
  for (int v = 0; v < 8; ++v) {
    unsigned tot = 0; int lanes = 0, first = -1, last = -1;
    for (int l = 0; l < 64; ++l) if (bad[v * 64 + l]) { tot += bad[v * 64 + l]; ++lanes; if (first < 0) first = l; last = l; }
    std::printf("read of D register 0, %2d wait states behind the 2nd of two chained instructions: %10u stale of %.0f, in %d lanes (%d..%d)\n", waits[v], tot,
                per_lane * 64, lanes, first, last);
  }
  hipMemset(dbad, 0, 10 * 64 * 4);
#define RUNP(V) kp<V><<<1024, 512>>>(dA, dB, dbad, rounds);
  RUNP(0) RUNP(1) RUNP(2) RUNP(3) RUNP(4)
  hipDeviceSynchronize();
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const int pw[5] = {6, 7, 8, 9, 11};
  for (int v = 0; v < 5; ++v)
    for (int e = 0; e < 2; ++e) {
      unsigned tot = 0; int lanes = 0, first = -1, last = -1;
      for (int l = 0; l < 64; ++l) if (bad[v * 128 + e * 64 + l]) { tot += bad[v * 128 + e * 64 + l]; ++lanes; if (first < 0) first = l; last = l; }
      std::printf("packed read of D0:D1, %2d wait states, element %d: %10u stale of %.0f, in %d lanes (%d..%d)\n", pw[v], e, tot, per_lane * 64, lanes, first, last);
    }
  hipMemset(dbad, 0, 10 * 64 * 4);
#define RUNC(V) kc<V><<<1024, 512>>>(dA, dB, dbad, rounds);
  RUNC(0) RUNC(1) RUNC(2) RUNC(3) RUNC(4) RUNC(5) RUNC(6) RUNC(7) RUNC(8)
  hipDeviceSynchronize();
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const char* cn[9] = {"SrcC reg 0, +0 wait", "SrcC reg 0, +2 wait", "SrcC reg 0, +4 wait", "SrcC reg 0, +8 wait", "SrcC reg 0, +12 wait",
                       "vDst reg 0, +0 wait", "vDst reg 0, +4 wait", "vDst reg 0, +8 wait", "vDst reg 0, +12 wait"};
  for (int v = 0; v < 9; ++v) {
    unsigned tot = 0; int lanes = 0, first = -1, last = -1;
    for (int l = 0; l < 64; ++l) if (bad[v * 64 + l]) { tot += bad[v * 64 + l]; ++lanes; if (first < 0) first = l; last = l; }
    std::printf("vector write to %-22s: %10u wrong D0 words of %.0f, in %d lanes (%d..%d)\n", cn[v], tot, per_lane * 64, lanes, first, last);
  }
  hipMemset(dbad, 0, 10 * 64 * 4);
#define RUNS(V) ks<V><<<1024, 512>>>(dA, dB, dbad, 40);
  RUNS(0) RUNS(1) RUNS(2) RUNS(3) RUNS(4) RUNS(5) RUNS(6) RUNS(7) RUNS(8)
  hipDeviceSynchronize();
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const char* sn[9] = {"packed multiply, +0 wait", "packed multiply, +1 wait", "packed multiply, +2 wait", "packed multiply, +3 wait", "packed multiply, +5 wait",
                       "two plain multiplies, +0 wait (1 for the first)", "two plain multiplies, +1 wait", "two plain multiplies, +2 wait", "two plain multiplies, +3 wait"};
  for (int v = 0; v < 9; ++v) {
    unsigned t0 = 0, t1 = 0; int first = -1, last = -1;
    for (int l = 0; l < 64; ++l) if (bad[v * 64 + l]) { t0 += bad[v * 64 + l] & 0xFFFF; t1 += bad[v * 64 + l] >> 16; if (first < 0) first = l; last = l; }
    std::printf("SrcC 0:1 written by %-48s: stale C0 %8u, stale C1 %8u of %d each, lanes %d..%d\n", sn[v], t0, t1, 1024 * 8 * 40 * 64, first, last);
  }
  hipMemset(dbad, 0, 10 * 64 * 4);
#define RUNV(V) kv<V><<<1024, 512>>>(dA, dB, dbad, 40);
  RUNV(0) RUNV(1) RUNV(2) RUNV(3) RUNV(4)
  hipDeviceSynchronize();
  hipMemcpy(bad.data(), dbad, 2560, hipMemcpyDeviceToHost);
  const char* vn[5] = {"plain -> packed", "packed -> plain", "packed -> packed", "packed -> packed fma (op_sel_hi)", "packed -> v_pk_mov_b32 (op_sel)"};
  for (int v = 0; v < 5; ++v) {
    unsigned t0 = 0, t1 = 0;
    for (int l = 0; l < 64; ++l) { t0 += bad[v * 64 + l] & 0xFFFF; t1 += bad[v * 64 + l] >> 16; }
    std::printf("back to back, behind a matrix instruction: %-36s wrong element 0 %8u, element 1 %8u of %d each\n", vn[v], t0, t1, 1024 * 8 * 40 * 64);
  }
  return 0;
}
