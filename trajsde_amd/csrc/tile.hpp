// tile.hpp -- the wave-level building blocks shared by every kernel of the path (gfx950 / CDNA4 only).
//
// Design: "row on lane".  A wave owns a tile of 16 rows (agent-samples, edges or nodes).  Lane l holds
// row n = l & 15 and, of that row's features, the quads selected by g = l >> 4: a 64-feature activation
// is   f4 a[4]   with   a[jt][c] = X[row n][16*jt + 4*g + c].
// A linear layer is computed TRANSPOSED on the matrix cores, Y^T = W * X^T.  Described here for the exact-fp32
// v_mfma_f32_16x16x4_f32 (A = 16 output features x 4 k, B = 4 k x 16 rows, D = 16 features x 16 rows); the
// split-precision products further down (fp16x3 by default, K = 32 per instruction) keep the same D layout:
//   - the D fragment (col = lane&15 = row, row = 4*(lane>>4)+reg = feature) is *already* the
//     activation layout above, so chains of layers never leave registers: no LDS transpose,
//     no cross-lane traffic (cf. cdna_hip_programming.md section 3, accumulator tile as next operand);
//   - the k index consumed by instruction (q, c) on lane group g is k = 16q + 4g + c, i.e. exactly
//     a[q][c]; the matching weight element for lane (i, g) is W[16*jo + i][16q + 4g + c], four
//     contiguous floats of the row-major nn.Linear weight -> one ds_read_b128 per 4 MFMAs.
// Weights live in LDS in "fragment order" (pack.hip): [jo][q][lane][4], conflict-free b128 reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "layouts.hpp"   // TSDE_SPLIT_H3

// round-6 forms of the edge attention's stream (A/B switches of tools/build_variant.sh; all on by default):
//   TSDE_R6_SOFTMAX  logits in log2 units (2^x without the pre-multiply), one v_max per head group, sums updated in place
//   TSDE_R6_IN2      first layers: operand packed by v_fma_mixhi, all 16 matrix instructions of a tile pair back to back
//   TSDE_R6_ATOMS    vector work issued between the matrix instructions of the W_A and lin_v products
#ifndef TSDE_R6_SOFTMAX
#define TSDE_R6_SOFTMAX 1
#endif
#ifndef TSDE_R6_IN2
#define TSDE_R6_IN2 1
#endif
#ifndef TSDE_R6_ATOMS
#define TSDE_R6_ATOMS 1
#endif
//   TSDE_R6_TOP      edge walk of the fused attention in 32-bit arithmetic through buffer descriptors, scalar loop counter
#ifndef TSDE_R6_TOP
#define TSDE_R6_TOP 1
#endif
//   TSDE_R6_SPLIT    operand split: low pieces by v_fma_mix_f32 + v_cvt_pkrtz (half-rate) instead of v_fma_mixlo/hi_f16 (quarter-rate)
#ifndef TSDE_R6_SPLIT
#define TSDE_R6_SPLIT 1
#endif

// Correctness guard (HISTORY.md section 5 item 8, trajsde_amd/build.py): built with the SLP vectoriser on, identical launches of the
// backward tile kernels disagree in their low-order bits.  The build passes -fno-slp-vectorize together with -DTSDE_NO_SLP=1; a
// recipe that forgets the pair stops here instead of shipping a library whose results are not reproducible.
#if !defined(TSDE_NO_SLP) || !TSDE_NO_SLP
#error "build trajsde csrc with -fno-slp-vectorize -DTSDE_NO_SLP=1 (python -m trajsde_amd.build): see csrc/tile.hpp split_pair"
#endif

namespace tsde {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int D = 64;          // embed width of every stage (CFG:32)
constexpr int TILE_ROWS = 16;  // rows per wave tile

struct Lane {
  int lane, n, g;
  __device__ __forceinline__ Lane() {
    lane = threadIdx.x & 63;
    n = lane & 15;
    g = lane >> 4;
  }
};

// ---------------------------------------------------------------- matrix-core linear layers
// acc[jo] += W[16jo.., :] * in   for a weight stored in fp32 fragment order at `w` (LDS or global), exact fp32 MFMA.
// (Used by the bf16x6 build; in the fp16x3 build every image is a split-precision image and `linear_acc` below
// forwards to linear_acc_x6.)
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_f32(f4 (&acc)[JT_OUT], const f4 (&in)[JT_IN], const float* w, int lane) {
#ifdef TSDE_NO_PREFETCH
#pragma unroll
  for (int q = 0; q < JT_IN; ++q) {
    f4 wf[JT_OUT];
#pragma unroll
    for (int jo = 0; jo < JT_OUT; ++jo) wf[jo] = *reinterpret_cast<const f4*>(w + ((jo * JT_IN + q) * 64 + lane) * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int jo = 0; jo < JT_OUT; ++jo)
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jo][c], in[q][c], acc[jo], 0, 0, 0);
  }
#else
  // software-pipelined: the fragments of k-chunk q+1 are in flight while chunk q feeds the matrix cores
  f4 wf[2][JT_OUT];
#pragma unroll
  for (int jo = 0; jo < JT_OUT; ++jo) wf[0][jo] = *reinterpret_cast<const f4*>(w + ((jo * JT_IN) * 64 + lane) * 4);
#pragma unroll
  for (int q = 0; q < JT_IN; ++q) {
    if (q + 1 < JT_IN) {
#pragma unroll
      for (int jo = 0; jo < JT_OUT; ++jo)
        wf[(q + 1) & 1][jo] = *reinterpret_cast<const f4*>(w + ((jo * JT_IN + q + 1) * 64 + lane) * 4);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int jo = 0; jo < JT_OUT; ++jo)
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q & 1][jo][c], in[q][c], acc[jo], 0, 0, 0);
  }
#endif
}

// ---------------------------------------------------------------- bf16x6 split-precision linear layers
// fp32-accurate contraction on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16
// pieces x = x1 + x2 + x3 (8 mantissa bits each, by truncation, so the pieces keep x's sign), and
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)          (dropped terms <= 2^-24 relative)
// is evaluated with six v_mfma_f32_16x16x32_bf16 (products of bf16 are exact in fp32, accumulation is fp32).
// Six 16-cycle K=32 instructions replace eight 32-cycle K=4 fp32 instructions: 2.67x the matrix throughput at the
// same accuracy as the fp32 path (measured: max error vs fp64 1.4e-6 against 2.7e-6 for plain fp32, K=64).
// Layout: for k-step s (32 features = activation quads jt = 2s, 2s+1) lane (n, g) supplies its own 8 values
//   slot j: feature 32s + 16(j>>2) + 4g + (j&3)  ==  a[2s + (j>>2)][j&3]
// and the weight fragment of lane (i, g) is W[16jo+i][that feature]; the D fragment is the same 16x16 fp32
// layout as before, so the "accumulator is the next operand" property of the row-on-lane design is kept.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#if TSDE_SPLIT_H3
// fp16x3: x = h + l with h = fp16(x) (round toward zero, v_cvt_pkrtz_f16_f32) and l = fp16(x - h): 22 significant bits,
// the residual x - h is exact in fp32 and rounded to nearest (split_pair below).  a*b ~= a_h b_h +
// a_h b_l + a_l b_h (three v_mfma_f32_16x16x32_f16; the dropped a_l b_l is 2^-22 relative).  1.5 VALU per value to split
// against 5.5 for three bf16 pieces, and half the matrix instructions.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));

typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
// fp16x3 operand split of two values: hi = fp16(x) (round toward zero, one v_cvt_pkrtz for the pair), lo = fp16(x - hi): the residual is
// an fma of the fp16 half (extended), -1 and x -- exact in fp32 -- rounded to nearest into the lo half.  Written as plain arithmetic, the
// compiler selects v_fma_mixlo_f16 / v_fma_mixhi_f16 for it (the half read in place, the fp32 fma and the rounding in ONE instruction
// per value): three instructions per pair, none of them inline assembly, so all of them are seen by the scheduler and the hazard
// recognizer.  Two things make the selection happen: the multiplier is opaque (a literal -1 folds the fma into convert + subtract), and
// the library is built with -fno-slp-vectorize (otherwise the two fmas of a pair become one v_pk_fma_f32 behind two converts).
// History, because the earlier forms were WRONG in a way no parity test sees (identical launches disagreeing in low-order bits; found by
// comparing repeated backward calls word for word: tests/test_gpu_backward.py *_bitwise_identical, tools/encoder_bwd_repeat.py):
//  * rounds 2-3: the same three instructions as ONE inline-assembly block per k-step; a few tiles per 10^5 differed from run to run.
//  * then v_cvt_pkrtz, two inline-assembly v_fma_mix_f32, v_cvt_pk_f16_f32: ten times rarer, still there (a column of a 16-row tile off by
//    2^-11 of its smallest term in 40 % of the encoder backward calls at 64 x 128 agents).
//  * what all failing builds share is the SLP vectoriser (trajsde_amd/build.py FLAGS): with it off, the inline-assembly form is clean too
//    (-DTSDE_SPLIT_ASM, 0 differing words in 40 calls) -- but an asm block stays invisible to the hazard recognizer, so it is not shipped.
//  Edge attention alone, one box: this form 0.777 ms, the assembly form 0.771, v_dot2c_f32_f16 (-DTSDE_SPLIT_DOT2: inexact below 2^-14,
//  tools/microbench/dot2.hip) 0.788, converts + subtract (-DTSDE_SPLIT_SUB) 0.81-0.86.
__device__ __forceinline__ float opaque_minus_one() {
  float m = -1.0f;
  asm("" : "+s"(m));             // no instruction, a scalar register: nothing a vector hazard could involve
  return m;
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const hp2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  float r0, r1;                                                                    // x - h, exact
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const hf2 hh = __builtin_bit_cast(hf2, h);
#if defined(TSDE_SPLIT_ASM)                                                        // (experiments only: see the history above)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x1));
#elif defined(TSDE_SPLIT_SUB)
  r0 = x0 - float(hh[0]);                                                          // two converts + two subtracts
  r1 = x1 - float(hh[1]);
#elif defined(TSDE_SPLIT_DOT2)
  const hf2 e0 = __builtin_bit_cast(hf2, 0x0000BC00u), e1 = __builtin_bit_cast(hf2, 0xBC000000u);      // (-1, 0), (0, -1)
  r0 = __builtin_amdgcn_fdot2(hh, e0, x0, false);                                  // v_dot2c_f32_f16: x0 - h[0]
  r1 = __builtin_amdgcn_fdot2(hh, e1, x1, false);
#else
  const float m1 = opaque_minus_one();
  r0 = __builtin_fmaf(float(hh[0]), m1, x0);
  r1 = __builtin_fmaf(float(hh[1]), m1, x1);
#endif
#if TSDE_R6_SPLIT
  // Round 6 (tools/microbench/vissue.hip): v_fma_mixlo_f16 / v_fma_mixhi_f16 are QUARTER-rate instructions on this chip -- 8 cycles
  // of the SIMD each, alone or beside a second wave, like v_exp_f32 -- while v_fma_mix_f32 (fp32 result) and v_cvt_pkrtz_f16_f32 take 4.
  // The residuals stay fp32 (two v_fma_mix_f32) and are packed by a second v_cvt_pkrtz: 16 cycles per pair instead of 20.  The low
  // piece is then the residual TRUNCATED to 11 bits instead of rounded: x - (hi + lo) < 2^-22 |x| either way (22-bit operands).
  hi = hb;
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
#else
  const hf2 l = hf2{_Float16(r0), _Float16(r1)};
  hi = hb;
  lo = __builtin_bit_cast(unsigned, l);
#endif
}
// The 8 values of one k-step
__device__ __forceinline__ void split_kstep(const f4& qa, const f4& qb, u4& hi, u4& lo) {
  unsigned h[4], l[4];
  split_pair(qa[0], qa[1], h[0], l[0]);
  split_pair(qa[2], qa[3], h[1], l[1]);
  split_pair(qb[0], qb[1], h[2], l[2]);
  split_pair(qb[2], qb[3], h[3], l[3]);
  hi = u4{h[0], h[1], h[2], h[3]};
  lo = u4{l[0], l[1], l[2], l[3]};
}

// acc[jo] += W * in with W stored as two fp16 pieces in fragment order [jo][s][piece][lane][8] (pack.hip MAT6);
// jo-major like the fp32 images, so row blocks packed separately concatenate into one taller matrix
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_x6(f4 (&acc)[JT_OUT], const f4 (&in)[JT_IN], const float* w, int lane) {
  static_assert(JT_IN % 2 == 0, "k-steps cover 32 features");
  constexpr int KS = JT_IN / 2;
  u4 b1[KS], b2[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) split_kstep(in[2 * s], in[2 * s + 1], b1[s], b2[s]);
#ifdef TSDE_MFMA_CHAIN
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int jo = 0; jo < JT_OUT; ++jo) {
      const float* p = w + (jo * KS + s) * 512 + lane * 4;
      const h8 a1 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(p));
      const h8 a2 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(p + 256));
      const h8 x1 = __builtin_bit_cast(h8, b1[s]), x2 = __builtin_bit_cast(h8, b2[s]);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x2, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, x1, acc[jo], 0, 0, 0);
    }
  }
#else
  // Two output tiles at a time, their three products interleaved: an instruction that takes its accumulator from the one issued
  // right before it waits for that result (measured: ~27 cycles per instruction in such a chain against 16 when the chain
  // alternates between two accumulators -- tools/microbench/coexec.hip modes 22 / 14).  Per accumulator the order of the products
  // is unchanged: same bits.
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const h8 x1 = __builtin_bit_cast(h8, b1[s]), x2 = __builtin_bit_cast(h8, b2[s]);
#pragma unroll
    for (int jo = 0; jo < JT_OUT; jo += 2) {
      const float* p = w + (jo * KS + s) * 512 + lane * 4;
      const h8 a1 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(p));
      const h8 a2 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(p + 256));
      if (jo + 1 < JT_OUT) {
        const float* r = w + ((jo + 1) * KS + s) * 512 + lane * 4;
        const h8 c1 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(r));
        const h8 c2 = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(r + 256));
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1, acc[jo], 0, 0, 0);
        acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, x1, acc[jo + 1], 0, 0, 0);
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x2, acc[jo], 0, 0, 0);
        acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, x2, acc[jo + 1], 0, 0, 0);
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, x1, acc[jo], 0, 0, 0);
        acc[jo + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(c2, x1, acc[jo + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                   // (left alone, the scheduler regroups the six into two chains of three)
      } else {
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1, acc[jo], 0, 0, 0);
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x2, acc[jo], 0, 0, 0);
        acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, x1, acc[jo], 0, 0, 0);
      }
    }
  }
#endif
}

// the same contraction for TWO row tiles of the wave at once (weight fragments read from LDS once)
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_x6_2(f4 (&acc0)[JT_OUT], f4 (&acc1)[JT_OUT], const f4 (&in0)[JT_IN], const f4 (&in1)[JT_IN],
                                                const float* w, int lane) {
  static_assert(JT_IN % 2 == 0, "k-steps cover 32 features");
  constexpr int KS = JT_IN / 2;
  u4 p1[KS], p2[KS], q1[KS], q2[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    split_kstep(in0[2 * s], in0[2 * s + 1], p1[s], p2[s]);
    split_kstep(in1[2 * s], in1[2 * s + 1], q1[s], q2[s]);
  }
  // the weight fragments of step (s, jo) + 1 are read from LDS while step (s, jo) feeds the matrix cores (two fragment pairs
  // in registers): the ds_read latency stays off the chain of matrix instructions
  constexpr int STEPS = KS * JT_OUT;
  u4 f1[2], f2[2];
  f1[0] = *reinterpret_cast<const u4*>(w + lane * 4);
  f2[0] = *reinterpret_cast<const u4*>(w + lane * 4 + 256);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int s = i / JT_OUT, jo = i % JT_OUT;
    if (i + 1 < STEPS) {
      const int s2 = (i + 1) / JT_OUT, jo2 = (i + 1) % JT_OUT;
      const float* p = w + (jo2 * KS + s2) * 512 + lane * 4;
      f1[(i + 1) & 1] = *reinterpret_cast<const u4*>(p);
      f2[(i + 1) & 1] = *reinterpret_cast<const u4*>(p + 256);
      __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink the reads to two instructions before their use)
    }
    const h8 a1 = __builtin_bit_cast(h8, f1[i & 1]), a2 = __builtin_bit_cast(h8, f2[i & 1]);
    const h8 x1 = __builtin_bit_cast(h8, p1[s]), x2 = __builtin_bit_cast(h8, p2[s]);
    const h8 y1 = __builtin_bit_cast(h8, q1[s]), y2 = __builtin_bit_cast(h8, q2[s]);
    acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1, acc0[jo], 0, 0, 0);
    acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, y1, acc1[jo], 0, 0, 0);
    acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x2, acc0[jo], 0, 0, 0);
    acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, y2, acc1[jo], 0, 0, 0);
    acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, x1, acc0[jo], 0, 0, 0);
    acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, y1, acc1[jo], 0, 0, 0);
  }
}
#ifndef TSDE_FRAG_AHEAD
#define TSDE_FRAG_AHEAD 1
#endif
// the same contraction for NT row tiles of one wave: every weight fragment read from LDS feeds 3 NT matrix instructions.
// Per tile the products and their order are those of linear_acc_x6 (same bits).
template <int NT, int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_x6_n(f4 (&acc)[NT][JT_OUT], const f4 (&in)[NT][JT_IN], const float* w, int lane) {
  static_assert(JT_IN % 2 == 0, "k-steps cover 32 features");
  constexpr int KS = JT_IN / 2;
  u4 xh[NT][KS], xl[NT][KS];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) split_kstep(in[t][2 * s], in[t][2 * s + 1], xh[t][s], xl[t][s]);
  constexpr int STEPS = KS * JT_OUT;
  constexpr int PF = TSDE_FRAG_AHEAD;                      // matrix steps (6 NT... 3 NT instructions) a fragment pair is read ahead of its use
  u4 f1[PF + 1], f2[PF + 1];
#pragma unroll
  for (int i = 0; i < PF && i < STEPS; ++i) {
    const float* p = w + ((i % JT_OUT) * KS + i / JT_OUT) * 512 + lane * 4;
    f1[i] = *reinterpret_cast<const u4*>(p);
    f2[i] = *reinterpret_cast<const u4*>(p + 256);
  }
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int s = i / JT_OUT, jo = i % JT_OUT;
    if (i + PF < STEPS) {
      const int s2 = (i + PF) / JT_OUT, jo2 = (i + PF) % JT_OUT;
      const float* p = w + (jo2 * KS + s2) * 512 + lane * 4;
      f1[(i + PF) % (PF + 1)] = *reinterpret_cast<const u4*>(p);
      f2[(i + PF) % (PF + 1)] = *reinterpret_cast<const u4*>(p + 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    const h8 a1 = __builtin_bit_cast(h8, f1[i % (PF + 1)]), a2 = __builtin_bit_cast(h8, f2[i % (PF + 1)]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, xh[t][s]), acc[t][jo], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, xl[t][s]), acc[t][jo], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(h8, xh[t][s]), acc[t][jo], 0, 0, 0);
  }
}

// ---- round 6: the same products with the operand split and the matrix steps as separate calls, so that a kernel can hand the steps
// independent vector work ("atoms") to issue in the shadow of the matrix instructions.  A wave's stream is in order: a vector stage
// that sits between two products runs with the matrix pipe idle, whereas up to two vector instructions per matrix instruction issued
// BETWEEN them cost nothing (tools/microbench/coexec.hip modes 18 / 19: 48 x (mfma + 2 fma) = 14.9 cycles per group).
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for_(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_<N, F, I + 1>(static_cast<F&&>(f));
  }
}
struct NoAtoms {
  template <class I>
  __device__ __forceinline__ void operator()(I) const {}
};
// operand pieces of NT tiles (the split of linear_acc_x6_n, element for element)
template <int NT>
__device__ __forceinline__ void split_rows_n(u4 (&xh)[NT][2], u4 (&xl)[NT][2], const f4 (&in)[NT][4]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) split_kstep(in[t][2 * s], in[t][2 * s + 1], xh[t][s], xl[t][s]);
}
// ReLU (as an integer max on the bits of a matrix result: in2_mfma_relu_n) + split of one k-step of one tile
__device__ __forceinline__ void relu_split_kstep(const f4& ya, const f4& yb, u4& hi, u4& lo) {
  f4 r0, r1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r0[k] = __int_as_float(max(__float_as_int(ya[k]), 0));
    r1[k] = __int_as_float(max(__float_as_int(yb[k]), 0));
  }
  split_kstep(r0, r1, hi, lo);
}
// acc[t][jo] += W x_t over the two k-steps (64 inputs), jo = 0 .. JT_OUT-1, steps in the order of linear_acc_x6_n (k-step major), every
// fragment pair read one step ahead and feeding 3 NT matrix instructions; after the matrix instructions of step i the caller's
// atoms(i) -- the compiler may move them among that step's matrix instructions, not across the step's fence.  Same bits as
// linear_acc_x6_n per accumulator.
template <int NT, int JT_OUT, class Atoms>
__device__ __forceinline__ void product_x6_n(f4 (&acc)[NT][JT_OUT], const u4 (&xh)[NT][2], const u4 (&xl)[NT][2], const float* w, int lane,
                                             Atoms&& atoms) {
  constexpr int KS = 2, STEPS = KS * JT_OUT;
  u4 f1[2], f2[2];
  f1[0] = *reinterpret_cast<const u4*>(w + lane * 4);
  f2[0] = *reinterpret_cast<const u4*>(w + lane * 4 + 256);
  static_for_<STEPS>([&](auto I) {
    constexpr int i = decltype(I)::value, s = i / JT_OUT, jo = i % JT_OUT;
    if constexpr (i + 1 < STEPS) {
      constexpr int s2 = (i + 1) / JT_OUT, jo2 = (i + 1) % JT_OUT;
      const float* p = w + (jo2 * KS + s2) * 512 + lane * 4;
      f1[(i + 1) & 1] = *reinterpret_cast<const u4*>(p);
      f2[(i + 1) & 1] = *reinterpret_cast<const u4*>(p + 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    const h8 a1 = __builtin_bit_cast(h8, f1[i & 1]), a2 = __builtin_bit_cast(h8, f2[i & 1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, xh[t][s]), acc[t][jo], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, xl[t][s]), acc[t][jo], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(h8, xh[t][s]), acc[t][jo], 0, 0, 0);
    atoms(I);
  });
}
#else
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned M = 0xFFFF0000u;
  const float h0 = __uint_as_float(__float_as_uint(x0) & M), h1 = __uint_as_float(__float_as_uint(x1) & M);
  hi = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
  const float r0 = x0 - h0, r1 = x1 - h1;                       // exact
  const float m0 = __uint_as_float(__float_as_uint(r0) & M), m1 = __uint_as_float(__float_as_uint(r1) & M);
  mid = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  const float l0 = r0 - m0, l1 = r1 - m1;                       // exact, <= 8 significant bits: already bf16
  lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

// split the 8 slot values of one k-step (two activation quads) into three packed bf16x8 operands
__device__ __forceinline__ void split_kstep(const f4& qa, const f4& qb, u4& hi, u4& mid, u4& lo) {
  unsigned h[4], m[4], l[4];
  split_pair(qa[0], qa[1], h[0], m[0], l[0]);
  split_pair(qa[2], qa[3], h[1], m[1], l[1]);
  split_pair(qb[0], qb[1], h[2], m[2], l[2]);
  split_pair(qb[2], qb[3], h[3], m[3], l[3]);
  hi = u4{h[0], h[1], h[2], h[3]};
  mid = u4{m[0], m[1], m[2], m[3]};
  lo = u4{l[0], l[1], l[2], l[3]};
}

// acc[jo] += W * in with W stored as three bf16 pieces in fragment order [jo][s][piece][lane][8] (pack.hip MAT6)
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_x6(f4 (&acc)[JT_OUT], const f4 (&in)[JT_IN], const float* w, int lane) {
  static_assert(JT_IN % 2 == 0, "k-steps cover 32 features");
  constexpr int KS = JT_IN / 2;
  u4 b1[KS], b2[KS], b3[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) split_kstep(in[2 * s], in[2 * s + 1], b1[s], b2[s], b3[s]);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int jo = 0; jo < JT_OUT; ++jo) {
      const float* p = w + (jo * KS + s) * 768 + lane * 4;
      const bf8 a1 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p));
      const bf8 a2 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + 256));
      const bf8 a3 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + 512));
      const bf8 x1 = __builtin_bit_cast(bf8, b1[s]), x2 = __builtin_bit_cast(bf8, b2[s]), x3 = __builtin_bit_cast(bf8, b3[s]);
      // one fp32 accumulation chain; each add rounds at 2^-24 of the running sum, like an fp32 fma chain would
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2, acc[jo], 0, 0, 0);
      acc[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1, acc[jo], 0, 0, 0);
    }
  }
}

// The same contraction for TWO row tiles of the wave at once: every weight fragment read from LDS feeds both tiles'
// matrix-core instructions, which halves the LDS traffic per tile.  (LDS bandwidth, 128 B/clk/CU, is what bounds the
// edge kernels: a 64x64 bf16x6 product reads 24 KB of fragments per wave.)  Per tile the arithmetic and its order are
// those of linear_acc_x6, so the results are bitwise the same.
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc_x6_2(f4 (&acc0)[JT_OUT], f4 (&acc1)[JT_OUT], const f4 (&in0)[JT_IN], const f4 (&in1)[JT_IN],
                                                const float* w, int lane) {
  static_assert(JT_IN % 2 == 0, "k-steps cover 32 features");
  constexpr int KS = JT_IN / 2;
  u4 p1[KS], p2[KS], p3[KS], q1[KS], q2[KS], q3[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    split_kstep(in0[2 * s], in0[2 * s + 1], p1[s], p2[s], p3[s]);
    split_kstep(in1[2 * s], in1[2 * s + 1], q1[s], q2[s], q3[s]);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int jo = 0; jo < JT_OUT; ++jo) {
      const float* p = w + (jo * KS + s) * 768 + lane * 4;
      const bf8 a1 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p));
      const bf8 a2 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + 256));
      const bf8 a3 = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(p + 512));
      const bf8 x1 = __builtin_bit_cast(bf8, p1[s]), x2 = __builtin_bit_cast(bf8, p2[s]), x3 = __builtin_bit_cast(bf8, p3[s]);
      const bf8 y1 = __builtin_bit_cast(bf8, q1[s]), y2 = __builtin_bit_cast(bf8, q2[s]), y3 = __builtin_bit_cast(bf8, q3[s]);
      // the two tiles' chains are independent: interleaving them also hides the matrix-core latency of each chain
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y1, acc1[jo], 0, 0, 0);
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y2, acc1[jo], 0, 0, 0);
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, y1, acc1[jo], 0, 0, 0);
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y3, acc1[jo], 0, 0, 0);
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, y2, acc1[jo], 0, 0, 0);
      acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1, acc0[jo], 0, 0, 0);
      acc1[jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, y1, acc1[jo], 0, 0, 0);
    }
  }
}

#endif   // TSDE_SPLIT_H3

// The general-purpose contraction of the node-level and backward kernels.  fp16x3 build: a 64x64 split-precision image
// is exactly as large as the fp32 one (MAT64X6 == MAT64), so pack.hip writes EVERY matrix image as fp16 planes and this
// is the split-precision product -- a 64-k product costs 6 x 16 matrix-core cycles instead of 16 x 32.  bf16x6 build:
// fp32 fragment images and the exact fp32 instruction.
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_acc(f4 (&acc)[JT_OUT], const f4 (&in)[JT_IN], const float* w, int lane) {
#if TSDE_SPLIT_H3
  linear_acc_x6<JT_OUT, JT_IN>(acc, in, w, lane);
#else
  linear_acc_f32<JT_OUT, JT_IN>(acc, in, w, lane);
#endif
}

// per-feature vector (bias, LayerNorm gamma/beta, ...) stored plainly: v[16*jt + 4*g + c]
template <int JT>
__device__ __forceinline__ void load_vec(f4 (&out)[JT], const float* v, int g) {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) out[jt] = *reinterpret_cast<const f4*>(v + 16 * jt + 4 * g);
}

template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear(f4 (&out)[JT_OUT], const f4 (&in)[JT_IN], const float* w, const float* bias,
                                       const Lane& L) {
  load_vec<JT_OUT>(out, bias, L.g);
  linear_acc<JT_OUT, JT_IN>(out, in, w, L.lane);
}

template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_x6(f4 (&out)[JT_OUT], const f4 (&in)[JT_IN], const float* w, const float* bias, const Lane& L) {
  load_vec<JT_OUT>(out, bias, L.g);
  linear_acc_x6<JT_OUT, JT_IN>(out, in, w, L.lane);
}

// ---------------------------------------------------------------- row reductions (4 lanes hold one row)
// all-reduce over the 4 lanes {n, n+16, n+32, n+48} that hold one row, without touching LDS:
// v_permlane16_swap exchanges odd 16-lane rows of its first operand with even rows of the second, so swapping a
// register with a copy of itself yields (r0,r0,r2,r2) and (r1,r1,r3,r3); v_permlane32_swap does the same for halves.
__device__ __forceinline__ float xor16_sum(float v) {
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float row_sum(float v) { return xor32_sum(xor16_sum(v)); }
__device__ __forceinline__ float row_max(float v) {
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// Adjoint product of the backward kernels: acc += W^T d for an adjoint tile d and the transposed image of W.
// Adjoints are small and span many binades (1e-9 .. 1), outside what fp16 pieces hold, so in the fp16x3 build each ROW
// of d (a lane's 16 values + its 3 partner lanes) is first scaled by the power of two that brings its largest element
// into [1, 2) -- exact -- multiplied in split precision, and the result is scaled back while it is accumulated.
// Relative to the row's largest element this keeps 2^-22; an fp32 product is no better than 2^-24 of the same scale.
template <int JT_OUT, int JT_IN>
__device__ __forceinline__ void linear_adj(f4 (&acc)[JT_OUT], const f4 (&d)[JT_IN], const float* wt, const Lane& L) {
#if TSDE_SPLIT_H3
  float m = 0.f;
#pragma unroll
  for (int jt = 0; jt < JT_IN; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) m = fmaxf(m, fabsf(d[jt][c]));
  m = row_max(m);
  const unsigned e = __float_as_uint(m) & 0x7F800000u;          // 2^floor(log2 m); 0 for an all-zero row
  const float up = __uint_as_float(0x7F000000u - e), down = __uint_as_float(e);
  f4 ds[JT_IN], t[JT_OUT];
#pragma unroll
  for (int jt = 0; jt < JT_IN; ++jt) ds[jt] = d[jt] * up;
#pragma unroll
  for (int jo = 0; jo < JT_OUT; ++jo) t[jo] = f4{0.f, 0.f, 0.f, 0.f};
  linear_acc_x6<JT_OUT, JT_IN>(t, ds, wt, L.lane);
#pragma unroll
  for (int jo = 0; jo < JT_OUT; ++jo)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[jo][c] = fmaf(t[jo][c], down, acc[jo][c]);
#else
  linear_acc_f32<JT_OUT, JT_IN>(acc, d, wt, L.lane);
#endif
}

// 1/sqrt(x) for the LayerNorms (x = var + eps >= 1e-5): v_rsq_f32 plus one Newton step, 5 instructions with one
// transcendental -- `1.0f / sqrtf(x)` expands to the IEEE sqrt and division sequences, ~25 instructions per call,
// which was 15 % of the VALU stream of the edge kernel.  Within 1 ulp of the correctly rounded value.
// Round 6: the Newton step is gone.  v_rsq_f32 is specified to 1 ulp; the step bought the last half ulp of a factor whose consumers are
// 22-bit operands (2^-22), for four more vector instructions per LayerNorm row (32 an iteration of the edge kernel, whose SIMDs are
// bound by the instructions they issue: profiles/r06_edge_issue_model.md).  -DTSDE_R6_RSQ=0 restores it.
#ifndef TSDE_R6_RSQ
#define TSDE_R6_RSQ 1
#endif
__device__ __forceinline__ float rsqrt_nr(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
#if TSDE_R6_RSQ
  return y;
#else
  return y * fmaf(-0.5f * x * y, y, 1.5f);
#endif
}

template <int JT>
__device__ __forceinline__ void layer_norm(f4 (&a)[JT], const float* gamma, const float* beta, int g) {
  float s = 0.f;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) s += (a[jt][0] + a[jt][1]) + (a[jt][2] + a[jt][3]);
  const float mean = row_sum(s) * (1.0f / (16 * JT));
  float v = 0.f;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = a[jt][c] - mean;
      a[jt][c] = d;
      v += d * d;
    }
  const float rstd = rsqrt_nr(row_sum(v) * (1.0f / (16 * JT)) + 1e-5f);
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    const f4 ga = *reinterpret_cast<const f4*>(gamma + 16 * jt + 4 * g);
    const f4 be = *reinterpret_cast<const f4*>(beta + 16 * jt + 4 * g);
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] = a[jt][c] * rstd * ga[c] + be[c];
  }
}

// LayerNorm of a row that arrives FEATURE-CENTRED (its producing matrix and bias had the mean over the 64 outputs removed
// when they were packed, pack.hip PK_MAT6_CENTRED): the mean is zero by construction, the variance is what is left to reduce
__device__ __forceinline__ float centred_rstd(const f4 (&d)[4]) {
  float v = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) v = fmaf(d[jt][c], d[jt][c], v);
  return rsqrt_nr(row_sum(v) * (1.0f / 64) + 1e-5f);
}

// dot of a 64-feature activation with a plain vector, reduced over the row
__device__ __forceinline__ float row_dot(const f4 (&a)[4], const float* w, int g) {
  float s = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const f4 wv = *reinterpret_cast<const f4*>(w + 16 * jt + 4 * g);
#pragma unroll
    for (int c = 0; c < 4; ++c) s = fmaf(a[jt][c], wv[c], s);
  }
  return row_sum(s);
}

// ---------------------------------------------------------------- pointwise
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(e^{2x}+1): v_exp_f32 + v_rcp_f32, abs error ~2e-7; saturates correctly at +-inf
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
// tanh of a pre-activation that arrives ALREADY multiplied by 2 / ln 2 (the layer in front of it was packed with that factor:
// pack.hip Packer::scale): v_exp_f32, add, v_rcp_f32, fma -- the multiply of fast_tanh is gone (64 of ~890 vector instructions of an
// SDE tile-step)
constexpr float TANH_PRESCALE = 2.8853900817779268f;
__device__ __forceinline__ float tanh_prescaled(float u) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(u) + 1.0f); }

template <int JT>
__device__ __forceinline__ void relu(f4 (&a)[JT]) {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] = fmaxf(a[jt][c], 0.f);
}
template <int JT>
__device__ __forceinline__ void tanh_(f4 (&a)[JT]) {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] = fast_tanh(a[jt][c]);
}
// sigmoid of a pre-activation that arrives multiplied by -1 / ln 2: 1 / (1 + 2^u)
constexpr float SIGMOID_PRESCALE = -1.4426950408889634f;
__device__ __forceinline__ float sigmoid_prescaled(float u) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u)); }
template <int JT>
__device__ __forceinline__ void tanh_prescaled_(f4 (&a)[JT]) {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] = tanh_prescaled(a[jt][c]);
}
template <int JT>
__device__ __forceinline__ void sigmoid_(f4 (&a)[JT]) {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[jt][c] = fast_sigmoid(a[jt][c]);
}

// first layer of the 2-d input embeddings: out[f] = W[f][0]*x0 + W[f][1]*x1 + b[f], W row-major [64][2]
__device__ __forceinline__ void linear_in2(f4 (&out)[4], float x0, float x1, const float* w, const float* b, int g) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int f0 = 16 * jt + 4 * g;
    const f4 wa = *reinterpret_cast<const f4*>(w + 2 * f0);
    const f4 wb = *reinterpret_cast<const f4*>(w + 2 * f0 + 4);
    const f4 bb = *reinterpret_cast<const f4*>(b + f0);
    out[jt][0] = fmaf(x1, wa[1], fmaf(x0, wa[0], bb[0]));
    out[jt][1] = fmaf(x1, wa[3], fmaf(x0, wa[2], bb[1]));
    out[jt][2] = fmaf(x1, wb[1], fmaf(x0, wb[0], bb[2]));
    out[jt][3] = fmaf(x1, wb[3], fmaf(x0, wb[2], bb[3]));
  }
}

// ReLU(LayerNorm(Linear(2,64)(x))) with the LayerNorm statistics in closed form (layouts.hpp In2L; pack.hip PK_LN2):
// 3 fma + 1 max per element instead of 2 fma for the layer, two row reductions and 3 more per element for the LayerNorm.
// rstd of the layer's LayerNorm for inputs (x0, x1): var = |L^T (x0, x1, 1)|^2
__device__ __forceinline__ float in2_rstd(float x0, float x1, const float* c) {
  const f4 ch0 = *reinterpret_cast<const f4*>(c + 192), ch1 = *reinterpret_cast<const f4*>(c + 196);
  const float a = fmaf(ch0[0], x0, fmaf(ch0[1], x1, ch0[2])), b = fmaf(ch0[3], x1, ch1[0]);
  return rsqrt_nr(fmaf(a, a, fmaf(b, b, ch1[1] * ch1[1])) + 1e-5f);
}
// features f0 .. f0+3 of ReLU(LayerNorm(Linear(2,64)(x))), given x * rstd and rstd
__device__ __forceinline__ f4 in2_ln_relu4(float x0r, float x1r, float rstd, const float* c, const float* beta, int f0) {
  const f4 w0 = *reinterpret_cast<const f4*>(c + f0), w1 = *reinterpret_cast<const f4*>(c + 64 + f0);
  const f4 gb = *reinterpret_cast<const f4*>(c + 128 + f0), be = *reinterpret_cast<const f4*>(beta + f0);
  f4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaf(w0[k], x0r, fmaf(w1[k], x1r, fmaf(gb[k], rstd, be[k]))), 0.f);
  return o;
}
__device__ __forceinline__ void in2_ln_relu(f4 (&out)[4], float x0, float x1, const float* c, const float* beta, int g) {
  const float rstd = in2_rstd(x0, x1, c);
  const float x0r = x0 * rstd, x1r = x1 * rstd;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) out[jt] = in2_ln_relu4(x0r, x1r, rstd, c, beta, 16 * jt + 4 * g);
}

#if TSDE_SPLIT_H3
// The same block on the matrix cores (layouts.hpp IN2F): the row operand x0r_h x1r_h | x0r_l x1r_l | r_h r_l | 1 1 of one row
// (hi / lo pieces as in split_pair: no inline assembly, see there) ...
__device__ __forceinline__ u4 in2_operand(float x0r, float x1r, float rstd) {
  unsigned h, l;
  split_pair(x0r, x1r, h, l);
  const hp2 rh = __builtin_amdgcn_cvt_pkrtz(rstd, 0.f);
#if TSDE_R6_IN2
  // r_h | r_l: the low piece rounded straight into the high half of the register that holds r_h (v_fma_mixhi_f16 keeps the low
  // half): no mask, no shift-or (round 5: v_and + v_lshl_or per operand)
  hf2 hr2 = __builtin_bit_cast(hf2, rh);
  const float rr = __builtin_fmaf(float(hr2[0]), opaque_minus_one(), rstd);        // rstd - fp16(rstd), exact
  hr2[1] = _Float16(rr);
  // (the constant word through an opaque VECTOR register: with a literal there the compiler materialises the whole 128-bit tuple from
  //  scalar registers -- two v_mov_b64 -- and then overwrites three of its words, five moves per operand, twenty an iteration)
  unsigned ones = 0x3C003C00u;
  asm("" : "+v"(ones));
  return u4{h, l, __builtin_bit_cast(unsigned, hr2), ones};
#else
  const unsigned rhb = __builtin_bit_cast(unsigned, rh);
  float rr;                                                                        // rstd - fp16(rstd), exact
  rr = __builtin_fmaf(float(__builtin_bit_cast(hf2, rh)[0]), opaque_minus_one(), rstd);
  const hf2 rl = hf2{_Float16(rr), _Float16(0.f)};
  const unsigned hr = (rhb & 0xFFFFu) | (__builtin_bit_cast(unsigned, rl) << 16);  // r_h | r_l
  return u4{h, l, hr, 0x3C003C00u};
#endif
}
// ... and ReLU(LayerNorm(Linear(2,64)(x))) of NT row tiles: 4 fragments read once, 4 NT matrix instructions, 16 NT max
template <int NT>
__device__ __forceinline__ void in2_operands_n(u4 (&b)[NT], const float (&x0)[NT], const float (&x1)[NT], const float* c) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float rstd = in2_rstd(x0[t], x1[t], c);
    b[t] = in2_operand(x0[t] * rstd, x1[t] * rstd, rstd);
  }
}
template <int NT>
__device__ __forceinline__ void in2_mfma_relu_n(f4 (&out)[NT][4], const u4 (&b)[NT], const float* frag, int lane);
template <int NT>
__device__ __forceinline__ void in2_mfma_relu_n(f4 (&out)[NT][4], const float (&x0)[NT], const float (&x1)[NT], const float* c,
                                                const float* frag, int lane) {
  u4 b[NT];
  in2_operands_n<NT>(b, x0, x1, c);
  in2_mfma_relu_n<NT>(out, b, frag, lane);
}
template <int NT>
__device__ __forceinline__ void in2_mfma_relu_n(f4 (&out)[NT][4], const u4 (&b)[NT], const float* frag, int lane) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(frag + jt * 256 + lane * 4));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f4 y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, b[t]), f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      // ReLU as an integer max on the bits (negative floats are negative integers): fmaxf on a matrix result makes the
      // compiler canonicalise it first with a second v_max, and inline assembly would hide the matrix-result hazard from it
#pragma unroll
      for (int k = 0; k < 4; ++k) out[t][jt][k] = __int_as_float(max(__float_as_int(y[k]), 0));
    }
  }
}
// the 4 NT matrix instructions of one first layer alone, back to back (results before the ReLU): a kernel that has other work for the
// ~8 wait states between a matrix instruction and the first reader of its result issues all of a tile pair's first layers at once
template <int NT>
__device__ __forceinline__ void in2_mfma_raw_n(f4 (&y)[NT][4], const u4 (&b)[NT], const float* frag, int lane) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(frag + jt * 256 + lane * 4));
#pragma unroll
    for (int t = 0; t < NT; ++t) y[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, b[t]), f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
}
#endif

// ---------------------------------------------------------------- HBM <-> activation tiles ([rows][64] fp32)
__device__ __forceinline__ void load_row(f4 (&a)[4], const float* base, int64_t row, int g) {
  const float* p = base + row * D + 4 * g;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) a[jt] = *reinterpret_cast<const f4*>(p + 16 * jt);
}
__device__ __forceinline__ void store_row(const f4 (&a)[4], float* base, int64_t row, int g) {
  float* p = base + row * D + 4 * g;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<f4*>(p + 16 * jt) = a[jt];
}

// A 16 x 64 fp32 tile to HBM as WHOLE ROWS.  store_row writes from the row-on-lane layout: each of its four store instructions touches
// 16 rows, 64 bytes of each.  Through a wave-private LDS tile (16 x 68 floats, written row-on-lane, read back with 16 lanes per row)
// every store instruction writes four whole consecutive rows -- 1 KB contiguous.  Measured where a kernel writes its rows once and
// streams: the relative-pose embedding 0.187 -> 0.177 ms, the embedding-backward tail (three slabs) 1.03 -> 0.97 ms and 0.28 -> 0.23 ms (0.77 / 0.18 with no stores at all).
// `tile`: the calling wave's own ROWSTAGE floats; rows >= nrows are not stored.
constexpr int ROWSTAGE = 16 * 68;
__device__ __forceinline__ void store_tile_rows(float* tile, const f4 (&a)[4], float* __restrict__ out, int64_t row0, int64_t nrows,
                                                const Lane& L) {
  __builtin_amdgcn_wave_barrier();                            // the tile's previous readers are done (same wave, in order)
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<f4*>(tile + L.n * 68 + 16 * jt + 4 * L.g) = a[jt];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * q + (L.lane >> 4);
    const f4 v = *reinterpret_cast<const f4*>(tile + r * 68 + 4 * (L.lane & 15));
    if (row0 + r < nrows) *reinterpret_cast<f4*>(out + (row0 + r) * 64 + 4 * (L.lane & 15)) = v;
  }
}

#if TSDE_SPLIT_H3
// The same tile as SPLIT-PRECISION rows: row r = fp16 hi[64] | fp16 lo[64] (256 bytes, like the fp32 row), hi / lo the operand pieces of
// split_pair.  Lane (n, g) holds features 16 jt + 4 g + c of row n: 8 bytes of each plane per jt.
__device__ __forceinline__ void store_tile_rows_split(float* tile, const f4 (&a)[4], float* __restrict__ out, int64_t row0, int64_t nrows,
                                                      const Lane& L) {
  __builtin_amdgcn_wave_barrier();                            // the tile's previous readers are done (same wave, in order)
  unsigned* const t = reinterpret_cast<unsigned*>(tile);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    unsigned h0, l0, h1, l1;
    split_pair(a[jt][0], a[jt][1], h0, l0);
    split_pair(a[jt][2], a[jt][3], h1, l1);
    *reinterpret_cast<uint2*>(t + L.n * 68 + 8 * jt + 2 * L.g) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(t + L.n * 68 + 32 + 8 * jt + 2 * L.g) = uint2{l0, l1};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * q + (L.lane >> 4);
    const f4 v = *reinterpret_cast<const f4*>(tile + r * 68 + 4 * (L.lane & 15));
    if (row0 + r < nrows) *reinterpret_cast<f4*>(out + (row0 + r) * 64 + 4 * (L.lane & 15)) = v;
  }
}
#endif

// ---- "hidden state stored bf16 between kernels" (BASELINE configs[4]): the same [rows][64] activations with 2-byte elements.
// Arithmetic stays fp32 in registers; a row is rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) when it is stored and
// widened exactly when it is loaded.  `bf16` is uniform per launch (trajsde_state_storage), so the branch is a scalar one.
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 widen4(const bf4 v) { return f4{float(v[0]), float(v[1]), float(v[2]), float(v[3])}; }
__device__ __forceinline__ bf4 narrow4(const f4 v) { return bf4{__bf16(v[0]), __bf16(v[1]), __bf16(v[2]), __bf16(v[3])}; }
__device__ __forceinline__ void load_row_st(f4 (&a)[4], const void* base, int64_t row, int g, bool bf16) {
  if (bf16) {
    const __bf16* p = static_cast<const __bf16*>(base) + row * D + 4 * g;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) a[jt] = widen4(*reinterpret_cast<const bf4*>(p + 16 * jt));
  } else {
    load_row(a, static_cast<const float*>(base), row, g);
  }
}
__device__ __forceinline__ void store_row_st(const f4 (&a)[4], void* base, int64_t row, int g, bool bf16) {
  if (bf16) {
    __bf16* p = static_cast<__bf16*>(base) + row * D + 4 * g;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<bf4*>(p + 16 * jt) = narrow4(a[jt]);
  } else {
    store_row(a, static_cast<float*>(base), row, g);
  }
}
// one element of a row (kernels with lane = feature)
__device__ __forceinline__ float load_elem_st(const void* base, int64_t idx, bool bf16) {
  return bf16 ? float(static_cast<const __bf16*>(base)[idx]) : static_cast<const float*>(base)[idx];
}

// Weights in LDS are loop-invariant, so LICM would hoist every fragment read out of the tile / time loops
// and spill hundreds of VGPRs; a compiler-only memory barrier at the top of each loop body keeps the
// ds_reads next to their MFMAs (no instruction is emitted).
__device__ __forceinline__ void keep_lds_reads_here() { asm volatile("" ::: "memory"); }

// Static priority for the second-dispatched half of a workgroup's waves (MI355X_MICROARCH.md "Two waves per SIMD", item 4): the two
// waves of a SIMD are arbitrated by priority, then age, and the younger one loses every contested slot; one s_setprio 1 for that half
// before the main loop (no per-phase flips) hands it the older half's timing.  The guard must be a SCALAR comparison (readfirstlane):
// on a vector condition the compiler makes the scalar instruction unconditional.  -DTSDE_PRIO_YOUNG=1 (A/B: profiles/r05_ab_runs.md).
#ifndef TSDE_PRIO_YOUNG
#define TSDE_PRIO_YOUNG 0
#endif
__device__ __forceinline__ void prioritize_younger_half() {
#if TSDE_PRIO_YOUNG
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), half = __builtin_amdgcn_readfirstlane(int(blockDim.x >> 7));
  if (wave >= half) __builtin_amdgcn_s_setprio(1);
#endif
}

// stage a packed weight blob into LDS (whole workgroup), then barrier.  EIGHT loads of a thread are requested before the first is
// stored: written as `for (i ..) dst[i] = src[i]` the compiler emitted load, wait, store per iteration -- a 128 KB image staged by
// 512 threads was 16 global-memory latencies in a row, ~10 us at the head of EVERY launch of every tile kernel (a 512-tile k_ffn
// launch took 14 us for 2 us of work; round 5, seen in the listing).
// (stage_copy: the copy alone -- n_floats a multiple of 4, both sides 16-byte aligned; no barrier)
__device__ __forceinline__ void stage_copy(float* lds, const float* blob, int n_floats) {
  const f4* src = reinterpret_cast<const f4*>(blob);
  f4* dst = reinterpret_cast<f4*>(lds);
  const int n4 = n_floats >> 2, step = blockDim.x;
  int i = threadIdx.x;
  for (; i + 7 * step < n4; i += 8 * step) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * step] = v[u];
  }
  for (; i + step < n4; i += 2 * step) {
    const f4 a = src[i], b = src[i + step];
    dst[i] = a;
    dst[i + step] = b;
  }
  for (; i < n4; i += step) dst[i] = src[i];
}
__device__ __forceinline__ void stage_blob(float* lds, const float* blob, int n_floats) {
  stage_copy(lds, blob, n_floats);
  __syncthreads();
}

}  // namespace tsde
