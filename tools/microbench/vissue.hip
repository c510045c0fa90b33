// vissue.hip -- what ONE SIMD issues per cycle when its waves run nothing but vector instructions of one kind, and what is left of
// that beside a partner wave that runs matrix instructions.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/vissue tools/microbench/vissue.hip && tools/microbench/vissue
//
// Round 6 question: tools/microbench/coexec.hip shows two waves of independent v_fma_f32 retiring one
// instruction per 2.4 SIMD cycles (mode 1) and vector phases overlapping a partner's matrix phases (modes 3 / 4), yet the edge
// kernel with its matrix instructions removed still needs ~4.5 SIMD cycles per vector instruction at two waves per SIMD (DESIGN
// section 5 "Round 2", ablation).  Which property of the kernel's vector stream -- instruction kind, encoding size, literal
// operands, code size (the loop body is ~11 KB), dependent chains -- costs the factor two?  Every kind below is measured
//   * alone, one and two waves per SIMD, loop bodies of 192 and of 1 536 instructions (1.5 KB and 12 KB of code);
//   * as the V wave of a SIMD whose other wave issues 48 matrix instructions per 192 vector instructions (coexec mode 2).
// Cycles are s_memtime ticks of the stamped waves (median over workgroups), per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define REP8(I) I(0) I(2) I(4) I(6) I(8) I(10) I(12) I(14)
// one block = 16 instructions, register i of the wave's 16
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define I_MUL32(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\n\t"
#define I_FMAC32(i) "v_fmac_f32_e32 %" #i ", %16, %17\n\t"
#define I_LIT(i) "v_add_f32_e32 %" #i ", 0x3727c5ac, %" #i "\n\t"
#define I_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %16, 0x3fc00000\n\t"
#define I_MIXLO(i) "v_fma_mixlo_f16 %" #i ", %" #i ", %16, %17 op_sel_hi:[1,0,0]\n\t"
#define I_PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %16\n\t"
#define I_MAXI(i) "v_max_i32_e32 %" #i ", 0, %" #i "\n\t"
#define I_MAXF(i) "v_max_f32_e32 %" #i ", %16, %" #i "\n\t"
#define I_EXP(i) "v_exp_f32_e32 %" #i ", %" #i "\n\t"
#define I_RSQ(i) "v_rsq_f32_e32 %" #i ", %" #i "\n\t"
#define I_CND(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, s[2:3]\n\t"
#define I_MOV(i) "v_mov_b32_e32 %" #i ", %16\n\t"
#define I_PERM16(i) "v_permlane16_swap_b32_e32 %" #i ", %16\n\t"
#define I_CHAIN(i) "v_fma_f32 %0, %0, %16, %17\n\t"
#define I_CHAIN2(i) "v_fma_f32 %0, %0, %16, %17\n\t" "v_fma_f32 %1, %1, %16, %17\n\t"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n\t"        /* on register pairs: see body_pk */
#define I_NOP(i) "s_nop 0\n\t"
#define I_AND(i) "v_and_b32_e32 %" #i ", 0xffffe000, %" #i "\n\t"
#define I_SUB(i) "v_sub_f32_e32 %" #i ", %16, %" #i "\n\t"
#define I_PKMAXH(i) "v_pk_max_f16 %" #i ", %" #i ", %16\n\t"
#define I_PKADDH(i) "v_pk_add_f16 %" #i ", %" #i ", %16\n\t"
#define I_PKFMAH(i) "v_pk_fma_f16 %" #i ", %" #i ", %16, %17\n\t"
#define I_CVT32_16(i) "v_cvt_f32_f16_e32 %" #i ", %" #i "\n\t"
#define I_CVT16_32(i) "v_cvt_f16_f32_e32 %" #i ", %" #i "\n\t"
#define I_PACK(i) "v_pack_b32_f16 %" #i ", %" #i ", %16\n\t"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n\t"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 16, %16\n\t"
#define I_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %16, %17\n\t"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n\t"
#define I_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n\t"
#define I_MIXF32(i) "v_fma_mix_f32 %" #i ", %" #i ", %16, %17 op_sel_hi:[1,0,0]\n\t"
#define I_MIXHI(i) "v_fma_mixhi_f16 %" #i ", %" #i ", %16, %17 op_sel_hi:[1,0,0]\n\t"
#define I_ADDU(i) "v_add_u32_e32 %" #i ", %16, %" #i "\n\t"
#define I_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n\t"
#define I_XOR(i) "v_xor_b32_e32 %" #i ", %16, %" #i "\n\t"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n\t"
#define I_PKBF16(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %16\n\t"
#define I_RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
#define I_DOT2C(i) "v_dot2c_f32_f16_e32 %" #i ", %16, %17\n\t"
#define I_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %16\n\t"
#define I_CMP(i) "v_cmp_gt_f32_e32 vcc, %16, %" #i "\n\t"
#define I_CNDVCC(i) "v_cndmask_b32_e32 %" #i ", %16, %" #i ", vcc\n\t"
#define I_MINF(i) "v_min_f32_e32 %" #i ", %16, %" #i "\n\t"
#define I_MAXU(i) "v_max_u32_e32 %" #i ", %16, %" #i "\n\t"
#define I_FMAF16(i) "v_fma_f16 %" #i ", %" #i ", %16, %17\n\t"
#define I_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %16, %17 bitop3:0x80\n\t"
#define I_MAXIMUM3(i) "v_maximum3_f32 %" #i ", %" #i ", %16, %17\n\t"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n\t"
#define I_SIN(i) "v_sin_f32_e32 %" #i ", %" #i "\n\t"
#define I_LOG(i) "v_log_f32_e32 %" #i ", %" #i "\n\t"
#define I_SQRT(i) "v_sqrt_f32_e32 %" #i ", %" #i "\n\t"
#define I_CVTU(i) "v_cvt_f32_u32_e32 %" #i ", %" #i "\n\t"
#define I_LSHR(i) "v_lshrrev_b32_e32 %" #i ", 9, %" #i "\n\t"
#define I_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 7\n\t"
#define I_MULU24(i) "v_mul_u32_u24_e32 %" #i ", %16, %" #i "\n\t"
#define I_ACCW(i) "v_accvgpr_write_b32 a" #i ", %" #i "\n\t"
#define I_DSREAD(i) "ds_read_b128 %" #i ", %8\n\t"

#define OUT16 "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), \
              "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
#define B12(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I) REP16(I)
#define B96(I) B12(I) B12(I) B12(I) B12(I) B12(I) B12(I) B12(I) B12(I)

enum Kind { K_FMA, K_MUL32, K_FMAC32, K_LIT, K_FMAAK, K_MIXLO, K_PKRTZ, K_MAXI, K_MAXF, K_EXP, K_RSQ, K_CND, K_MOV, K_PERM16, K_CHAIN, K_CHAIN2,
            K_NOP, K_AND, K_SUB, K_PKMAXH, K_PKADDH, K_PKFMAH, K_CVT32_16, K_CVT16_32, K_PACK, K_PERM, K_LSHLOR, K_BFI, K_MED3, K_MAX3, K_MIXF32, K_MIXHI,
            K_ADDU, K_LSHL, K_XOR, K_MULLO, K_PKBF16, K_RCP, K_DOT2C, K_LDEXP, K_CMP, K_CNDVCC, K_MINF, K_MAXU, K_FMAF16, K_BITOP3, K_MAXIMUM3, K_MULHI, K_SIN, K_LOG, K_SQRT, K_CVTU, K_LSHR, K_ALIGNBIT, K_MULU24, K_MAD64,
            K_LSHLADD64, K_PKMULF32, K_PKFMAF32, K_PKADDF32, K_COUNT };
static const char* KIND_NAME[] = {"v_fma_f32 (VOP3, 8 B)", "v_mul_f32_e32 (VOP2, 4 B)", "v_fmac_f32_e32 (4 B, 16 chains)", "v_add_f32 + literal (8 B)",
                                  "v_fmaak_f32 (literal, 8 B)", "v_fma_mixlo_f16", "v_cvt_pkrtz_f16_f32", "v_max_i32_e32", "v_max_f32_e32",
                                  "v_exp_f32", "v_rsq_f32", "v_cndmask_b32_e64 (SGPR mask)", "v_mov_b32", "v_permlane16_swap", "v_fma_f32, ONE chain",
                                  "v_fma_f32, TWO chains", "s_nop 0", "v_and_b32 + literal", "v_sub_f32_e32", "v_pk_max_f16", "v_pk_add_f16", "v_pk_fma_f16",
                                  "v_cvt_f32_f16", "v_cvt_f16_f32", "v_pack_b32_f16", "v_perm_b32", "v_lshl_or_b32", "v_bfi_b32", "v_med3_f32", "v_max3_f32",
                                  "v_fma_mix_f32", "v_fma_mixhi_f16", "v_add_u32", "v_lshlrev_b32", "v_xor_b32", "v_mul_lo_u32", "v_cvt_pk_bf16_f32", "v_rcp_f32",
                                  "v_dot2c_f32_f16", "v_ldexp_f32", "v_cmp_gt_f32 (vcc)", "v_cndmask_b32_e32 (vcc)", "v_min_f32", "v_max_u32", "v_fma_f16",
                                  "v_bitop3_b32", "v_maximum3_f32", "v_mul_hi_u32", "v_sin_f32", "v_log_f32", "v_sqrt_f32", "v_cvt_f32_u32", "v_lshrrev_b32",
                                  "v_alignbit_b32", "v_mul_u32_u24", "v_mad_u64_u32 (8 pairs)", "v_lshl_add_u64 (8 pairs)", "v_pk_mul_f32 (per 2 values)", "v_pk_fma_f32 (per 2 values)", "v_pk_add_f32 (per 2 values)"};

template <int KIND, bool BIG>
__device__ __forceinline__ void body(float (&x)[16], float m, float c) {
#define EMIT(I) do { if (BIG) asm volatile(B96(I) : OUT16 : "v"(m), "v"(c) : "s2", "s3"); else asm volatile(B12(I) : OUT16 : "v"(m), "v"(c) : "s2", "s3"); } while (0)
  if constexpr (KIND == K_FMA) EMIT(I_FMA);
  else if constexpr (KIND == K_MUL32) EMIT(I_MUL32);
  else if constexpr (KIND == K_FMAC32) EMIT(I_FMAC32);
  else if constexpr (KIND == K_LIT) EMIT(I_LIT);
  else if constexpr (KIND == K_FMAAK) EMIT(I_FMAAK);
  else if constexpr (KIND == K_MIXLO) EMIT(I_MIXLO);
  else if constexpr (KIND == K_PKRTZ) EMIT(I_PKRTZ);
  else if constexpr (KIND == K_MAXI) EMIT(I_MAXI);
  else if constexpr (KIND == K_MAXF) EMIT(I_MAXF);
  else if constexpr (KIND == K_EXP) EMIT(I_EXP);
  else if constexpr (KIND == K_RSQ) EMIT(I_RSQ);
  else if constexpr (KIND == K_CND) EMIT(I_CND);
  else if constexpr (KIND == K_MOV) EMIT(I_MOV);
  else if constexpr (KIND == K_PERM16) EMIT(I_PERM16);
  else if constexpr (KIND == K_CHAIN) EMIT(I_CHAIN);
  else if constexpr (KIND == K_CHAIN2) { if (BIG) asm volatile(B96(I_CHAIN2) : OUT16 : "v"(m), "v"(c)); else asm volatile(B12(I_CHAIN2) : OUT16 : "v"(m), "v"(c)); }
  else if constexpr (KIND == K_NOP) EMIT(I_NOP);
  else if constexpr (KIND == K_AND) EMIT(I_AND);
  else if constexpr (KIND == K_SUB) EMIT(I_SUB);
  else if constexpr (KIND == K_PKMAXH) EMIT(I_PKMAXH);
  else if constexpr (KIND == K_PKADDH) EMIT(I_PKADDH);
  else if constexpr (KIND == K_PKFMAH) EMIT(I_PKFMAH);
  else if constexpr (KIND == K_CVT32_16) EMIT(I_CVT32_16);
  else if constexpr (KIND == K_CVT16_32) EMIT(I_CVT16_32);
  else if constexpr (KIND == K_PACK) EMIT(I_PACK);
  else if constexpr (KIND == K_PERM) EMIT(I_PERM);
  else if constexpr (KIND == K_LSHLOR) EMIT(I_LSHLOR);
  else if constexpr (KIND == K_BFI) EMIT(I_BFI);
  else if constexpr (KIND == K_MED3) EMIT(I_MED3);
  else if constexpr (KIND == K_MAX3) EMIT(I_MAX3);
  else if constexpr (KIND == K_MIXF32) EMIT(I_MIXF32);
  else if constexpr (KIND == K_MIXHI) EMIT(I_MIXHI);
  else if constexpr (KIND == K_ADDU) EMIT(I_ADDU);
  else if constexpr (KIND == K_LSHL) EMIT(I_LSHL);
  else if constexpr (KIND == K_XOR) EMIT(I_XOR);
  else if constexpr (KIND == K_MULLO) EMIT(I_MULLO);
  else if constexpr (KIND == K_PKBF16) EMIT(I_PKBF16);
  else if constexpr (KIND == K_RCP) EMIT(I_RCP);
  else if constexpr (KIND == K_DOT2C) EMIT(I_DOT2C);
  else if constexpr (KIND == K_LDEXP) EMIT(I_LDEXP);
  else if constexpr (KIND == K_CMP) { if (BIG) asm volatile(B96(I_CMP) : OUT16 : "v"(m), "v"(c) : "vcc"); else asm volatile(B12(I_CMP) : OUT16 : "v"(m), "v"(c) : "vcc"); }
  else if constexpr (KIND == K_CNDVCC) EMIT(I_CNDVCC);
  else if constexpr (KIND == K_MINF) EMIT(I_MINF);
  else if constexpr (KIND == K_MAXU) EMIT(I_MAXU);
  else if constexpr (KIND == K_FMAF16) EMIT(I_FMAF16);
  else if constexpr (KIND == K_BITOP3) EMIT(I_BITOP3);
  else if constexpr (KIND == K_MAXIMUM3) EMIT(I_MAXIMUM3);
  else if constexpr (KIND == K_MULHI) EMIT(I_MULHI);
  else if constexpr (KIND == K_SIN) EMIT(I_SIN);
  else if constexpr (KIND == K_LOG) EMIT(I_LOG);
  else if constexpr (KIND == K_SQRT) EMIT(I_SQRT);
  else if constexpr (KIND == K_CVTU) EMIT(I_CVTU);
  else if constexpr (KIND == K_LSHR) EMIT(I_LSHR);
  else if constexpr (KIND == K_ALIGNBIT) EMIT(I_ALIGNBIT);
  else if constexpr (KIND == K_MULU24) EMIT(I_MULU24);
  else if constexpr (KIND == K_MAD64 || KIND == K_LSHLADD64) {
    // 64-bit results on 8 register pairs: 96 (768) instructions per call
    typedef unsigned long long u64t;
    u64t y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (u64t(__float_as_uint(x[2 * i])) << 32) | __float_as_uint(x[2 * i + 1]);
    const unsigned mu = __float_as_uint(m);
#define Q_MAD(i) "v_mad_u64_u32 %" #i ", s[2:3], %8, %8, %" #i "\n\t"
#define Q_LSA(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %" #i "\n\t"
#define QR8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define Q12(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I) QR8(I)
#define Q96(I) Q12(I) Q12(I) Q12(I) Q12(I) Q12(I) Q12(I) Q12(I) Q12(I)
#define OUTQ "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
    if constexpr (KIND == K_MAD64) { if (BIG) asm volatile(Q96(Q_MAD) : OUTQ : "v"(mu) : "s2", "s3"); else asm volatile(Q12(Q_MAD) : OUTQ : "v"(mu) : "s2", "s3"); }
    else { if (BIG) asm volatile(Q96(Q_LSA) : OUTQ : "v"(mu)); else asm volatile(Q12(Q_LSA) : OUTQ : "v"(mu)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[2 * i] = __uint_as_float(unsigned(y[i] >> 32)); x[2 * i + 1] = __uint_as_float(unsigned(y[i])); }
  }
  else if constexpr (KIND == K_PKMULF32 || KIND == K_PKFMAF32 || KIND == K_PKADDF32) {
    // packed fp32 on 8 register pairs: 96 (768) instructions = 192 (1 536) values per call, so the columns read "per two values"
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y[8], mm = f2{m, m}, cc = f2{c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = f2{x[2 * i], x[2 * i + 1]};
#define P_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n\t"
#define P_FMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define P_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n\t"
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define P12(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)
#define P96(I) P12(I) P12(I) P12(I) P12(I) P12(I) P12(I) P12(I) P12(I)
#define OUT8 "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
#define EMITP(I) do { if (BIG) asm volatile(P96(I) : OUT8 : "v"(mm), "v"(cc)); else asm volatile(P12(I) : OUT8 : "v"(mm), "v"(cc)); } while (0)
    if constexpr (KIND == K_PKMULF32) EMITP(P_MUL);
    else if constexpr (KIND == K_PKFMAF32) EMITP(P_FMA);
    else EMITP(P_ADD);
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[2 * i] = y[i][0]; x[2 * i + 1] = y[i][1]; }
  }
}

#define M8 \
  "v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\t"  "v_mfma_f32_16x16x32_f16 %1, %8, %10, %1\n\t" \
  "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"  "v_mfma_f32_16x16x32_f16 %3, %8, %10, %3\n\t" \
  "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\t"  "v_mfma_f32_16x16x32_f16 %5, %8, %10, %5\n\t" \
  "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\t"  "v_mfma_f32_16x16x32_f16 %7, %8, %10, %7\n\t"
__device__ __forceinline__ void phase_m(f4 (&acc)[8], const h8& a, const h8& b0, const h8& b1) {      // 48 matrix instructions
  asm volatile(M8 M8 M8 M8 M8 M8
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
               : "v"(a), "v"(b0), "v"(b1));
}

// ROLE 0: every wave runs the vector body.  ROLE 1: waves 0-3 run 48 matrix instructions per iteration, waves 4-7 the vector body
// (192 instructions); the stamped wave is a vector one.  ROLE 2: every wave alternates the 48 matrix instructions and the vector
// body, all waves in the same order (coexec mode 4).
template <int KIND, bool BIG, int ROLE>
__global__ __launch_bounds__(512) void k_issue(float* out, unsigned long long* cyc, int iters) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 1.0f + 1e-3f * float((threadIdx.x + i) & 31);
  const float m = 0.999f + 1e-6f * float(threadIdx.x & 3), c = 1e-3f;
  f4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a, b0, b1;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = _Float16(0.01f * float((threadIdx.x + i) & 7)); b0[i] = _Float16(0.02f); b1[i] = _Float16(0.03f); }
  asm volatile("s_mov_b64 s[2:3], -1" ::: "s2", "s3");
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (ROLE == 1 && wave < 4) {
    for (int it = 0; it < iters; ++it) phase_m(acc, a, b0, b1);
  } else if (ROLE == 2) {
    for (int it = 0; it < iters; ++it) {
      phase_m(acc, a, b0, b1);
      body<KIND, BIG>(x, m, c);
    }
  } else {
    for (int it = 0; it < iters; ++it) body<KIND, BIG>(x, m, c);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, bool BIG, int ROLE>
static double run(int threads, int iters, float* out, unsigned long long* cyc, int nblocks, int stamped_wave) {
  hipMemset(cyc, 0, nblocks * 8 * sizeof(unsigned long long));
  k_issue<KIND, BIG, ROLE><<<nblocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  k_issue<KIND, BIG, ROLE><<<nblocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < nblocks; ++b) v.push_back(double(h[b * 8 + stamped_wave]));
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int KIND>
static void kind_rows(float* out, unsigned long long* cyc, int nblocks) {
  const int it_s = 400, it_b = 50;                       // 192 x 400 = 1536 x 50 instructions per wave
  const double n = ((KIND == K_PKMULF32 || KIND == K_PKFMAF32 || KIND == K_PKADDF32 || KIND == K_MAD64 || KIND == K_LSHLADD64) ? 96.0 : (KIND == K_CHAIN2 ? 384.0 : 192.0)) * it_s;
  const double one_s = run<KIND, false, 0>(256, it_s, out, cyc, nblocks, 0), two_s = run<KIND, false, 0>(512, it_s, out, cyc, nblocks, 5);
  const double one_b = run<KIND, true, 0>(256, it_b, out, cyc, nblocks, 0), two_b = run<KIND, true, 0>(512, it_b, out, cyc, nblocks, 5);
  const double beside = run<KIND, false, 1>(512, it_s, out, cyc, nblocks, 5);
  const double alt = run<KIND, false, 2>(512, it_s, out, cyc, nblocks, 5);
  // per instruction and SIMD: one wave = its own cycles; two waves = the pair retires 2 n instructions in the stamped wave's time
  printf("%-34s | 1 wave %5.2f | 2 waves %5.2f per instr & SIMD | 12 KB body: %5.2f / %5.2f | V wave beside an M wave (48 mfma per 192): %5.2f per V instr "
         "| both alternate 48 mfma / 192 V: %6.0f cycles per iteration and wave pair\n",
         KIND_NAME[KIND], one_s / n, two_s / (2 * n), one_b / n, two_b / (2 * n), beside / n, alt / it_s);
  fflush(stdout);
}

template <int K>
static void all_kinds(float* out, unsigned long long* cyc, int nblocks) {
  if constexpr (K < K_COUNT) {
    kind_rows<K>(out, cyc, nblocks);
    all_kinds<K + 1>(out, cyc, nblocks);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int nblocks = p.multiProcessorCount;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, size_t(nblocks) * 512 * sizeof(float));
  hipMalloc(&cyc, size_t(nblocks) * 8 * sizeof(unsigned long long));
  printf("# %s, %d CUs; one 512-thread (or 256-thread) workgroup per CU; s_memtime cycles of one stamped wave, median over workgroups\n", p.name, nblocks);
  printf("# reference: 48 mfma 16x16x32 alone = 768 cycles per wave; 2 waves x 48 = 1536 per pair\n");
  all_kinds<0>(out, cyc, nblocks);
  return 0;
}
