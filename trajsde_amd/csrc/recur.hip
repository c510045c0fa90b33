// recur.hip -- the encoder's latent SDE + GRU recurrence (ENC:128-182): 21 iterations of
//   h' = h + f(t0,h)*dt + g(t0,h,nus_mask) * dW          one Euler-Maruyama step (SDEINT:477-485, App. D)
//   h  = GRU_Unit(aa_out[t], h', valid[:,t])             (ODEU:136-152)
// fp32 weights of one iteration are 263 KB, more than the 160 KB of LDS, so an iteration is two launches:
// k_enc_sde_step (f, g_nus, g_argo images: 118 KB) and k_enc_gru_step (GRU image: 149 KB).
// The dual diffusion nets are selected per row; a tile whose 16 rows share the source evaluates one net only.
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "philox.hpp"
#include "range.hpp"
#include "sde_funcs.hpp"
#include "stamps.hpp"
#include "tile.hpp"

TSDE_STAMP_TABLE(recur, 16)     // diagnostic builds: [2p] work of phase p+1, [2p+1] its barrier, [14] top of the iteration (noise, biases)

namespace tsde {
#ifndef TSDE_STAMPS
static unsigned long long* const g_stamps_recur = nullptr;
#endif

__global__ __launch_bounds__(512) void k_enc_sde_step(const float* __restrict__ img_g, const float* __restrict__ h_in,
                                                      const float* __restrict__ hidden0, int Nt, float dt, float sq, float sn,
                                                      float cs, int idx, int noise_step0, NoiseArg na, const uint8_t* __restrict__ nus,
                                                      const int32_t* __restrict__ eos, const int32_t* __restrict__ pick_slot,
                                                      float* __restrict__ h_ode, float* __restrict__ diff_pick) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, EncSdeL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    f4 y[4], f[4], z[4];
    if (h_in != nullptr) load_row(y, h_in, r, L.g);
    else load_vec<4>(y, hidden0, L.g);                               // ENC:78 learned initial state
    range_note(absmax<4>(y), RS_ENC_STATE);
    drift_eval(f, y, lds + EncSdeL::F, sn, cs, L);
    const bool is_nus = nus[r] != 0;
    const unsigned long long m = __ballot(is_nus);
    float gs;
    if (m == ~0ull) gs = diff_eval(y, lds + EncSdeL::GN, sn, cs, L);             // ENC:470-482
    else if (m == 0ull) gs = diff_eval(y, lds + EncSdeL::GA, sn, cs, L);
    else {
      const float a = diff_eval(y, lds + EncSdeL::GN, sn, cs, L);
      const float b = diff_eval(y, lds + EncSdeL::GA, sn, cs, L);
      gs = is_nus ? a : b;
    }
    noise_row(z, na, STREAM_ENCODER, noise_step0 + idx, r, Nt, L.g);
    em_update(y, f, gs, z, dt, sq);
    if (row < Nt) {
      store_row(y, h_ode, row, L.g);
      const int slot = pick_slot[r];
      if (diff_pick != nullptr && slot >= 0 && eos[r] == idx) {                               // ENC:171,190-191 diffusion of the kept step
        f4 gv[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) gv[jt] = f4{gs, gs, gs, gs};
        store_row(gv, diff_pick, slot, L.g);
      }
    }
  }
}

__global__ __launch_bounds__(512) void k_enc_gru_step(const float* __restrict__ img_g, const float* __restrict__ h_ode,
                                                      const float* __restrict__ x_t, int Nt, int N, int t, int TT, int idx,
                                                      const uint8_t* __restrict__ pad, const int32_t* __restrict__ orig,
                                                      const int32_t* __restrict__ eos, float* __restrict__ h_out,
                                                      float* __restrict__ local_out, float* __restrict__ latent_t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img_g, EncGruL::SIZE);
  using G = EncGruL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    f4 h[4], x[4], ur[8];
    load_row(h, h_ode, r, L.g);
    load_row(x, x_t, r, L.g);
    range_note(absmax<4>(h), RS_ENC_STATE);
    range_note(absmax<4>(x), RS_ENC_INPUT);
    load_vec<8>(ur, lds + G::BUR, L.g);
    linear_acc<8, 4>(ur, h, lds + G::WUR_H, L.lane);                  // y_concat = [h, x]
    linear_acc<8, 4>(ur, x, lds + G::WUR_X, L.lane);
    tanh_<8>(ur);
    f4 u1[4] = {ur[0], ur[1], ur[2], ur[3]}, r1[4] = {ur[4], ur[5], ur[6], ur[7]};
    f4 u[4], rg[4], n1[4], nw[4];
    linear<4, 4>(u, u1, lds + G::WU2, lds + G::BU2, L);
    sigmoid_<4>(u);
    linear<4, 4>(rg, r1, lds + G::WR2, lds + G::BR2, L);
    sigmoid_<4>(rg);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) rg[jt] *= h[jt];                   // reset_gate * h_cur
    load_vec<4>(n1, lds + G::BN0, L.g);
    linear_acc<4, 4>(n1, x, lds + G::WN_X, L.lane);                   // combined = [x, r*h]
    linear_acc<4, 4>(n1, rg, lds + G::WN_H, L.lane);
    tanh_<4>(n1);
    linear<4, 4>(nw, n1, lds + G::WN2, lds + G::BN2, L);
    const bool valid = !pad[int64_t(orig[r]) * TT + t];               // maski = actors_mask[:, t]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hn = (1.0f - u[jt][c]) * nw[jt][c] + u[jt][c] * h[jt][c];
        h[jt][c] = valid ? hn : h[jt][c];
      }
    if (row < Nt) {
      store_row(h, h_out, row, L.g);
      if (row < N) {
        if (eos[r] == idx) store_row(h, local_out, row, L.g);         // ENC:187-188 latent_ys[eos_idcs, arange]
        if (latent_t != nullptr) store_row(h, latent_t, row, L.g);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Cooperative persistent recurrence: ALL H iterations in one launch.
//   * a workgroup = 4 waves owns up to COOP_TMAX row tiles; wave w computes output features [16w, 16w+16) of every
//     layer, so its share of the 66 K weights is 256 VGPRs -- the whole drift / dual diffusion / GRU parameter set
//     lives in the register file (512 VGPRs per lane at one wave per SIMD) for all H steps: no LDS image, no
//     re-staging, no weight traffic after the prologue;
//   * activations are exchanged between the four waves through LDS (one 16x64 tile per layer, 7 barriers per step);
//     the hidden state never leaves LDS between steps; elementwise work (tanh, Philox, gates) is split four ways too;
//   * fp16x3 build: the matrices come from the EncCoopL6 image (two fp16 planes per matrix, tile.hpp), a wave's slice is
//     still 16 VGPRs per matrix, and a 64-k product is 6 dependent v_mfma_f32_16x16x32_f16 instead of 16 dependent
//     v_mfma_f32_16x16x4_f32 -- the recurrence is one wave per SIMD, so the length of that chain is what a step costs;
//     an activation tile is split into its two fp16 pieces once per wave after it is read from LDS (Opnd);
//   * bf16x6 build: plain fp32 fragments of the stage blob (three planes would not fit the register file).
// (COOP_TMAX, COOP_RS, COOP_TILE and StepTab are declared in kernels.hpp)

// A workgroup barrier that orders LDS traffic ONLY.  `__syncthreads()` also drains every outstanding global load and store of the wave
// (s_waitcnt vmcnt(0)) -- at each barrier: a saved tile requested an iteration ahead would be waited for at the very next barrier, and
// every stored slab row at the one after its store.  The two decoder kernels at the end of this file exchange data through the LDS alone (what a wave
// reads from global memory was written by earlier launches or by itself), so their barriers wait for the LDS counter and nothing else.  (Tried in the two recurrence kernels as well: no change -- their loads are
// requested a phase ahead and have arrived by the phase's barrier.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#if TSDE_SPLIT_H3
struct WSlice {   // one wave's 16 output rows of a 64x64 matrix: [plane][k-step] fp16x8 A fragments
  u4 p[2][2];
};
__device__ __forceinline__ WSlice load_slice(const float* mat6, int jo, int lane) {
  WSlice s;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) s.p[pl][ks] = *reinterpret_cast<const u4*>(mat6 + ((jo * 2 + ks) * 2 + pl) * 256 + lane * 4);
  return s;
}
struct Opnd {     // a 16x64 activation tile as B operands: [k-step] fp16x8 high and low pieces
  u4 hi[2], lo[2];
};
__device__ __forceinline__ Opnd make_opnd(const f4 (&in)[4]) {
  Opnd o;
  split_kstep(in[0], in[1], o.hi[0], o.lo[0]);
  split_kstep(in[2], in[3], o.hi[1], o.lo[1]);
  return o;
}
__device__ __forceinline__ void slice_mma(f4& acc, const WSlice& w, const Opnd& x) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const h8 a1 = __builtin_bit_cast(h8, w.p[0][ks]), a2 = __builtin_bit_cast(h8, w.p[1][ks]);
    const h8 x1 = __builtin_bit_cast(h8, x.hi[ks]), x2 = __builtin_bit_cast(h8, x.lo[ks]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, x1, acc, 0, 0, 0);
  }
}
// the same product for the T tiles of a workgroup at once, the tiles' chains interleaved instruction by instruction: a chain on ONE
// accumulator issues a matrix instruction every ~27 cycles, T independent chains one every ~16 (tools/microbench/coexec.hip).  Per
// accumulator the order of the products is that of slice_mma: same bits.
template <int T>
__device__ __forceinline__ void slice_mma_n(f4 (&acc)[T], const WSlice& w, const Opnd (&x)[T]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const h8 a1 = __builtin_bit_cast(h8, w.p[0][ks]), a2 = __builtin_bit_cast(h8, w.p[1][ks]);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, x[t].hi[ks]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(h8, x[t].lo[ks]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(h8, x[t].hi[ks]), acc[t], 0, 0, 0);
  }
}
#else
struct WSlice {   // one wave's 16 output rows of a 64x64 matrix: 4 k-chunks of A fragments
  f4 q[4];
};
__device__ __forceinline__ WSlice load_slice(const float* mat_frag, int jo, int lane) {
  WSlice s;
#pragma unroll
  for (int q = 0; q < 4; ++q) s.q[q] = *reinterpret_cast<const f4*>(mat_frag + ((jo * 4 + q) * 64 + lane) * 4);
  return s;
}
struct Opnd {
  f4 q[4];
};
__device__ __forceinline__ Opnd make_opnd(const f4 (&in)[4]) { return Opnd{{in[0], in[1], in[2], in[3]}}; }
__device__ __forceinline__ void slice_mma(f4& acc, const WSlice& w, const Opnd& x) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.q[q][c], x.q[q][c], acc, 0, 0, 0);
}
// (the T tiles' chains interleaved, as in the fp16x3 build; per accumulator the order of slice_mma)
template <int T>
__device__ __forceinline__ void slice_mma_n(f4 (&acc)[T], const WSlice& w, const Opnd (&x)[T]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.q[q][c], x[t].q[q][c], acc[t], 0, 0, 0);
}
#endif
// full 16x64 activation tile from LDS as B operands / one wave's 16-feature slice to LDS
__device__ __forceinline__ void lds_read_tile(f4 (&in)[4], const float* tile, const Lane& L) {
#pragma unroll
  for (int q = 0; q < 4; ++q) in[q] = *reinterpret_cast<const f4*>(tile + L.n * COOP_RS + 16 * q + 4 * L.g);
}
__device__ __forceinline__ void lds_write_slice(float* tile, const f4& v, int w, const Lane& L) {
  *reinterpret_cast<f4*>(tile + L.n * COOP_RS + 16 * w + 4 * L.g) = v;
}
// Operand tiles: activations that only ever feed a matrix product.  fp16x3 build: the PRODUCING wave splits its 4 values per
// lane into the two fp16 pieces and stores them in operand order -- [piece][k-step][lane][8 halves], 4 KB per tile -- so a
// consumer's B fragments are four conflict-free 16-byte reads and nobody re-splits the tile (every one of the four waves used
// to split all 16 values per lane of every tile it read: ~40 % of the kernel's vector instructions).  A wave's output features
// 16w + 4g + c are the halves 4(w&1) .. +3 of k-step w>>1 in the SAME lane's fragment, so producer and consumer lanes coincide.
#if TSDE_SPLIT_H3
constexpr int COOP_OT = 1024;                                 // floats per operand tile
__device__ __forceinline__ void opnd_write(float* tile, const f4& v, int w, const Lane& L) {
  unsigned h0, l0, h1, l1;
  split_pair(v[0], v[1], h0, l0);
  split_pair(v[2], v[3], h1, l1);
  uint2* base = reinterpret_cast<uint2*>(tile);
  const int ks = w >> 1, half = w & 1;
  // [piece][k-step][half][lane] in 8-byte units: a wave's 64 stores are 512 contiguous bytes.  (Round 3 kept the two halves of a lane's
  // fragment side by side -- one 16-byte read for the consumer, but lanes 16 bytes apart storing 8: every store a 2-way bank conflict, the
  // only LDS conflicts of the forward)  The consumer's two 8-byte reads are one ds_read2_b64.
  base[((0 * 2 + ks) * 2 + half) * 64 + L.lane] = uint2{h0, h1};
  base[((1 * 2 + ks) * 2 + half) * 64 + L.lane] = uint2{l0, l1};
}
__device__ __forceinline__ Opnd opnd_read(const float* tile, const Lane& L) {
  const uint2* b = reinterpret_cast<const uint2*>(tile);
  Opnd o;
  auto frag = [&](int pk) {
    const uint2 x = b[(pk * 2 + 0) * 64 + L.lane], y = b[(pk * 2 + 1) * 64 + L.lane];
    return u4{x.x, x.y, y.x, y.y};
  };
  o.hi[0] = frag(0);
  o.hi[1] = frag(1);
  o.lo[0] = frag(2);
  o.lo[1] = frag(3);
  return o;
}
#else
constexpr int COOP_OT = COOP_TILE;
__device__ __forceinline__ void opnd_write(float* tile, const f4& v, int w, const Lane& L) { lds_write_slice(tile, v, w, L); }
__device__ __forceinline__ Opnd opnd_read(const float* tile, const Lane& L) {
  f4 t[4];
  lds_read_tile(t, tile, L);
  return make_opnd(t);
}
#endif
__device__ __forceinline__ f4 vec_slice(const float* v, int w, int g) { return *reinterpret_cast<const f4*>(v + 16 * w + 4 * g); }
__device__ __forceinline__ f4 tanh4(f4 a) { return f4{fast_tanh(a[0]), fast_tanh(a[1]), fast_tanh(a[2]), fast_tanh(a[3])}; }
__device__ __forceinline__ f4 sigm4(f4 a) { return f4{fast_sigmoid(a[0]), fast_sigmoid(a[1]), fast_sigmoid(a[2]), fast_sigmoid(a[3])}; }
// the cooperative kernel's activations: its layers arrive pre-multiplied by the activation's constant (pack.hip EncCoopL6) in the
// fp16x3 build; in the bf16x6 build, where it reads the plain fp32 fragments, these are the ordinary forms
#if TSDE_SPLIT_H3
__device__ __forceinline__ f4 tanhp4(f4 a) { return f4{tanh_prescaled(a[0]), tanh_prescaled(a[1]), tanh_prescaled(a[2]), tanh_prescaled(a[3])}; }
__device__ __forceinline__ f4 sigmp4(f4 a) { return f4{sigmoid_prescaled(a[0]), sigmoid_prescaled(a[1]), sigmoid_prescaled(a[2]), sigmoid_prescaled(a[3])}; }
#else
__device__ __forceinline__ f4 tanhp4(f4 a) { return tanh4(a); }
__device__ __forceinline__ f4 sigmp4(f4 a) { return sigm4(a); }
#endif

// Register files (round 4).  The sixteen weight slices are 256 registers per lane -- all of the wave's ACCUMULATION registers
// (a0 .. a255) and nothing else: they are written there once (pin_agpr) and the matrix instructions read them in place as their A
// operand.  That needs the accumulators of the products in ordinary registers, which this file asks of the compiler
// (build.py PER_FILE_FLAGS: -mllvm -amdgpu-mfma-vgpr-form).  Before, the compiler kept the slices in ordinary registers, spilled
// what did not fit into a4 .. a255, copied every spilled slice back (v_accvgpr_read) in front of each use -- 120 copies per tile
// and iteration -- and ran EVERY product of every tile through the one accumulator a[0:3], with four v_accvgpr_write before and
// four v_accvgpr_read behind each: the tiles of a workgroup could not overlap at all (0.083 / 0.144 / 0.212 / 0.267 ms for 1 / 2 /
// 3 / 4 tiles per workgroup: linear).
__device__ __forceinline__ unsigned pin_agpr1(unsigned v) {
  unsigned r;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
  return r;
}
#if TSDE_SPLIT_H3
__device__ __forceinline__ WSlice pin_agpr(const WSlice& s) {
  WSlice r;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      r.p[pl][ks] = u4{pin_agpr1(s.p[pl][ks][0]), pin_agpr1(s.p[pl][ks][1]), pin_agpr1(s.p[pl][ks][2]), pin_agpr1(s.p[pl][ks][3])};
  return r;
}
#else
__device__ __forceinline__ WSlice pin_agpr(const WSlice& s) { return s; }
#endif

enum : int { SRC_NUS = 0, SRC_ARGO = 1, SRC_MIXED = 2 };     // diffusion net(s) a workgroup's tiles need (ENC:470-482)

template <int TW, bool SAVE>
__global__ __launch_bounds__(256) void k_enc_recur_coop(const float* __restrict__ sde_img, const float* __restrict__ gru_img,
                                                        const float* __restrict__ coop6, const float* __restrict__ h0, const float* __restrict__ aa_out,
                                                        int Nt, int N, int H, int TT, int tiles_per_wg, StepTab tab, int noise_step0,
                                                        NoiseArg na, const uint8_t* __restrict__ nus,
                                                        const uint8_t* __restrict__ pad, const int32_t* __restrict__ orig,
                                                        const int32_t* __restrict__ eos, const int32_t* __restrict__ pick_slot,
                                                        float* __restrict__ kept, float* __restrict__ diff_pick,
                                                        float* __restrict__ latent_ys, int aa_bf16, RecurTape tp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Lane L;
  const int w = threadIdx.x >> 6;
  // tiles of this workgroup: blockIdx.x * TW + k, k < TW -- consecutive rows, so a workgroup's tiles are rows of one scene (one
  // source, one diffusion net) except where a scene boundary or the fake agents' rows fall inside.  TW is a compile-time constant so
  // that the per-phase code of the TW tiles is straight-line and the scheduler interleaves their LDS reads, splits and matrix chains;
  // tiles past the end of the rows are computed on clamped rows and never stored (inb)
  constexpr int T = TW;
  (void)tiles_per_wg;
  constexpr int PER = COOP_TILE + 5 * COOP_OT;                  // per tile: the fp32 state + five operand tiles
  auto Yb = [&](int k) { return lds + k * PER; };                                   // hidden state, fp32 (the update reads its own slice)
  auto Ys = [&](int k) { return lds + k * PER + COOP_TILE; };                       // the same state as a matrix operand
  auto Ab = [&](int k) { return lds + k * PER + COOP_TILE + 1 * COOP_OT; };
  auto Bb = [&](int k) { return lds + k * PER + COOP_TILE + 2 * COOP_OT; };
  auto Cb = [&](int k) { return lds + k * PER + COOP_TILE + 3 * COOP_OT; };
  auto Xb = [&](int k) { return lds + k * PER + COOP_TILE + 4 * COOP_OT; };         // x_t tile of the step, each wave brings a quarter
  float* GP = lds + T * PER;                                    // [tile][wave][16]

  // ---- register-resident weights: this wave's slice of every matrix
  const float* F = sde_img + EncSdeL::F;
  const float* GN = sde_img + EncSdeL::GN;
  const float* GA = sde_img + EncSdeL::GA;
  using G = EncGruL;
#if TSDE_SPLIT_H3
  using C6 = EncCoopL6;
  auto sl = [&](int m) { return pin_agpr(load_slice(coop6 + m * MAT64X6, w, L.lane)); };
  const WSlice wf0 = sl(C6::F0), wf2 = sl(C6::F2), wf4 = sl(C6::F4), wn0 = sl(C6::N0), wn2 = sl(C6::N2), wa0 = sl(C6::A0), wa2 = sl(C6::A2);
  const WSlice wuh = sl(C6::UH), wrh = sl(C6::RH), wux = sl(C6::UX), wrx = sl(C6::RX), wu2 = sl(C6::U2), wr2 = sl(C6::R2);
  const WSlice wnx = sl(C6::NX), wnh = sl(C6::NH), wn2g = sl(C6::N2G);
#else
  const WSlice wf0 = load_slice(F + DriftL::W0, w, L.lane), wf2 = load_slice(F + DriftL::W2, w, L.lane), wf4 = load_slice(F + DriftL::W4, w, L.lane);
  const WSlice wn0 = load_slice(GN + DiffL::W0, w, L.lane), wn2 = load_slice(GN + DiffL::W2, w, L.lane);
  const WSlice wa0 = load_slice(GA + DiffL::W0, w, L.lane), wa2 = load_slice(GA + DiffL::W2, w, L.lane);
  const WSlice wuh = load_slice(gru_img + G::WUR_H, w, L.lane), wrh = load_slice(gru_img + G::WUR_H, 4 + w, L.lane);
  const WSlice wux = load_slice(gru_img + G::WUR_X, w, L.lane), wrx = load_slice(gru_img + G::WUR_X, 4 + w, L.lane);
  const WSlice wu2 = load_slice(gru_img + G::WU2, w, L.lane), wr2 = load_slice(gru_img + G::WR2, w, L.lane);
  const WSlice wnx = load_slice(gru_img + G::WN_X, w, L.lane), wnh = load_slice(gru_img + G::WN_H, w, L.lane);
  const WSlice wn2g = load_slice(gru_img + G::WN2, w, L.lane);
#endif
  // ---- the bias / head vectors: copied to LDS once, read where they are used (a wave-uniform 16-byte read per 16 lanes; held in
  //      registers for the whole launch they are 84 of the 256 ordinary registers and the three-tile form spills)
  enum : int { V_FB0, V_FWS, V_FWC, V_NB0, V_NWS, V_NWC, V_AB0, V_AWS, V_AWC, V_FB2, V_FB4, V_NB2, V_AB2, V_NW4, V_AW4, V_GBU, V_GBR,
               V_GBU2, V_GBR2, V_GBN0, V_GBN2, V_COUNT };
  static_assert(V_COUNT == COOP_BIAS_VECS, "kernels.hpp coop_lds_floats");
  float* BV = lds + T * PER + T * 64;
  {
    const float* src[V_COUNT] = {F + DriftL::B0, F + DriftL::WS, F + DriftL::WC, GN + DiffL::B0, GN + DiffL::WS, GN + DiffL::WC,
                                 GA + DiffL::B0, GA + DiffL::WS, GA + DiffL::WC, F + DriftL::B2, F + DriftL::B4, GN + DiffL::B2,
                                 GA + DiffL::B2, GN + DiffL::W4, GA + DiffL::W4, gru_img + G::BUR, gru_img + G::BUR + 64,
                                 gru_img + G::BU2, gru_img + G::BR2, gru_img + G::BN0, gru_img + G::BN2};
    // (the EncCoopL6 matrices in front of a tanh / a sigmoid are packed times TANH_PRESCALE / SIGMOID_PRESCALE: their biases follow)
#if TSDE_SPLIT_H3
    constexpr float TP = TANH_PRESCALE, SP = SIGMOID_PRESCALE;
#else
    constexpr float TP = 1.f, SP = 1.f;
#endif
    const float mul[V_COUNT] = {TP, TP, TP, TP, TP, TP, TP, TP, TP, TP, 1.f, TP, TP, 1.f, 1.f, TP, TP, SP, SP, TP, 1.f};
    if (threadIdx.x < 64) {
#pragma unroll
      for (int v = 0; v < V_COUNT; ++v) BV[v * 64 + threadIdx.x] = src[v][threadIdx.x] * mul[v];
    }
  }
  auto bias = [&](int v) { return *reinterpret_cast<const f4*>(BV + v * 64 + 16 * w + 4 * L.g); };
  const float n_b4 = GN[DiffL::B4], a_b4 = GA[DiffL::B4];
  // the first-layer biases with the (sin t, cos t) columns folded in depend on the step alone: all H x 3 of them are formed here, once
  // (the step loop's top used to read nine vectors and form them: with the step table's scalar loads that was 17 % of a step, round 4's
  // phase clocks)
  float* BT = BV + V_COUNT * 64;                               // [H][3][64]
  if (threadIdx.x < 192) {
    const int net = threadIdx.x >> 6, c = threadIdx.x & 63;
    const float* base = net == 0 ? F + DriftL::B0 : (net == 1 ? GN + DiffL::B0 : GA + DiffL::B0);
    const float* ws_ = net == 0 ? F + DriftL::WS : (net == 1 ? GN + DiffL::WS : GA + DiffL::WS);
    const float* wc_ = net == 0 ? F + DriftL::WC : (net == 1 ? GN + DiffL::WC : GA + DiffL::WC);
#if TSDE_SPLIT_H3
    constexpr float TPB = TANH_PRESCALE;
#else
    constexpr float TPB = 1.f;
#endif
    const float b = base[c] * TPB, s_ = ws_[c] * TPB, c_ = wc_[c] * TPB;
    for (int i = 0; i < H; ++i) BT[(i * 3 + net) * 64 + c] = fmaf(c_, tab.cs[i], fmaf(s_, tab.sn[i], b));
  }
  auto step_bias = [&](int idx, int net) { return *reinterpret_cast<const f4*>(BT + (idx * 3 + net) * 64 + 16 * w + 4 * L.g); };

  // ---- per-tile row bookkeeping and the initial state
  int64_t rowk[COOP_TMAX];
  bool inb[COOP_TMAX], is_nus[COOP_TMAX];
  int eosk[COOP_TMAX], slotk[COOP_TMAX], origk[COOP_TMAX];
  uint32_t ridk[COOP_TMAX];
  bool all_nus = true, all_argo = true;
#pragma unroll
  for (int k = 0; k < COOP_TMAX; ++k) {
    if (k < T) {
      const int64_t row = (int64_t(blockIdx.x) * T + k) * 16 + L.n;
      inb[k] = row < Nt;
      rowk[k] = inb[k] ? row : Nt - 1;
      is_nus[k] = nus[rowk[k]] != 0;
      const unsigned long long m = __ballot(is_nus[k]);
      all_nus = all_nus && m == ~0ull;
      all_argo = all_argo && m == 0ull;
      eosk[k] = eos[rowk[k]];
      slotk[k] = pick_slot[rowk[k]];
      origk[k] = orig[rowk[k]];
      ridk[k] = na.row_ids ? uint32_t(na.row_ids[rowk[k]]) : uint32_t(rowk[k]);
      const f4 y = h0 ? vec_slice(h0, w, L.g) : f4{0.f, 0.f, 0.f, 0.f};      // same initial vector for every row (ENC:78 / :257)
      range_note(absmax4(y), RS_ENC_STATE);
      lds_write_slice(Yb(k), y, w, L);
      opnd_write(Ys(k), y, w, L);
    }
  }
  // which diffusion net(s) the workgroup's rows use: uniform over the workgroup (every wave looks at the same rows), decided once,
  // so the phases below branch once per phase instead of once per tile
  const int src_kind = __builtin_amdgcn_readfirstlane(all_nus ? SRC_NUS : (all_argo ? SRC_ARGO : SRC_MIXED));
  const bool injected = na.z != nullptr;
  const uint64_t nkey = injected ? 0ull : noise_key(na);
  __syncthreads();
  PhaseClock<16> clk;                                      // (diagnostic builds only: stamps.hpp)
  clk.start();
  float m_state = 0.f, m_input = 0.f;                      // fp16 range guard (range.hpp): running maxima, noted once behind the loop

  for (int idx = 0; idx < H; ++idx) {
    const int t = H - 1 - idx;
    const float dt = tab.dt[idx], sq = tab.sq[idx];
    // SAVE (the training forward): this wave's 16 channels of every activation the backward reads go to the tape slabs
    // [H][Nt][64] of csrc/encoder_bwd.hip (what k_enc_recur_save wrote one tile per wave) -- fire-and-forget stores
    auto keep = [&](float* slab, int k, const f4& v) {
      if (SAVE && inb[k]) *reinterpret_cast<f4*>(slab + (int64_t(idx) * Nt + rowk[k]) * 64 + 16 * w + 4 * L.g) = v;
    };
    // first-layer biases with the (sin t, cos t) columns folded in (BT above)
    const f4 bf0 = step_bias(idx, 0), bn0 = step_bias(idx, 1), ba0 = step_bias(idx, 2);
    bool validk[COOP_TMAX];                                  // maski = actors_mask[:, t] (ENC:176): asked for here, used in P7
#pragma unroll
    for (int k = 0; k < COOP_TMAX; ++k)
      if (k < T) validk[k] = !pad[int64_t(origk[k]) * TT + t];
    f4 xq[COOP_TMAX];                                        // this wave's quarter of the x_t rows: loaded here, to LDS in P3
    f4 zq[COOP_TMAX];                                        // the step's normals for this wave's 16 channels (used in P3)
#pragma unroll
    for (int k = 0; k < COOP_TMAX; ++k)
      if (k < T) {                                           // in flight early; fp32 or bf16 storage (tile.hpp)
        const int64_t at = (int64_t(t) * Nt + rowk[k]) * 64 + 16 * w + 4 * L.g;
        xq[k] = aa_bf16 ? widen4(*reinterpret_cast<const bf4*>(reinterpret_cast<const __bf16*>(aa_out) + at))
                        : *reinterpret_cast<const f4*>(aa_out + at);
      }
#define TS_TILES _Pragma("unroll") for (int k = 0; k < T; ++k)
    auto state_slice = [&](int k) { return *reinterpret_cast<const f4*>(Yb(k) + L.n * COOP_RS + 16 * w + 4 * L.g); };
    clk.mark(14);
    // ---- P1: first layers of f and g.  Every phase has the same shape: the tiles' operand reads, then the products with the
    //      tiles' chains interleaved (slice_mma_n), then the pointwise tails and the exchange stores
    {
      Opnd y[T];
      f4 a[T], g[T];
      TS_TILES y[k] = opnd_read(Ys(k), L);
      TS_TILES a[k] = bf0;
      slice_mma_n<T>(a, wf0, y);
      if (src_kind == SRC_NUS) {
        TS_TILES g[k] = bn0;
        slice_mma_n<T>(g, wn0, y);
      } else if (src_kind == SRC_ARGO) {
        TS_TILES g[k] = ba0;
        slice_mma_n<T>(g, wa0, y);
      } else {
        f4 g2[T];
        TS_TILES { g[k] = bn0; g2[k] = ba0; }
        slice_mma_n<T>(g, wn0, y);
        slice_mma_n<T>(g2, wa0, y);
        TS_TILES g[k] = is_nus[k] ? g[k] : g2[k];
      }
      // the step's normals: independent of the state, so they are computed HERE, in the shadow of the matrix instructions just
      // issued (a lone wave per SIMD overlaps nothing else with them), not ahead of the phase
      if (injected) {
        TS_TILES zq[k] = *reinterpret_cast<const f4*>(na.z + (int64_t(noise_step0 + idx) * Nt + rowk[k]) * 64 + 16 * w + 4 * L.g);
      } else {
        TS_TILES zq[k] = philox_normal4(nkey, STREAM_ENCODER, uint32_t(noise_step0 + idx), ridk[k], uint32_t(4 * w + L.g));
      }
      TS_TILES {
        if (SAVE) keep(tp.HIN, k, state_slice(k));
        a[k] = tanhp4(a[k]);
        keep(tp.H1, k, a[k]);
        opnd_write(Ab(k), a[k], w, L);
        g[k] = tanhp4(g[k]);
        keep(tp.G1, k, g[k]);
        opnd_write(Bb(k), g[k], w, L);
      }
    }
    clk.mark(0);
    __syncthreads();
    clk.mark(1);
    // ---- P2: second layers; partial dot of the diffusion head
    {
      Opnd f1[T], g1[T];
      f4 a[T], g[T];
      float part[T];
      auto head = [&](const f4& h2, const f4& wv) { return row_sum(h2[0] * wv[0] + h2[1] * wv[1] + h2[2] * wv[2] + h2[3] * wv[3]); };
      TS_TILES { f1[k] = opnd_read(Ab(k), L); g1[k] = opnd_read(Bb(k), L); }
      TS_TILES a[k] = bias(V_FB2);
      slice_mma_n<T>(a, wf2, f1);
      if (src_kind == SRC_NUS) {
        TS_TILES g[k] = bias(V_NB2);
        slice_mma_n<T>(g, wn2, g1);
        TS_TILES { g[k] = tanhp4(g[k]); part[k] = head(g[k], bias(V_NW4)); }
      } else if (src_kind == SRC_ARGO) {
        TS_TILES g[k] = bias(V_AB2);
        slice_mma_n<T>(g, wa2, g1);
        TS_TILES { g[k] = tanhp4(g[k]); part[k] = head(g[k], bias(V_AW4)); }
      } else {
        f4 g2[T];
        TS_TILES { g[k] = bias(V_NB2); g2[k] = bias(V_AB2); }
        slice_mma_n<T>(g, wn2, g1);
        slice_mma_n<T>(g2, wa2, g1);
        TS_TILES {
          g[k] = tanhp4(g[k]);
          g2[k] = tanhp4(g2[k]);
          const float pn = head(g[k], bias(V_NW4)), pa = head(g2[k], bias(V_AW4));
          part[k] = is_nus[k] ? pn : pa;
          g[k] = is_nus[k] ? g[k] : g2[k];
        }
      }
      TS_TILES {
        a[k] = tanhp4(a[k]);
        keep(tp.H2, k, a[k]);
        opnd_write(Cb(k), a[k], w, L);
        keep(tp.G2, k, g[k]);
        if (L.g == 0) GP[(k * 4 + w) * 16 + L.n] = part[k];
      }
    }
    clk.mark(2);
    __syncthreads();
    clk.mark(3);
    // ---- P3: drift output, diffusion scalar, Euler-Maruyama update of this wave's 16 state channels
    {
      Opnd f2[T];
      f4 f[T];
      TS_TILES f2[k] = opnd_read(Cb(k), L);
      TS_TILES f[k] = bias(V_FB4);
      slice_mma_n<T>(f, wf4, f2);
      float gsk[T];
      TS_TILES {
        const float b4 = is_nus[k] ? n_b4 : a_b4;
        const float gs = fast_sigmoid(GP[(k * 4 + 0) * 16 + L.n] + GP[(k * 4 + 1) * 16 + L.n] + GP[(k * 4 + 2) * 16 + L.n] +
                                      GP[(k * 4 + 3) * 16 + L.n] + b4);
        const f4 z = zq[k];
        f4 y = state_slice(k);
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = (y[c] + f[k][c] * dt) + gs * (z[c] * sq);       // SDEINT:483
        keep(tp.HODE, k, y);
        keep(tp.XS, k, xq[k]);
        if (SAVE && inb[k] && w == 0 && L.g == 0) tp.GS[int64_t(idx) * Nt + rowk[k]] = gs;
        m_state = fmaxf(m_state, absmax4(y));                                             // operands of the GRU's split products
        m_input = fmaxf(m_input, absmax4(xq[k]));
        lds_write_slice(Yb(k), y, w, L);                                                  // Y now holds h' (all P1 reads are behind two barriers)
        opnd_write(Ys(k), y, w, L);
        opnd_write(Xb(k), xq[k], w, L);
        gsk[k] = gs;
      }
      TS_TILES
        if (diff_pick != nullptr && inb[k] && slotk[k] >= 0 && eosk[k] == idx)
          *reinterpret_cast<f4*>(diff_pick + int64_t(slotk[k]) * 64 + 16 * w + 4 * L.g) = f4{gsk[k], gsk[k], gsk[k], gsk[k]};
    }
    clk.mark(4);
    __syncthreads();
    clk.mark(5);
    // ---- P4: GRU gates, first layers (y_concat = [h', x])
    {
      Opnd hp[T], xin[T];
      f4 u1[T], r1[T];
      TS_TILES { hp[k] = opnd_read(Ys(k), L); xin[k] = opnd_read(Xb(k), L); }
      TS_TILES { u1[k] = bias(V_GBU); r1[k] = bias(V_GBR); }
      slice_mma_n<T>(u1, wuh, hp);
      slice_mma_n<T>(r1, wrh, hp);
      slice_mma_n<T>(u1, wux, xin);
      slice_mma_n<T>(r1, wrx, xin);
      TS_TILES {
        u1[k] = tanhp4(u1[k]);
        r1[k] = tanhp4(r1[k]);
        keep(tp.U1, k, u1[k]);
        keep(tp.R1, k, r1[k]);
        opnd_write(Ab(k), u1[k], w, L);
        opnd_write(Bb(k), r1[k], w, L);
      }
    }
    clk.mark(6);
    __syncthreads();
    clk.mark(7);
    // ---- P5: gates; reset * h'
    f4 ug[T];
    {
      Opnd u1[T], r1[T];
      f4 r[T];
      TS_TILES { u1[k] = opnd_read(Ab(k), L); r1[k] = opnd_read(Bb(k), L); }
      TS_TILES { ug[k] = bias(V_GBU2); r[k] = bias(V_GBR2); }
      slice_mma_n<T>(ug, wu2, u1);
      slice_mma_n<T>(r, wr2, r1);
      TS_TILES {
        ug[k] = sigmp4(ug[k]);
        const f4 hs = state_slice(k);
        r[k] = sigmp4(r[k]);
        const f4 rhs = r[k] * hs;
        keep(tp.UU, k, ug[k]);
        keep(tp.RR, k, r[k]);
        keep(tp.RH, k, rhs);
        opnd_write(Cb(k), rhs, w, L);
      }
    }
    clk.mark(8);
    __syncthreads();
    clk.mark(9);
    // ---- P6: candidate state, first layer (combined = [x, r*h'])
    {
      Opnd rh[T], xin[T];
      f4 n1[T];
      TS_TILES { xin[k] = opnd_read(Xb(k), L); rh[k] = opnd_read(Cb(k), L); }       // the x_t tile again (not kept live across the phases)
      TS_TILES n1[k] = bias(V_GBN0);
      slice_mma_n<T>(n1, wnx, xin);
      slice_mma_n<T>(n1, wnh, rh);
      TS_TILES {
        n1[k] = tanhp4(n1[k]);
        keep(tp.N1, k, n1[k]);
        opnd_write(Ab(k), n1[k], w, L);
      }
    }
    clk.mark(10);
    __syncthreads();
    clk.mark(11);
    // ---- P7: candidate state, second layer; gated blend; masked update; picks
    {
      Opnd n1[T];
      f4 nw[T];
      TS_TILES n1[k] = opnd_read(Ab(k), L);
      TS_TILES nw[k] = bias(V_GBN2);
      slice_mma_n<T>(nw, wn2g, n1);
      TS_TILES {
        keep(tp.NW, k, nw[k]);
        f4 hs = state_slice(k);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float hn = (1.0f - ug[k][c]) * nw[k][c] + ug[k][c] * hs[c];
          hs[c] = validk[k] ? hn : hs[c];
        }
        lds_write_slice(Yb(k), hs, w, L);
        opnd_write(Ys(k), hs, w, L);
        nw[k] = hs;
      }
      TS_TILES {                                             // the rare stores behind the tiles' straight-line work
        const int64_t row = rowk[k];
        if (inb[k] && row < N) {
          if (eosk[k] == idx) *reinterpret_cast<f4*>(kept + row * 64 + 16 * w + 4 * L.g) = nw[k];
          if (latent_ys != nullptr) *reinterpret_cast<f4*>(latent_ys + (int64_t(idx) * N + row) * 64 + 16 * w + 4 * L.g) = nw[k];
        }
      }
    }
#undef TS_TILES
    clk.mark(12);
    __syncthreads();
    clk.mark(13);
  }
  range_note(m_state, RS_ENC_STATE);
  range_note(m_input, RS_ENC_INPUT);
  if (!SAVE && w == 0 && L.lane == 0) clk.flush(g_stamps_recur, (unsigned long long)(H) * T);
}

#define TS_COOP_INST(TW, SV)                                                                                                      \
  template __global__ void k_enc_recur_coop<TW, SV>(const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, \
                                                    StepTab, int, NoiseArg, const uint8_t*, const uint8_t*, const int32_t*, const int32_t*,  \
                                                    const int32_t*, float*, float*, float*, int, RecurTape);
TS_COOP_INST(1, false) TS_COOP_INST(2, false) TS_COOP_INST(3, false) TS_COOP_INST(4, false)
TS_COOP_INST(1, true) TS_COOP_INST(2, true) TS_COOP_INST(3, true) TS_COOP_INST(4, true)
#undef TS_COOP_INST

// ------------------------------------------------------------------------------------------------------------------
// The recurrence BACKWARD in the same cooperative form (training; replaces encoder_bwd.hip k_enc_recur_bwd where it applies): the sixteen
// TRANSPOSED matrices (GruBwdL / EncSdeBwdL images) live in the registers of the four waves, wave w produces channels [16w, 16w+16) of
// every delta, the deltas that feed a product travel through fp32 LDS tiles, d h stays in registers from iteration to iteration.
// The one-tile-per-wave kernel re-staged its two 128 KB images every half iteration on 129 workgroups (0.83 ms at 64 x 128 agents).
// Deltas are tiny and fp16 has five exponent bits (why tile.hpp linear_adj scales the rows of every adjoint product).  Here a row is
// normalised ONCE per iteration: everything the iteration computes for a row is linear in the gradient entering it (d h of the later
// iteration, the kept-latent gradient, the DiffBCE gradient of the diffusion value), so the row's largest such magnitude fixes a power of
// two that is applied at the top of the iteration and taken off again at every store and at the bottom; in between the deltas are O(1)
// and travel as ordinary pre-split operand tiles (opnd_write / opnd_read) -- one exchange per phase, no per-product maxima.
// Nine barriers per iteration:  G0 row magnitudes | G1 gate / candidate deltas | G2 new_state.2^T, update.2^T | G3 new_state.0^T |
// G4 reset.2^T | G5 gate first layers^T -> d x_t, d h';  drift.4 input, diffusion dot | S2 drift.4^T, diffusion heads |
// S3 drift.2^T, diffusion.2^T | (S4 first layers^T -> d h of the previous iteration: its readers are fenced by the next G0).
__device__ __forceinline__ f4 zero_mma(const WSlice& w, const Opnd& x) {
  f4 t = f4{0.f, 0.f, 0.f, 0.f};
  slice_mma(t, w, x);
  return t;
}
__device__ __forceinline__ f4 dtanh4(const f4& d, const f4& y) { return f4{d[0] * (1.f - y[0] * y[0]), d[1] * (1.f - y[1] * y[1]), d[2] * (1.f - y[2] * y[2]), d[3] * (1.f - y[3] * y[3])}; }

template <int TW>
__global__ __launch_bounds__(256) void k_enc_recur_bwd_coop(RecurBwdCoopArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Lane L;
  const int w = threadIdx.x >> 6;
  constexpr int T = TW;
  constexpr int PER = COOP_BWD_TILES * COOP_OT + 128;
  auto Tb = [&](int k, int i) { return lds + k * PER + i * COOP_OT; };
  auto GPb = [&](int k) { return lds + k * PER + COOP_BWD_TILES * COOP_OT; };            // [wave][16]: partial diffusion dots
  auto MXb = [&](int k) { return lds + k * PER + COOP_BWD_TILES * COOP_OT + 64; };       // [wave][16]: row magnitudes
  const int Nt = a.Nt, N = a.N, H = a.H;
  using G = GruBwdL;
  using S = EncSdeBwdL;
  auto sl = [&](const float* img, int off) { return pin_agpr(load_slice(img + off, w, L.lane)); };   // (see k_enc_recur_coop: 256 accumulation registers)
  const WSlice wn2 = sl(a.gru_t, G::WN2T), wnx = sl(a.gru_t, G::WNXT), wnh = sl(a.gru_t, G::WNHT), wu2 = sl(a.gru_t, G::WU2T), wr2 = sl(a.gru_t, G::WR2T);
  const WSlice wuh = sl(a.gru_t, G::UHT), wrh = sl(a.gru_t, G::RHT), wux = sl(a.gru_t, G::UXT), wrx = sl(a.gru_t, G::RXT);
  const WSlice wf0 = sl(a.sde_t, S::F_W0T), wf2 = sl(a.sde_t, S::F_W2T), wf4 = sl(a.sde_t, S::F_W4T);
  const WSlice wgn0 = sl(a.sde_t, S::GN_W0T), wgn2 = sl(a.sde_t, S::GN_W2T), wga0 = sl(a.sde_t, S::GA_W0T), wga2 = sl(a.sde_t, S::GA_W2T);
  const f4 w4n = vec_slice(a.sde_t + S::GN_W4, w, L.g), w4a = vec_slice(a.sde_t + S::GA_W4, w, L.g);

  int64_t rowk[COOP_TMAX];
  bool inb[COOP_TMAX], is_nus[COOP_TMAX];
  int eosk[COOP_TMAX], origk[COOP_TMAX];
  unsigned long long nusmask[COOP_TMAX];
  f4 dh[COOP_TMAX];
#pragma unroll
  for (int k = 0; k < COOP_TMAX; ++k)
    if (k < T) {
      const int64_t row = (int64_t(blockIdx.x) + int64_t(k) * gridDim.x) * 16 + L.n;
      inb[k] = row < Nt;
      rowk[k] = inb[k] ? row : Nt - 1;
      is_nus[k] = a.nus[rowk[k]] != 0;
      nusmask[k] = __ballot(is_nus[k]);
      eosk[k] = a.eos[rowk[k]];
      origk[k] = a.orig[rowk[k]];
      dh[k] = f4{0.f, 0.f, 0.f, 0.f};                      // nothing reads the final state except through the kept latents
    }
  const int ch = 16 * w + 4 * L.g;                          // this lane's four channels of its row
  // tape slices are requested ONE PHASE before they are used (a phase is ~1.5 us at one wave per SIMD: a load issued inside it would be
  // its longest instruction) and held for that phase only -- holding an iteration's twelve slices per tile spilled at two tiles already
  auto at_i = [&](const float* slab, int k, int i) { return *reinterpret_cast<const f4*>(slab + (int64_t(i) * Nt + rowk[k]) * 64 + ch); };
  f4 hs[COOP_TMAX], pu[COOP_TMAX], pnw[COOP_TMAX];          // h', u, candidate of the iteration about to run
#pragma unroll
  for (int k = 0; k < COOP_TMAX; ++k)
    if (k < T) {
      hs[k] = at_i(a.tp.HODE, k, H - 1);
      pu[k] = at_i(a.tp.UU, k, H - 1);
      pnw[k] = at_i(a.tp.NW, k, H - 1);
    }

  for (int idx = H - 1; idx >= 0; --idx) {
    const int t = H - 1 - idx;
    const float dt = a.dt[idx], sq = a.sq[idx];
    auto at = [&](const float* slab, int k) { return at_i(slab, k, idx); };
    f4 dho[COOP_TMAX], dx[COOP_TMAX], pa[COOP_TMAX], pb[COOP_TMAX], d0[COOP_TMAX];   // pa / pb: the slices fetched for the next phase
    float pgs[COOP_TMAX], up[COOP_TMAX], down[COOP_TMAX], dldg[COOP_TMAX];
    auto keep = [&](float* slab, int k, const f4& v) {       // stores carry the true scale
      if (inb[k]) *reinterpret_cast<f4*>(slab + (int64_t(idx) * Nt + rowk[k]) * 64 + ch) = v * down[k];
    };
#define TS_EACH_TILE _Pragma("unroll") for (int k = 0; k < COOP_TMAX; ++k) if (k < T)
    // ---- G0: the gradient entering the iteration and its magnitude per row
    TS_EACH_TILE {
      pa[k] = at(a.tp.N1, k);
      pb[k] = at(a.tp.U1, k);
      d0[k] = dh[k];
      if (rowk[k] < N && eosk[k] == idx) d0[k] += *reinterpret_cast<const f4*>(a.dlat + rowk[k] * 64 + ch);      // ENC:187-188
      dldg[k] = (inb[k] && eosk[k] == idx) ? a.DLDG[rowk[k]] : 0.f;               // ENC:171,190-191: joins at the diffusion head
      const float m = fmaxf(row_max(absmax4(d0[k])), fabsf(dldg[k]));
      if (L.g == 0) MXb(k)[16 * w + L.n] = m;
    }
    __syncthreads();
    // ---- G1: through the gated blend (ODEU:147-151), on the normalised row
    TS_EACH_TILE {
      const float* pm = MXb(k) + L.n;
      const unsigned e = __float_as_uint(fmaxf(fmaxf(pm[0], pm[16]), fmaxf(pm[32], pm[48]))) & 0x7F800000u;   // 0 for an all-zero row
      up[k] = __uint_as_float(0x7F000000u - e);
      down[k] = __uint_as_float(e);
      const bool valid = !a.pad[int64_t(origk[k]) * a.TT + t] && inb[k];
      f4 dnw, du;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = d0[k][c] * up[k];
        const float dd = valid ? d : 0.f, uu = pu[k][c];
        dnw[c] = dd * (1.0f - uu);                                                 // d candidate
        du[c] = dd * (hs[k][c] - pnw[k][c]) * uu * (1.0f - uu);                   // d u_pre
        dho[k][c] = valid ? dd * uu : d;                                           // masked rows pass the state through
      }
      keep(a.DNW, k, dnw);
      keep(a.DUP, k, du);
      opnd_write(Tb(k, 0), dnw, w, L);
      opnd_write(Tb(k, 1), du, w, L);
    }
    __syncthreads();
    // ---- G2: new_state_net.2^T, update_gate.2^T
    TS_EACH_TILE {
      const f4 n1 = pa[k], u1 = pb[k];
      pa[k] = at(a.tp.RR, k);
      const f4 t4 = dtanh4(zero_mma(wn2, opnd_read(Tb(k, 0), L)), n1);            // d n1_pre
      const f4 du1 = dtanh4(zero_mma(wu2, opnd_read(Tb(k, 1), L)), u1);           // d u1_pre
      keep(a.DN1P, k, t4);
      keep(a.DU1, k, du1);
      opnd_write(Tb(k, 2), t4, w, L);
      opnd_write(Tb(k, 3), du1, w, L);
    }
    __syncthreads();
    // ---- G3: new_state_net.0^T on [x, r h']
    TS_EACH_TILE {
      const f4 rr = pa[k];
      pa[k] = at(a.tp.R1, k);
      const Opnd o = opnd_read(Tb(k, 2), L);
      dx[k] = zero_mma(wnx, o);
      const f4 drh = zero_mma(wnh, o);
      f4 drp;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float rv = rr[c];
        dho[k][c] = fmaf(drh[c], rv, dho[k][c]);
        drp[c] = drh[c] * hs[k][c] * rv * (1.0f - rv);                             // d r_pre
      }
      keep(a.DRP, k, drp);
      opnd_write(Tb(k, 0), drp, w, L);
    }
    __syncthreads();
    // ---- G4: reset_gate.2^T
    TS_EACH_TILE {
      const f4 dr1 = dtanh4(zero_mma(wr2, opnd_read(Tb(k, 0), L)), pa[k]);
      keep(a.DR1, k, dr1);
      opnd_write(Tb(k, 1), dr1, w, L);
    }
    __syncthreads();
    // ---- G5: gate first layers^T on [h', x] -> d x_t and d h'; then the Euler-Maruyama step (SDEINT:483): d f = dt d h', the
    //      diffusion scalar's gradient is the row dot of d h' with z sqrt(dt)
    TS_EACH_TILE {
      pa[k] = at(a.tp.H2, k);
      pb[k] = at(a.tp.G2, k);
      pgs[k] = a.tp.GS[int64_t(idx) * Nt + rowk[k]];
      const Opnd ou = opnd_read(Tb(k, 3), L), orr = opnd_read(Tb(k, 1), L);
      slice_mma(dho[k], wuh, ou);
      slice_mma(dho[k], wrh, orr);
      slice_mma(dx[k], wux, ou);
      slice_mma(dx[k], wrx, orr);
      if (inb[k]) *reinterpret_cast<f4*>(a.DAA + (int64_t(t) * Nt + rowk[k]) * 64 + ch) = dx[k] * down[k];
      if (!inb[k]) dho[k] = f4{0.f, 0.f, 0.f, 0.f};
      const f4 df = dho[k] * dt;
      keep(a.DF, k, df);
      opnd_write(Tb(k, 0), df, w, L);
      f4 z;
      if (a.na.z != nullptr) z = *reinterpret_cast<const f4*>(a.na.z + (int64_t(idx) * Nt + rowk[k]) * 64 + ch);
      else z = philox_normal4(noise_key(a.na), STREAM_ENCODER, uint32_t(idx),
                              a.na.row_ids ? uint32_t(a.na.row_ids[rowk[k]]) : uint32_t(rowk[k]), uint32_t(4 * w + L.g));
      float cdot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) cdot = fmaf(z[c] * sq, dho[k][c], cdot);
      cdot = row_sum(cdot);
      if (L.g == 0) GPb(k)[16 * w + L.n] = cdot;
    }
    __syncthreads();
    // ---- S2: drift.4^T; the diffusion head (64 -> 1, sigmoid) of the row's own net
    TS_EACH_TILE {
      const f4 h2 = pa[k], g2 = pb[k];
      pa[k] = at(a.tp.H1, k);
      pb[k] = at(a.tp.G1, k);
      const f4 dh2 = dtanh4(zero_mma(wf4, opnd_read(Tb(k, 0), L)), h2);
      keep(a.DH2, k, dh2);
      opnd_write(Tb(k, 2), dh2, w, L);
      const float* gp = GPb(k) + L.n;
      const float dg = ((gp[0] + gp[16]) + (gp[32] + gp[48])) + dldg[k] * up[k];
      const float gs = pgs[k];
      const float dgp = inb[k] ? dg * gs * (1.0f - gs) : 0.f;
      const float seln = is_nus[k] ? dgp : 0.f, sela = is_nus[k] ? 0.f : dgp;
      f4 dg2n, dg2a;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gg = 1.0f - g2[c] * g2[c];
        dg2n[c] = seln * w4n[c] * gg;
        dg2a[c] = sela * w4a[c] * gg;
      }
      keep(a.DG2N, k, dg2n);
      keep(a.DG2A, k, dg2a);
      if (inb[k] && w == 0 && L.g == 0) {
        a.DGPN[int64_t(idx) * Nt + rowk[k]] = seln * down[k];
        a.DGPA[int64_t(idx) * Nt + rowk[k]] = sela * down[k];
      }
      if (nusmask[k] != 0ull) opnd_write(Tb(k, 3), dg2n, w, L);
      if (nusmask[k] != ~0ull) opnd_write(Tb(k, 4), dg2a, w, L);
    }
    __syncthreads();
    // ---- S3: drift.2^T, diffusion.2^T
    TS_EACH_TILE {
      const f4 h1 = pa[k], g1 = pb[k];
      const f4 dh1 = dtanh4(zero_mma(wf2, opnd_read(Tb(k, 2), L)), h1);
      keep(a.DH1, k, dh1);
      opnd_write(Tb(k, 0), dh1, w, L);
      f4 dg1n = f4{0.f, 0.f, 0.f, 0.f}, dg1a = dg1n;
      if (nusmask[k] != 0ull) {
        dg1n = dtanh4(zero_mma(wgn2, opnd_read(Tb(k, 3), L)), g1);
        opnd_write(Tb(k, 1), dg1n, w, L);
      }
      if (nusmask[k] != ~0ull) {
        dg1a = dtanh4(zero_mma(wga2, opnd_read(Tb(k, 4), L)), g1);
        opnd_write(Tb(k, 5), dg1a, w, L);
      }
      keep(a.DG1N, k, dg1n);
      keep(a.DG1A, k, dg1a);
      if (idx > 0) {                                         // the next iteration's blend inputs
        hs[k] = at_i(a.tp.HODE, k, idx - 1);
        pu[k] = at_i(a.tp.UU, k, idx - 1);
        pnw[k] = at_i(a.tp.NW, k, idx - 1);
      }
    }
    __syncthreads();
    // ---- S4: first layers^T (the 64 state columns of the 66-wide inputs) -> d h entering the previous iteration, true scale.
    //      No barrier behind it: the next writes to the tiles it reads (0, 1, 5) come after G0's barrier.
    TS_EACH_TILE {
      f4 dyn = dho[k];
      slice_mma(dyn, wf0, opnd_read(Tb(k, 0), L));
      if (nusmask[k] != 0ull) slice_mma(dyn, wgn0, opnd_read(Tb(k, 1), L));
      if (nusmask[k] != ~0ull) slice_mma(dyn, wga0, opnd_read(Tb(k, 5), L));
      dh[k] = dyn * down[k];
    }
#undef TS_EACH_TILE
  }
#pragma unroll
  for (int k = 0; k < COOP_TMAX; ++k)
    if (k < T)
      if (inb[k]) *reinterpret_cast<f4*>(a.dh_out + rowk[k] * 64 + ch) = dh[k];
}
template __global__ void k_enc_recur_bwd_coop<1>(RecurBwdCoopArgs);
template __global__ void k_enc_recur_bwd_coop<2>(RecurBwdCoopArgs);
template __global__ void k_enc_recur_bwd_coop<3>(RecurBwdCoopArgs);
template __global__ void k_enc_recur_bwd_coop<4>(RecurBwdCoopArgs);

// ------------------------------------------------------------------------------------------------------------------
// The decoder backward's forward REPLAY of the winning paths in the cooperative form (decoder_bwd.hip k_sde_replay is the one-wave
// form).  Only N paths are replayed -- 384 row tiles at 128 x 48 agents, a third of the chip's SIMDs with one wave each -- and an
// iteration of the one-wave kernel is ~4.5 us of ONE wave's issue slots (64 tanh, 16 normals, five operand splits per lane beside 120
// matrix instructions) 61 times over.  Here the four waves of a workgroup share a tile as in the recurrence above: wave w produces
// channels [16w, 16w + 16) of every layer of BOTH nets (its slices of the five matrices stay in registers), the activations that feed
// a product travel as pre-split operand tiles through the LDS, the diffusion head's dot product as four partial sums.  Three barriers
// an iteration.  The matrices are the fused forward kernel's own image (DecSdeL6: layers in front of a tanh packed times 2 / ln 2,
// first layers of drift and diffusion stacked), and per output the products run in the fused kernel's order: the replayed drift,
// activations and states are the forward's own to the bit; the diffusion value sums its 64 terms in another order.
#if TSDE_SPLIT_H3
__global__ __launch_bounds__(256) void k_sde_replay_coop(const float* __restrict__ img, const int32_t* __restrict__ best, int N, int K,
                                                         int n_euler, const float* __restrict__ step_tab, NoiseArg na,
                                                         float* __restrict__ states, float* __restrict__ H1, float* __restrict__ H2,
                                                         float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ GS) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using DD = DecSdeL6;
  float* const Yop = lds;                                  // operand tiles: the state, first-layer activations of f / g, f's second
  float* const Af = lds + COOP_OT;
  float* const Ag = lds + 2 * COOP_OT;
  float* const Bf = lds + 3 * COOP_OT;
  float* const dotp = lds + 4 * COOP_OT;                   // [wave][16 rows]: partial dots of the diffusion head
  const Lane L;
  const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const WSlice w0f = load_slice(img + DD::W0FG, w, L.lane), w0g = load_slice(img + DD::W0FG, 4 + w, L.lane);
  const WSlice w2f = load_slice(img + DD::F_W2, w, L.lane), w4f = load_slice(img + DD::F_W4, w, L.lane);
  const WSlice w2g = load_slice(img + DD::G_W2, w, L.lane);
  const int ch = 16 * w + 4 * L.g;                          // this lane's four channels of its row
  const f4 b0f = *reinterpret_cast<const f4*>(img + DD::B0FG + ch), b0g = *reinterpret_cast<const f4*>(img + DD::B0FG + 64 + ch);
  const f4 wsf = *reinterpret_cast<const f4*>(img + DD::WSFG + ch), wsg = *reinterpret_cast<const f4*>(img + DD::WSFG + 64 + ch);
  const f4 wcf = *reinterpret_cast<const f4*>(img + DD::WCFG + ch), wcg = *reinterpret_cast<const f4*>(img + DD::WCFG + 64 + ch);
  const f4 b2f = vec_slice(img + DD::F_B2, w, L.g), b4f = vec_slice(img + DD::F_B4, w, L.g), b2g = vec_slice(img + DD::G_B2, w, L.g);
  const f4 w4g = vec_slice(img + DD::G_W4, w, L.g);
  const float b4g = img[DD::G_B4];
  const int ntiles = (N + 15) / 16;
  const int64_t slab = int64_t(N) * D;
  const uint64_t key = noise_key(na);                      // (read once: through `seed_dev` it is a load, and it was one per iteration)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * 16 + L.n;
    const bool live = row < N;
    const int i = live ? row : N - 1;
    const int64_t r = int64_t(best[i]) * N + i;
    const uint32_t rid = na.row_ids ? uint32_t(na.row_ids[r]) : uint32_t(r);
    f4 y = *reinterpret_cast<const f4*>(states + int64_t(i) * D + ch);
    lds_barrier();                                       // the previous tile's readers of the operand tiles are done
    opnd_write(Yop, y, w, L);
    for (int k = 0; k < n_euler; ++k) {
      const float dt = step_tab[k * 8 + 1], sq = step_tab[k * 8 + 2], sn = step_tab[k * 8 + 3], cs = step_tab[k * 8 + 4];
      // the iteration's normals for this wave's 16 channels: vector work that needs nothing, first
      f4 z;
      if (na.z != nullptr) z = *reinterpret_cast<const f4*>(na.z + (int64_t(k) * N * K + r) * D + ch);
      else z = philox_normal4(key, STREAM_DECODER, uint32_t(k), rid, uint32_t(4 * w + L.g));
      lds_barrier();                                     // y of this iteration is in Yop
      // ---- first layers of both nets on one read of the state
      f4 h1, g1;
      {
        const Opnd oy = opnd_read(Yop, L);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h1[c] = fmaf(wcf[c], cs, fmaf(wsf[c], sn, b0f[c]));
          g1[c] = fmaf(wcg[c], cs, fmaf(wsg[c], sn, b0g[c]));
        }
        slice_mma(h1, w0f, oy);
        slice_mma(g1, w0g, oy);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h1[c] = tanh_prescaled(h1[c]);
          g1[c] = tanh_prescaled(g1[c]);
        }
      }
      opnd_write(Af, h1, w, L);
      opnd_write(Ag, g1, w, L);
      if (live) {
        *reinterpret_cast<f4*>(H1 + k * slab + int64_t(row) * D + ch) = h1;
        *reinterpret_cast<f4*>(G1 + k * slab + int64_t(row) * D + ch) = g1;
      }
      lds_barrier();
      // ---- second layers; the diffusion head's partial dot over this wave's channels
      f4 h2 = b2f, g2 = b2g;
      slice_mma(h2, w2f, opnd_read(Af, L));
      slice_mma(g2, w2g, opnd_read(Ag, L));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        h2[c] = tanh_prescaled(h2[c]);
        g2[c] = tanh_prescaled(g2[c]);
      }
      opnd_write(Bf, h2, w, L);
      {
        float pd = g2[0] * w4g[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) pd = fmaf(g2[c], w4g[c], pd);
        pd = row_sum(pd);                                  // over the four lanes of the row: this wave's 16 channels
        if (L.g == 0) dotp[16 * w + L.n] = pd;
      }
      if (live) {
        *reinterpret_cast<f4*>(H2 + k * slab + int64_t(row) * D + ch) = h2;
        *reinterpret_cast<f4*>(G2 + k * slab + int64_t(row) * D + ch) = g2;
      }
      lds_barrier();
      // ---- drift output, diffusion value, the Euler-Maruyama step on this wave's channels (sde_funcs.hpp em_update)
      f4 f = b4f;
      slice_mma(f, w4f, opnd_read(Bf, L));
      const float gs = fast_sigmoid(((dotp[L.n] + dotp[16 + L.n]) + (dotp[32 + L.n] + dotp[48 + L.n])) + b4g);
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = (y[c] + f[c] * dt) + gs * (z[c] * sq);
      opnd_write(Yop, y, w, L);                            // (its readers of this iteration passed two barriers ago)
      if (live) {
        *reinterpret_cast<f4*>(states + (k + 1) * slab + int64_t(row) * D + ch) = y;
        if (w == 0 && L.g == 0) GS[int64_t(k) * N + row] = gs;
      }
    }
  }
}

// The decoder's REVERSE SWEEP through the Euler-Maruyama steps in the same cooperative form (decoder_bwd.hip k_sde_bwd is the one-wave
// form: 5.5 us an iteration on one wave a tile).  Wave w owns channels [16w, 16w + 16) of every delta; its slices of the five TRANSPOSED
// matrices (SweepL) stay in registers; deltas that feed a product travel as pre-split operand tiles.  Deltas span many binades and fp16
// has five exponent bits, so -- as in k_enc_recur_bwd_coop -- a row is normalised ONCE per iteration by the power of two of the largest
// magnitude of the gradient entering it (everything the iteration computes for the row is linear in that gradient); the factor comes
// off again at every store and where the products rejoin d y.  Four barriers an iteration:
//   B0 row magnitudes, partial dots of the diffusion scalar | B1 d f -> drift.4^T input | B2 d h2, d g2 | B3 d h1, d g1 | (first layers^T -> d y)
// Same stored deltas as the one-wave kernel up to the rounding of one shared scale per row instead of one per product.
__global__ __launch_bounds__(256) void k_sde_bwd_coop(const SdeBwdCoopArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using SL = SweepL;
  auto Tb = [&](int i) { return lds + i * COOP_OT; };      // operand tiles 0..4
  float* const MX = lds + 5 * COOP_OT;                      // [wave][16]: row magnitudes
  float* const CD = MX + 64;                                // [wave][16]: partial dots z sqrt(h) . d y
  // the step and output tables, once, into the LDS: read through the argument struct they are VECTOR loads from global memory (the
  // compiler cannot prove that the kernel's own stores leave them alone), and the `while` over the output steps below made one or two of
  // them, dependent, per iteration -- a round trip to L2 each on a chain that has nothing else to do
  float* const stab = CD + 64;                              // [n_euler][8]
  float* const otab = stab + 8 * a.n_euler;                 // [T][4]
  for (int q = threadIdx.x; q < 8 * a.n_euler; q += blockDim.x) stab[q] = a.step_tab[q];
  for (int q = threadIdx.x; q < 4 * a.T; q += blockDim.x) otab[q] = a.out_tab[q];
  const Lane L;
  const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const WSlice wf4 = load_slice(a.img + SL::F_W4T, w, L.lane), wf2 = load_slice(a.img + SL::F_W2T, w, L.lane);
  const WSlice wf0 = load_slice(a.img + SL::F_W0T, w, L.lane), wg2 = load_slice(a.img + SL::G_W2T, w, L.lane);
  const WSlice wg0 = load_slice(a.img + SL::G_W0T, w, L.lane);
  const f4 w4g = vec_slice(a.img + SL::G_W4, w, L.g);
  const int ch = 16 * w + 4 * L.g;
  const int N = a.N, n_euler = a.n_euler;
  const int ntiles = (N + 15) / 16;
  const int64_t slab = int64_t(N) * D;
  f4 dv4 = f4{0.f, 0.f, 0.f, 0.f};
  float dc4 = 0.f;
  const uint64_t key = noise_key(a.na);
  __syncthreads();                                          // the tables are staged
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * 16 + L.n;
    const bool live = row < N;
    const int i = live ? row : N - 1;
    const int64_t r = int64_t(a.best[i]) * N + i;
    const uint32_t rid = a.na.row_ids ? uint32_t(a.na.row_ids[r]) : uint32_t(r);
    auto at = [&](const float* slabs, int k) { return *reinterpret_cast<const f4*>(slabs + k * slab + int64_t(i) * D + ch); };
    f4 dy = f4{0.f, 0.f, 0.f, 0.f};                         // dL/dy_{k+1} on entry of iteration k (this wave's channels)
    int o = a.T - 1;
    // saved activations, and the head gradient of the next output step: requested one iteration ahead
    f4 nh2 = at(a.H2, n_euler - 1), nh1 = at(a.H1, n_euler - 1), ng2 = at(a.G2, n_euler - 1), ng1 = at(a.G1, n_euler - 1);
    float ngs = a.GS[int64_t(n_euler - 1) * N + i];
    f4 nds = o >= 0 ? at(a.DS, o) : f4{0.f, 0.f, 0.f, 0.f};
    for (int k = n_euler - 1; k >= 0; --k) {
      const float dt = stab[k * 8 + 1], sq = stab[k * 8 + 2];
      const f4 h2 = nh2, h1 = nh1, g2 = ng2, g1 = ng1;
      const float gs = ngs;
      // (injected normals are a load: it goes in FRONT of the look-ahead requests -- the memory counter is in order, and a value asked
      //  for after them could only be waited for together with them)
      f4 z;
      if (a.na.z != nullptr) z = *reinterpret_cast<const f4*>(a.na.z + (int64_t(k) * N * a.K + r) * D + ch);
      else z = philox_normal4(key, STREAM_DECODER, uint32_t(k), rid, uint32_t(4 * w + L.g));
      if (k > 0) {
        nh2 = at(a.H2, k - 1); nh1 = at(a.H1, k - 1); ng2 = at(a.G2, k - 1); ng1 = at(a.G1, k - 1);
        ngs = a.GS[int64_t(k - 1) * N + i];
      }
      // outputs interpolated between y_k and y_{k+1}: s_o = w0 y_k + w1 y_{k+1}
      f4 dprev = f4{0.f, 0.f, 0.f, 0.f};
      // (the FIRST output of a step is taken outside the loop: inside it, the request for the next output's rows is followed by the
      //  loop's back edge, where the compiler must assume the rows are used at once and waits for them -- and, the memory counter being
      //  in order, for everything requested before them: the whole look-ahead above, every iteration)
      if (o >= 0 && int(otab[o * 4]) == k + 1) {
        const float w0 = otab[o * 4 + 1], w1 = otab[o * 4 + 2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dy[c] = fmaf(w1, nds[c], dy[c]);                   // (DS rows of output o: asked for an iteration ago)
          dprev[c] = fmaf(w0, nds[c], dprev[c]);
        }
        --o;
        while (o >= 0 && int(otab[o * 4]) == k + 1) {        // further outputs inside the same Euler step: not in the shipped schedules
          const float v0 = otab[o * 4 + 1], v1 = otab[o * 4 + 2];
          const f4 ds = at(a.DS, o);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            dy[c] = fmaf(v1, ds[c], dy[c]);
            dprev[c] = fmaf(v0, ds[c], dprev[c]);
          }
          --o;
        }
        if (o >= 0) nds = at(a.DS, o);
      }
      // ---- B0: the row's magnitude and the diffusion scalar's dot product, this wave's share
      {
        const float m = row_max(fmaxf(fmaxf(fabsf(dy[0]), fabsf(dy[1])), fmaxf(fabsf(dy[2]), fabsf(dy[3]))));
        float cd = (z[0] * sq) * dy[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) cd = fmaf(z[c] * sq, dy[c], cd);
        cd = row_sum(cd);
        if (L.g == 0) {
          MX[16 * w + L.n] = m;
          CD[16 * w + L.n] = cd;
        }
      }
      lds_barrier();
      const unsigned e = __float_as_uint(fmaxf(fmaxf(MX[L.n], MX[16 + L.n]), fmaxf(MX[32 + L.n], MX[48 + L.n]))) & 0x7F800000u;   // 0: an all-zero row
      const float up = __uint_as_float(0x7F000000u - e), down = __uint_as_float(e);
      const float cdot = (CD[L.n] + CD[16 + L.n]) + (CD[32 + L.n] + CD[48 + L.n]);
      // ---- B1: drift net: y' gets f dt
      {
        f4 d;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = dt * dy[c];
        if (live) *reinterpret_cast<f4*>(a.DF + k * slab + int64_t(row) * D + ch) = d;
        opnd_write(Tb(0), d * up, w, L);
      }
      lds_barrier();
      // ---- B2: drift.4^T; the diffusion head (64 -> 1, sigmoid): y' gets g (z sqrt(h)), g one scalar per row
      {
        const f4 t = zero_mma(wf4, opnd_read(Tb(0), L));
        f4 d;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = t[c] * (1.0f - h2[c] * h2[c]);
        if (live) *reinterpret_cast<f4*>(a.DH2 + k * slab + int64_t(row) * D + ch) = d * down;
        opnd_write(Tb(1), d, w, L);
        const float dgp = cdot * gs * (1.0f - gs);
        const float dgps = dgp * up;
        f4 dg;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (live) dv4[c] = fmaf(dgp, g2[c], dv4[c]);
          dg[c] = dgps * w4g[c] * (1.0f - g2[c] * g2[c]);
        }
        if (live && w == 0) dc4 += dgp;
        if (live) *reinterpret_cast<f4*>(a.DG2 + k * slab + int64_t(row) * D + ch) = dg * down;
        opnd_write(Tb(2), dg, w, L);
      }
      lds_barrier();
      // ---- B3: drift.2^T, diffusion.2^T
      {
        const f4 t = zero_mma(wf2, opnd_read(Tb(1), L));
        const f4 u = zero_mma(wg2, opnd_read(Tb(2), L));
        f4 d, dg;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          d[c] = t[c] * (1.0f - h1[c] * h1[c]);
          dg[c] = u[c] * (1.0f - g1[c] * g1[c]);
        }
        if (live) {
          *reinterpret_cast<f4*>(a.DH1 + k * slab + int64_t(row) * D + ch) = d * down;
          *reinterpret_cast<f4*>(a.DG1 + k * slab + int64_t(row) * D + ch) = dg * down;
        }
        opnd_write(Tb(3), d, w, L);
        opnd_write(Tb(4), dg, w, L);
      }
      lds_barrier();
      // ---- first layers^T (the 64 state columns of the 66-wide inputs) -> d y_k, true scale.  No barrier behind it: the next writes
      //      to the tiles it reads (3, 4) come three barriers later.
      {
        f4 adj = zero_mma(wf0, opnd_read(Tb(3), L));
        slice_mma(adj, wg0, opnd_read(Tb(4), L));
#pragma unroll
        for (int c = 0; c < 4; ++c) dy[c] = fmaf(adj[c], down, dy[c] + dprev[c]);
      }
    }
    if (live) *reinterpret_cast<f4*>(a.DY0 + int64_t(row) * D + ch) = dy;
  }
  // this workgroup's row of vector partials: d diffusion.4.weight (this wave's 16 channels, summed over the rows) and its bias
  float* vp = a.vpart + int64_t(blockIdx.x) * SDE_SWEEP_V_FLOATS;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = dv4[c];
    x += __shfl_xor(x, 1);
    x += __shfl_xor(x, 2);
    x += __shfl_xor(x, 4);
    x += __shfl_xor(x, 8);
    dv4[c] = x;
  }
  if (L.n == 0) *reinterpret_cast<f4*>(vp + ch) = dv4;
  if (w == 0) {
    float x = L.g == 0 ? dc4 : 0.f;
    x += __shfl_xor(x, 1);
    x += __shfl_xor(x, 2);
    x += __shfl_xor(x, 4);
    x += __shfl_xor(x, 8);
    if (L.lane == 0) vp[64] = x;
  }
}

#endif

// forward_ood (ENC:311-313): outs [S,N,64] -> mean over samples [N,64] and std(0).mean(-1) [N] (unbiased std)
__global__ __launch_bounds__(256) void k_ood_stats(const float* __restrict__ samples, int S, int N, float* __restrict__ mean,
                                                   float* __restrict__ stds) {
  const int lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  float m = 0.f;
  for (int s = 0; s < S; ++s) m += samples[(int64_t(s) * N + row) * 64 + lane];
  m /= float(S);
  float v = 0.f;
  for (int s = 0; s < S; ++s) {
    const float d = samples[(int64_t(s) * N + row) * 64 + lane] - m;
    v += d * d;
  }
  float sd = S > 1 ? sqrtf(v / float(S - 1)) : 0.f;
  mean[row * 64 + lane] = m;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sd += __shfl_xor(sd, o);
  if (lane == 0) stds[row] = sd * (1.0f / 64.0f);
}

}  // namespace tsde
