// decoder.hip -- SDEDecoder (reference models/decoders/dec_hivt_nusargo_sde.py:77-105) for gfx950.
//
//   k_dec_init    y0 = ReLU(LN(Linear(128->64)(cat(global, local))))  (DEC:82)   and   pi head (DEC:93-94)
//   k_sde_decode  the whole Euler-Maruyama solve of stock torchsde.sdeint (DEC:88) fused with the loc/scale
//                 heads (DEC:95-98): one 16-path tile per wave, state in registers for all steps, drift and
//                 diffusion MLP weights + heads resident in LDS (~120 KB, fragment order), noise from
//                 in-kernel Philox (or injected), time bookkeeping replayed from the float32 schedule table
//                  so the micro-step of T in {30,50,60} is reproduced.
//   k_sde_step    the same step with the state round-tripping HBM (512 B / path-step), for the roofline report.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "layouts.hpp"
#include "philox.hpp"
#include "range.hpp"
#include "sde_funcs.hpp"
#include "stamps.hpp"
#include "tile.hpp"

TSDE_STAMP_TABLE(sde_step, 8)   // diagnostic builds (tools/phase_stamps.py): phases of one tile-step of k_sde_step

namespace tsde {
#ifndef TSDE_STAMPS
static unsigned long long* const g_stamps_sde_step = nullptr;
#endif

// Linear(64,64)-LN-ReLU-Linear(64,2) head on state s -> (o0, o1)
__device__ __forceinline__ void head_eval(float& o0, float& o1, const f4 (&s)[4], const float* img, const Lane& L) {
  f4 h[4];
  linear<4, 4>(h, s, img + HeadL::W0, img + HeadL::B0, L);
  layer_norm<4>(h, img + HeadL::G, img + HeadL::E, L.g);
  relu<4>(h);
  o0 = row_dot(h, img + HeadL::W3, L.g) + img[HeadL::B3];
  o1 = row_dot(h, img + HeadL::W3 + 64, L.g) + img[HeadL::B3 + 1];
}

#if TSDE_SPLIT_H3
// both heads of the fused kernel on one split of s: stacked first layers (HeadPairL6), fp16x3 matrix products
__device__ __forceinline__ void head_pair_eval(float& lx, float& ly, float& sx, float& sy, const f4 (&s)[4], const float* img,
                                               const Lane& L) {
  using HP = HeadPairL6;
  f4 h[8];
  linear_x6<8, 4>(h, s, img + HP::W0, img + HP::B0, L);
  f4 a[4] = {h[0], h[1], h[2], h[3]}, b[4] = {h[4], h[5], h[6], h[7]};
  layer_norm<4>(a, img + HP::G_LOC, img + HP::E_LOC, L.g);
  relu<4>(a);
  lx = row_dot(a, img + HP::W3_LOC, L.g) + img[HP::B3_LOC];
  ly = row_dot(a, img + HP::W3_LOC + 64, L.g) + img[HP::B3_LOC + 1];
  layer_norm<4>(b, img + HP::G_SC, img + HP::E_SC, L.g);
  relu<4>(b);
  sx = row_dot(b, img + HP::W3_SC, L.g) + img[HP::B3_SC];
  sy = row_dot(b, img + HP::W3_SC + 64, L.g) + img[HP::B3_SC + 1];
}
#endif

__global__ __launch_bounds__(1024) void k_dec_init(const float* __restrict__ blob, const float* __restrict__ local,
                                                   const float* __restrict__ global, int N, int K,
                                                   float* __restrict__ y0, float* __restrict__ pi, int st_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, blob, DecInitL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t rows = int64_t(N) * K;
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n;
    const int64_t r = row < rows ? row : rows - 1;
    f4 gl[4], lo[4], a[4];
    load_row(gl, global, r, L.g);
    load_row(lo, local, r % N, L.g);
    range_note(fmaxf(absmax<4>(gl), absmax<4>(lo)), RS_DEC_INPUT);
    load_vec<4>(a, lds + DecInitL::BA, L.g);
    linear_acc<4, 4>(a, gl, lds + DecInitL::WA_G, L.lane);
    linear_acc<4, 4>(a, lo, lds + DecInitL::WA_L, L.lane);
    layer_norm<4>(a, lds + DecInitL::AG, lds + DecInitL::AE, L.g);
    relu<4>(a);
    if (row < rows) store_row_st(a, y0, row, L.g, st_bf16 != 0);
    load_vec<4>(a, lds + DecInitL::BP, L.g);
    linear_acc<4, 4>(a, lo, lds + DecInitL::WP_L, L.lane);
    linear_acc<4, 4>(a, gl, lds + DecInitL::WP_G, L.lane);
    layer_norm<4>(a, lds + DecInitL::PG, lds + DecInitL::PE, L.g);
    relu<4>(a);
    const float p = row_dot(a, lds + DecInitL::WP3, L.g) + lds[DecInitL::BP3];
    if (row < rows && L.g == 0) pi[(r % N) * K + (r / N)] = p;          // pi is [N, K]  (.t() at DEC:94)
  }
}

template <bool X6, int MAXT>
__global__ __launch_bounds__(MAXT) void k_sde_decode(const float* __restrict__ blob, const float* __restrict__ y0,
                                                     int64_t rows, int T, int n_euler,
                                                     const float* __restrict__ step_tab, const float* __restrict__ out_tab,
                                                     float min_scale, NoiseArg na, float* __restrict__ loc, int st_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using DL = typename std::conditional<X6, DecSdeL6, DecSdeL>::type;
  stage_blob(lds, blob, DL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (rows + 15) / 16;
  // Tiles are dealt in blocks: workgroup b owns a contiguous share of them and hands them to its waves in rounds, wave 0 first -- so when
  // the share is not a multiple of the wave count, the extra tiles go to the waves with the lowest index, the ones that reach their SIMDs
  // first and are served first (findings 2 and 9).  A grid-stride deal gives those tiles to whole workgroups instead.
  const int64_t share = ntiles / gridDim.x, extra = ntiles % gridDim.x;
  const int64_t first = int64_t(blockIdx.x) * share + (int64_t(blockIdx.x) < extra ? int64_t(blockIdx.x) : extra);
  const int64_t mine = share + (int64_t(blockIdx.x) < extra ? 1 : 0);
  for (int64_t p = wave; p < mine; p += waves) {
    const int64_t tile = first + p;
    const int64_t row = tile * 16 + L.n;
    const int64_t r = row < rows ? row : rows - 1;
    f4 y[4], prev[4];
    load_row_st(y, y0, r, L.g, st_bf16 != 0);
    int o = 0;
    for (int k = 0; k < n_euler; ++k) {
      keep_lds_reads_here();
      const float dt = step_tab[k * 8 + 1], sq = step_tab[k * 8 + 2], sn = step_tab[k * 8 + 3], cs = step_tab[k * 8 + 4];
      f4 f[4], z[4];
      float gs;
#if TSDE_SPLIT_H3
      if constexpr (X6) {
        // this step's 128 first-layer biases, once per step in the wave's own LDS slot (they were 64 fma + 24 LDS reads per tile)
        float* tb = lds + DL::SIZE + wave * 128;
        __builtin_amdgcn_wave_barrier();                       // the previous step's reads of the slot are done (same wave, in order)
        sde_time_bias(tb, lds, sn, cs, L.lane);
        sde_time_bias(tb, lds, sn, cs, L.lane + 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        sde_fg_eval(f, gs, y, lds, tb, L);
      } else
#else
      if constexpr (X6) {
        drift_eval_x6(f, y, lds + DL::F, sn, cs, L);
        gs = diff_eval_x6(y, lds + DL::G, sn, cs, L);
      } else
#endif
      {
        drift_eval(f, y, lds + DL::F, sn, cs, L);
        gs = diff_eval(y, lds + DL::G, sn, cs, L);
      }
      noise_row(z, na, STREAM_DECODER, k, r, rows, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) prev[jt] = y[jt];
      em_update(y, f, gs, z, dt, sq);
      range_note(absmax<4>(y), RS_DEC_STATE);                 // the state is the next step's (and the heads') split operand
      // emit every output whose interpolation bracket closes with this step (linear_interp of the solver)
      while (o < T && int(out_tab[o * 4]) == k + 1) {
        keep_lds_reads_here();
        const float w0 = out_tab[o * 4 + 1], w1 = out_tab[o * 4 + 2];
        f4 s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int c = 0; c < 4; ++c) s[jt][c] = w0 * prev[jt][c] + w1 * y[jt][c];
        float lx, ly, sx, sy;
#if TSDE_SPLIT_H3
        if constexpr (X6) head_pair_eval(lx, ly, sx, sy, s, lds + DL::LOC, L);
        else
#endif
        {
          head_eval(lx, ly, s, lds + DL::LOC, L);
          head_eval(sx, sy, s, lds + DL::SCALE, L);
        }
        sx = (sx > 0.f ? sx : fast_exp(sx) - 1.0f) + 1.0f + min_scale;       // ELU(alpha=1) + 1 + min_scale (DEC:97-98)
        sy = (sy > 0.f ? sy : fast_exp(sy) - 1.0f) + 1.0f + min_scale;
        if (row < rows && L.g == 0) {
          f4 v = {lx, ly, sx, sy};
          *reinterpret_cast<f4*>(loc + (row * T + o) * 4) = v;
        }
        ++o;
      }
    }
  }
}

// step-granular variant: one Euler-Maruyama step, state read from and written to HBM (no heads)
//
// Round 4: the in-kernel phase clocks (tools/phase_stamps.py sde_step) showed a wave spending 31 % of a tile-step waiting for its 16
// state rows -- the load was issued at the top of the iteration and needed at once -- so the rows of a wave's NEXT tile now travel
// while it computes the current one: by LDS-DMA (global_load_lds_dwordx4, no registers: the kernel sits at 115 of the 128 registers
// that four waves per SIMD allow) into a 4 KB slot per wave, 16-byte chunks XOR-swizzled by the row so that the transfer reads four
// whole rows per instruction and the lanes' ds_read_b128 of "row on lane" find their chunks in distinct banks.
constexpr int SDE_STEP_SLOT = 16 * 64;                      // floats per wave: one 16 x 64 tile
template <bool X6>
__global__ __launch_bounds__(1024) void k_sde_step(const float* __restrict__ blob, const float* __restrict__ y_in,
                                                   float* __restrict__ y_out, int64_t rows, float dt, float sq, float sn,
                                                   float cs, int step, NoiseArg na, int st_bf16) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using DL = typename std::conditional<X6, DecSdeL6, DecSdeL>::type;
  stage_blob(lds, blob, DL::LOC);   // F and G images only
#if TSDE_SPLIT_H3
  float* tb = lds + DL::LOC;        // the step's 128 first-layer biases: constants of the launch
  if (X6) {
    if (threadIdx.x < 128) sde_time_bias(tb, lds, sn, cs, threadIdx.x);
    __syncthreads();
  }
#endif
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (rows + 15) / 16;
  PhaseClock<8> clk;                                       // (diagnostic builds only: stamps.hpp)
  clk.start();
  unsigned long long units = 0;
  (void)units;
  // one tile-step on the state rows in y: (row < rows: the row exists and is stored)
  // (plain_noise: Philox keyed by the kernel argument and the row index itself -- no loads, so nothing in the step waits on the
  //  vector-memory counter and the transfer of the next tile stays in flight; the other noise forms take the loop at the bottom)
  const uint64_t key0 = na.seed;
  auto noise_of = [&](f4 (&z)[4], int64_t r, bool plain_noise) {
    if (plain_noise) {
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) z[jt] = philox_normal4(key0, STREAM_DECODER, uint32_t(step), uint32_t(r), uint32_t(4 * jt + L.g));
    } else {
      noise_row(z, na, STREAM_DECODER, step, r, rows, L.g);
    }
  };
  auto tile_step = [&](f4 (&y)[4], int64_t row, int64_t r, bool store_all, bool plain_noise) {
    f4 f[4], z[4];
    range_note(absmax<4>(y), RS_DEC_STATE);
    float gs;
#if TSDE_SPLIT_H3
    if constexpr (X6) {
#if defined(TSDE_STAMPS) || (defined(TSDE_SDE_ALIGN) && TSDE_SDE_ALIGN >= 2)
      // the phases of sde_fg_eval spelled out between the clock's marks (same calls, same order)
      // (-DTSDE_SDE_ALIGN=2, a timing experiment of round 6: a workgroup barrier at every mark, so that the four waves of a SIMD run their
      //  vector-only phases together and their matrix phases together -- profiles/r06_sde_step_floor.md)
#if defined(TSDE_SDE_ALIGN) && TSDE_SDE_ALIGN >= 2 && !defined(TSDE_STAMPS)
      struct { __device__ __forceinline__ void mark(int) { __builtin_amdgcn_s_barrier(); } } clk;
      unsigned long long units = 0;
#endif
      using DD = DecSdeL6;
      noise_of(z, r, plain_noise);
      asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]));
      clk.mark(1);                                            // [1] Philox + Box-Muller: 16 normals per lane
      f4 h[8];
#pragma unroll
      for (int jo = 0; jo < 8; ++jo) h[jo] = *reinterpret_cast<const f4*>(tb + 16 * jo + 4 * L.g);
      linear_acc_x6<8, 4>(h, y, lds + DD::W0FG, L.lane);
      asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]));
      clk.mark(2);                                            // [2] first layers: split of the state + 48 matrix instructions
      tanh_prescaled_<8>(h);
      asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]));
      clk.mark(3);                                            // [3] 32 tanh per lane
      const f4 hf[4] = {h[0], h[1], h[2], h[3]}, hg[4] = {h[4], h[5], h[6], h[7]};
      f4 h2[4];
      linear_x6<4, 4>(h2, hf, lds + DD::F_W2, lds + DD::F_B2, L);
      tanh_prescaled_<4>(h2);
      linear_x6<4, 4>(f, h2, lds + DD::F_W4, lds + DD::F_B4, L);
      asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
      clk.mark(4);                                            // [4] drift: second layer + tanh + output layer (48 matrix instructions)
      linear_x6<4, 4>(h2, hg, lds + DD::G_W2, lds + DD::G_B2, L);
      tanh_prescaled_<4>(h2);
      gs = fast_sigmoid(row_dot(h2, lds + DD::G_W4, L.g) + lds[DD::G_B4]);
      asm volatile("" : "+v"(gs));
      clk.mark(5);                                            // [5] diffusion: second layer + tanh + head dot + sigmoid
      em_update(y, f, gs, z, dt, sq);
      if (store_all || row < rows) store_row_st(y, y_out, row, L.g, st_bf16 != 0);
      clk.mark(6);                                            // [6] update + store
      ++units;
      return;
#else
      sde_fg_eval(f, gs, y, lds, tb, L);
#endif
    } else
#else
    if constexpr (X6) {
      drift_eval_x6(f, y, lds + DL::F, sn, cs, L);
      gs = diff_eval_x6(y, lds + DL::G, sn, cs, L);
    } else
#endif
    {
      drift_eval(f, y, lds + DL::F, sn, cs, L);
      gs = diff_eval(y, lds + DL::G, sn, cs, L);
    }
    noise_of(z, r, plain_noise);
    em_update(y, f, gs, z, dt, sq);
    if (store_all || row < rows) store_row_st(y, y_out, row, L.g, st_bf16 != 0);
  };
  const int64_t stride = int64_t(gridDim.x) * waves;
  int64_t tile = int64_t(blockIdx.x) * waves + wave;
#if TSDE_SPLIT_H3
  if (X6 && st_bf16 == 0 && na.z == nullptr && na.row_ids == nullptr && na.seed_dev == nullptr) {
    // ---- full tiles (all 16 rows exist), the next one in flight while this one is computed
    const int64_t full = rows / 16;
    float* slot = lds + DL::LOC + 128 + wave * SDE_STEP_SLOT;
    const unsigned slot_addr = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)slot)));
    // lane l of transfer q brings chunk (l & 15) ^ rr of row rr = 4 q + (l >> 4), which lands at slot + 1024 q + 16 l bytes = chunk
    // position (l & 15) of row rr: position p of row rr holds chunk p ^ rr
    const int rr0 = L.lane >> 4, pc = L.lane & 15;
    auto send = [&](int64_t t) {
      const float* base = y_in + t * (16 * 64);
      const float* s0 = base + (rr0 + 0) * 64 + 4 * (pc ^ (rr0 + 0));
      const float* s1 = base + (rr0 + 4) * 64 + 4 * (pc ^ (rr0 + 4));
      const float* s2 = base + (rr0 + 8) * 64 + 4 * (pc ^ (rr0 + 8));
      const float* s3 = base + (rr0 + 12) * 64 + 4 * (pc ^ (rr0 + 12));
      unsigned m0_keep;
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %5\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %3, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %4, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(m0_keep)
                   : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(slot_addr)
                   : "memory", "scc");
    };
    bool first = true;
    if (tile < full) send(tile);
#if defined(TSDE_SDE_ALIGN) && TSDE_SDE_ALIGN >= 1
    // timing experiment (round 6): every wave of the workgroup takes its tile-step at the same time.  (Waves of one workgroup may differ
    // by one iteration; the tail of the list is then NOT processed by the short ones' partners: results are wrong, timings are what this is for.)
    const int64_t iters_wg = (full - int64_t(blockIdx.x) * waves + stride - 1) / stride;      // of the workgroup's first wave
    for (int64_t it_ = 0; it_ < iters_wg; ++it_, tile += stride) {
      __builtin_amdgcn_s_barrier();
      if (tile >= full) continue;
      keep_lds_reads_here();
#else
    for (; tile < full; tile += stride) {
      keep_lds_reads_here();
#endif
      // the four transfers of this tile are the oldest vector-memory operations in flight; behind them sit at most the four row
      // stores of the previous tile (vmcnt counts in issue order), so "all but four" means the tile has landed
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      first = false;
      f4 y[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) y[jt] = *reinterpret_cast<const f4*>(slot + L.n * 64 + 4 * ((4 * jt + L.g) ^ L.n));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : : "memory");   // the slot is free again
      clk.mark(0);                                              // [0] waiting for the state rows (and the loop overhead)
      if (tile + stride < full) send(tile + stride);
      const int64_t row = tile * 16 + L.n;
      tile_step(y, row, row, true, true);
    }
  }
#endif
  // ---- the last, partial tile (and every tile of the forms without the transfer): rows asked for where they are used
  for (; tile < ntiles; tile += stride) {
    const int64_t row = tile * 16 + L.n;
    keep_lds_reads_here();
    const int64_t r = row < rows ? row : rows - 1;
    f4 y[4];
    load_row_st(y, y_in, r, L.g, st_bf16 != 0);
#ifdef TSDE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    clk.mark(0);
#endif
    tile_step(y, row, r, false, false);
  }
#ifdef TSDE_STAMPS
  if ((wave & 3) == 0 && L.lane == 0) clk.flush(g_stamps_sde_step, units);
#endif
}

int pick_grid(int64_t ntiles, int waves_per_wg) {
  int64_t wgs = (ntiles + waves_per_wg - 1) / waves_per_wg;
  return int(wgs < 1 ? 1 : (wgs > 256 ? 256 : wgs));   // one resident workgroup per CU (LDS-bound), grid-stride beyond
}

}  // namespace tsde

using namespace tsde;

static NoiseArg to_arg(const trajsde_noise* n) {
  NoiseArg a{0, nullptr, nullptr};
  if (n) { a.seed = n->seed; a.z = n->z; a.row_ids = n->row_ids; a.seed_dev = n->seed_dev; }
  return a;
}

extern "C" {

int64_t trajsde_decoder_ws_bytes(int32_t N, int num_modes) { return align_up(int64_t(N) * num_modes * 64 * 4, 256) + 256; }

int trajsde_decoder_forward(int32_t N, int num_modes, int future_steps, const float* blob, const float* local_embed,
                            const float* global_embed, const float* step_table, int n_euler, const float* out_table,
                            float min_scale, const trajsde_noise* noise, void* ws, int64_t ws_bytes, float* loc,
                            float* pi, void* stream_) {
  TS_REQUIRE(blob && local_embed && global_embed && step_table && out_table && loc && pi && ws, "decoder_forward: null pointer");
  TS_REQUIRE(N > 0 && num_modes > 0 && future_steps > 0 && n_euler > 0, "decoder_forward: empty problem");
  if (ws_bytes < trajsde_decoder_ws_bytes(N, num_modes)) return fail(TRAJSDE_ERR_WORKSPACE, "decoder_forward: workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Carver cv(ws, ws_bytes);
  const int64_t rows = int64_t(N) * num_modes;
  float* y0 = cv.take<float>(rows * 64);
  const int64_t ntiles = (rows + 15) / 16;
  TS_LAUNCH(k_dec_init, pick_grid(ntiles, 8), 512, DecInitL::SIZE * 4, stream, blob + DecBlob::INIT, local_embed, global_embed, N,
            num_modes, y0, pi, state_bf16() ? 1 : 0);
  // 768 threads = 12 waves = 3 per SIMD (168 VGPRs each); 256 CUs x 12 waves = 3072 path tiles in flight
  static const int dthreads = []() { const char* v = getenv("TRAJSDE_THREADS_DECODE"); const int t = v ? atoi(v) : 768; return (t >= 64 && t <= 768 && t % 64 == 0) ? t : 768; }();
  static const bool x6 = []() { const char* e = getenv("TRAJSDE_DECODE_FP32"); return !(e && atoi(e) != 0); }();
  // <=512 threads: the 256-VGPR build (no spills, 2 waves/SIMD); above: the 168-VGPR build (3 waves/SIMD)
  // every wave adds a 128-float slab (the step's time-conditioned biases) behind the weight image: as many waves as the 160 KB hold
  // (the bf16x6 build's image is 1.5x the fp16x3 one: 9 waves there)
  auto fit = [&](int image_floats) {
    int t = dthreads;
    while (t > 64 && (image_floats + (t / 64) * 128) * 4 > 160 * 1024) t -= 64;
    return t;
  };
#define TS_DECODE(X6, MAXT, IMG, OFF, DT)                                                                                      \
  TS_LAUNCH((k_sde_decode<X6, MAXT>), pick_grid(ntiles, (DT) / 64), (DT), (IMG::SIZE + ((DT) / 64) * 128) * 4, stream, blob + OFF, y0, rows, \
            future_steps, n_euler, step_table, out_table, min_scale, to_arg(noise), loc, state_bf16() ? 1 : 0)
  const int dt6 = fit(DecSdeL6::SIZE), dt1 = fit(DecSdeL::SIZE);
  if (x6 && dt6 <= 512) TS_DECODE(true, 512, DecSdeL6, DecBlob::SDE6, dt6);
  else if (x6) TS_DECODE(true, 768, DecSdeL6, DecBlob::SDE6, dt6);
  else if (dt1 <= 512) TS_DECODE(false, 512, DecSdeL, DecBlob::SDE, dt1);
  else TS_DECODE(false, 768, DecSdeL, DecBlob::SDE, dt1);
#undef TS_DECODE
  return TRAJSDE_OK;
}

int trajsde_sde_step(int32_t rows, const float* blob, const float* y_in, float* y_out, const float* e, int step,
                     const trajsde_noise* noise, void* stream_) {
  TS_REQUIRE(blob && y_in && y_out && e && rows > 0, "sde_step: bad argument");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t ntiles = (int64_t(rows) + 15) / 16;
  static const bool x6 = []() { const char* v = getenv("TRAJSDE_DECODE_FP32"); return !(v && atoi(v) != 0); }();
  if (x6)
    TS_LAUNCH(k_sde_step<true>, pick_grid(ntiles, 16), 1024, (DecSdeL6::LOC + 128 + 16 * SDE_STEP_SLOT) * 4, stream, blob + DecBlob::SDE6, y_in, y_out, int64_t(rows), e[1],
              e[2], e[3], e[4], step, to_arg(noise), state_bf16() ? 1 : 0);
  else
    TS_LAUNCH(k_sde_step<false>, pick_grid(ntiles, 16), 1024, DecSdeL::LOC * 4, stream, blob + DecBlob::SDE, y_in, y_out, int64_t(rows), e[1],
              e[2], e[3], e[4], step, to_arg(noise), state_bf16() ? 1 : 0);
  return TRAJSDE_OK;
}

}  // extern "C"
