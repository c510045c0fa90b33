// pack.hip -- weight packing: state_dict tensors -> the kernels' LDS images (layouts.hpp), plus the
// parameter-name tables of the C-ABI.  One recipe per stage is the single source of truth: run "dry" it
// yields the ordered parameter names (trajsde_param_name), run "wet" it launches the pack kernels.
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "bwd.hpp"
#include "common.hpp"
#include "layouts.hpp"
#include "range.hpp"

namespace tsde {

std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}
int fail(int code, const std::string& msg) {
  last_error_ref() = msg;
  return code;
}

std::atomic<int> g_state_bf16{0};
bool state_bf16() { return g_state_bf16.load(std::memory_order_relaxed) != 0; }

// registry of the per-translation-unit range flag words (range.hpp); function-local static: safe during static initialisation
std::vector<RangeReader>& range_readers() {
  static std::vector<RangeReader> v;
  return v;
}
void register_range_reader(RangeReader r) { range_readers().push_back(r); }

// Packing runs as launches over job tables passed BY VALUE in the kernel arguments (64 jobs per launch; jobs that
// accumulate onto another's output go in a later launch): a stage has a few hundred small tensors and is re-packed after
// every optimizer step, so per-tensor launches were ~700 launches per training step.  Element i of a job is handled by
// the device functions below.
struct PackJob {
  int kind, count, pass;
  int p0, p1, p2, p3, p4;
  const float *src, *src2, *src3;
  float* dst;
};
enum PackKind { PK_VEC = 0, PK_COL, PK_MAT, PK_MAT_PAD, PK_MATT, PK_MAT6, PK_MAT6_STACK2, PK_SPLIT, PK_LN2, PK_MAT6_CENTRED, PK_VEC_CENTRED, PK_AFFINE, PK_MAT32, PK_IN2F };
constexpr int PACK_JOBS_PER_LAUNCH = 48;        // 48 x 64 B: under the 4 KB kernel-argument limit
struct PackJobs {
  PackJob j[PACK_JOBS_PER_LAUNCH];
};

// dst[i] (+)= src[i]
// (`scale_bits`: the bit pattern of an fp32 factor applied to the packed element, 0 = none -- Packer::scale: a constant of the
//  consuming kernel's arithmetic folded into the image, e.g. the 2 / ln 2 of tanh(x) = 1 - 2 / (2^(x 2 / ln 2) + 1))
__device__ __forceinline__ float pack_scaled(float x, int scale_bits) { return scale_bits ? x * __int_as_float(scale_bits) : x; }
__device__ __forceinline__ void k_pack_vec(int i, const float* __restrict__ src, float* __restrict__ dst, int n, int accumulate, int scale_bits) {
  if (i < n) dst[i] = accumulate ? dst[i] + pack_scaled(src[i], scale_bits) : pack_scaled(src[i], scale_bits);
}
// dst[r] = src[r*ld + col]
__device__ __forceinline__ void k_pack_col(int i, const float* __restrict__ src, float* __restrict__ dst, int rows, int ld, int col, int scale_bits) {
  if (i < rows) dst[i] = pack_scaled(src[i * ld + col], scale_bits);
}
// MFMA fragment order: dst[((jo*JTI + q)*64 + lane)*4 + c] = W[16jo + (lane&15)][col0 + 16q + 4(lane>>4) + c]
__device__ __forceinline__ void k_pack_mat(int i, const float* __restrict__ src, float* __restrict__ dst, int jto, int jti, int ld, int col0) {
  if (i >= jto * jti * 256) return;
  const int c = i & 3, lane = (i >> 2) & 63, q = (i >> 8) % jti, jo = (i >> 8) / jti;
  dst[i] = src[(16 * jo + (lane & 15)) * ld + col0 + 16 * q + 4 * (lane >> 4) + c];
}

// k_pack_mat with the rows >= rows_valid written as zeros (a weight whose row count is not a multiple of 16)
__device__ __forceinline__ void k_pack_mat_pad(int i, const float* __restrict__ src, float* __restrict__ dst, int jto, int jti, int ld,
                                               int rows_valid) {
  if (i >= jto * jti * 256) return;
  const int c = i & 3, lane = (i >> 2) & 63, q = (i >> 8) % jti, jo = (i >> 8) / jti;
  const int row = 16 * jo + (lane & 15);
  dst[i] = row < rows_valid ? src[row * ld + 16 * q + 4 * (lane >> 4) + c] : 0.f;
}

// fragment image [jo < jto][q < jti][lane][4] of a TRANSPOSED weight: element [r][c] = W[c][col0 + r] (W row-major, ld)
__device__ __forceinline__ void k_pack_matT(int i, const float* __restrict__ src, float* __restrict__ dst, int jto, int jti, int ld,
                                            int col0, int rows_valid) {
  if (i >= jto * jti * 256) return;
  const int c = i & 3, lane = (i >> 2) & 63, q = (i >> 8) % jti, jo = (i >> 8) / jti;
  const int row_t = 16 * jo + (lane & 15), col_t = 16 * q + 4 * (lane >> 4) + c;      // element of W^T
  dst[i] = col_t < rows_valid ? src[int64_t(col_t) * ld + col0 + row_t] : 0.f;      // rows of W beyond rows_valid read as 0
}

// the split-precision pieces of one weight element (tile.hpp): fp16x3 = two fp16 planes, round to nearest (the
// second piece then carries the signed remainder); bf16x6 = three exact bf16 truncation pieces
__device__ __forceinline__ void store_split(float x, unsigned short* __restrict__ dst, int i, int /*per_plane*/) {
  // i = ((jo*ks + s)*64 + lane)*8 + j; the pieces of one (jo, s) block sit next to each other: [jo][s][piece][lane][8]
  const int blk = i >> 9, within = i & 511;
#if TSDE_SPLIT_H3
  range_note(fabsf(x), RS_WEIGHT);                        // |w| >= 65504 has no fp16 image
  const _Float16 h = _Float16(x);
  const _Float16 l = _Float16(x - float(h));
  dst[(blk * 2) * 512 + within] = __builtin_bit_cast(unsigned short, h);
  dst[(blk * 2 + 1) * 512 + within] = __builtin_bit_cast(unsigned short, l);
#else
  const unsigned M = 0xFFFF0000u;
  const float h = __uint_as_float(__float_as_uint(x) & M);
  const float r = x - h;
  const float m = __uint_as_float(__float_as_uint(r) & M);
  const float l = r - m;
  dst[(blk * 3) * 512 + within] = (unsigned short)(__float_as_uint(h) >> 16);
  dst[(blk * 3 + 1) * 512 + within] = (unsigned short)(__float_as_uint(m) >> 16);
  dst[(blk * 3 + 2) * 512 + within] = (unsigned short)(__float_as_uint(l) >> 16);
#endif
}

// split-precision pieces: dst (16-bit) [jo][s][piece][lane][8]; element j of lane (i,g) in k-step s is the piece-th
// truncation piece of W[16jo + i][col0 + 32s + 16(j>>2) + 4g + (j&3)]   (tile.hpp linear_acc_x6)
__device__ __forceinline__ void k_pack_mat6(int i, const float* __restrict__ src, unsigned short* __restrict__ dst, int jto, int ks,
                                            int ld, int col0, int scale_bits) {
  const int per_plane = jto * ks * 512;
  if (i >= per_plane) return;
  const int j = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % ks, jo = (i >> 9) / ks;
  const float x = pack_scaled(src[(16 * jo + (lane & 15)) * ld + col0 + 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)], scale_bits);
  store_split(x, dst, i, per_plane);
}

// split-precision image of a (possibly transposed, possibly row-padded) matrix, the layout of k_pack_mat6:
//   plain:      element [r][c] = W[r][col0 + c], rows >= rows_valid read as 0
//   transposed: element [r][c] = W[c][col0 + r], rows of W >= rows_valid read as 0
__device__ __forceinline__ void k_pack_split(int i, const float* __restrict__ src, unsigned short* __restrict__ dst, int jto, int ks,
                                             int ld, int col0, int rows_valid, int transposed) {
  const int per_plane = jto * ks * 512;
  if (i >= per_plane) return;
  const int j = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % ks, jo = (i >> 9) / ks;
  const int r = 16 * jo + (lane & 15), c = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  float x;
  if (transposed) x = c < rows_valid ? src[int64_t(c) * ld + col0 + r] : 0.f;
  else x = r < rows_valid ? src[int64_t(r) * ld + col0 + c] : 0.f;
  store_split(x, dst, i, per_plane);
}

// lin_k | lin_v as ONE 128-row split-precision matrix: planes are [plane][jo 0..7][s][lane][8]
// colscale (or null): the matrix times diag(colscale) -- a LayerNorm's gamma folded into the layer that consumes its output
__device__ __forceinline__ void k_pack_mat6_stack2(int i, const float* __restrict__ wk, const float* __restrict__ wv,
                                                   const float* __restrict__ colscale, unsigned short* __restrict__ dst) {
  const int per_plane = 8 * 2 * 512;
  if (i >= per_plane) return;
  const int j = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % 2, jo = (i >> 9) / 2;
  const float* src = jo < 4 ? wk : wv;
  const int col = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  float x = src[(16 * (jo & 3) + (lane & 15)) * 64 + col];
  if (colscale != nullptr) x *= colscale[col];
  store_split(x, dst, i, per_plane);
}

// 64x64 weight with the mean over its OUTPUT features removed from every column: W_c = W - 1 (1^T W) / 64.  A LayerNorm
// that follows y = W x + b only ever sees y - mean(y) = W_c x + (b - mean(b)), so the kernel that multiplies by W_c gets the
// centred row straight out of the matrix cores (tile.hpp centred_rstd)
__device__ __forceinline__ void k_pack_mat6_centred(int i, const float* __restrict__ src, unsigned short* __restrict__ dst) {
  const int per_plane = 4 * 2 * 512;
  if (i >= per_plane) return;
  const int j = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % 2, jo = (i >> 9) / 2;
  const int col = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  double m = 0;
  for (int r = 0; r < 64; ++r) m += src[r * 64 + col];
  const float x = float(double(src[(16 * jo + (lane & 15)) * 64 + col]) - m / 64);
  store_split(x, dst, i, per_plane);
}
// Split-precision image of a (32 nblk) x 64 matrix for the 32x32x16 matrix instruction (edge32.hip): planes
// [jo < nblk][k-step s < 4][piece][lane][8], lane (m = lane & 31, hh = lane >> 5) holding row 32 jo + m and, as k slot 8 hh + jj
// of step s, input feature 32 (s >> 1) + 8 (2 (s & 1) + (jj >> 2)) + 4 hh + (jj & 3).  Rows 64.. come from W1 (lin_k | lin_v
// stacked).  centred: the mean over the 64 rows of the matrix a row comes from is removed from every column; colscale (or
// null): columns scaled (a LayerNorm's gamma folded into its consumer).
__device__ __forceinline__ void k_pack_mat32(int i, const float* __restrict__ W0, const float* __restrict__ W1,
                                             const float* __restrict__ colscale, unsigned short* __restrict__ dst, int nblk, int centred) {
  const int per_plane = nblk * 4 * 512;
  if (i >= per_plane) return;
  const int jj = i & 7, lane = (i >> 3) & 63, s = (i >> 9) & 3, jo = i >> 11;
  const int m = lane & 31, hh = lane >> 5, row = 32 * jo + m;
  const float* src = row < 64 ? W0 : W1;
  const int r = row & 63;
  const int col = 32 * (s >> 1) + 8 * (2 * (s & 1) + (jj >> 2)) + 4 * hh + (jj & 3);
  double x = src[r * 64 + col];
  if (centred) {
    double mu = 0;
    for (int k = 0; k < 64; ++k) mu += src[k * 64 + col];
    x -= mu / 64;
  }
  float xf = float(x);
  if (colscale != nullptr) xf *= colscale[col];
  store_split(xf, dst, i, per_plane);
}
// dst = (a [+ b]) - mean(a [+ b]), 64 elements
__device__ __forceinline__ void k_pack_vec_centred(int i, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst) {
  if (i >= 64) return;
  double m = 0;
  for (int f = 0; f < 64; ++f) m += double(a[f]) + (b ? double(b[f]) : 0.0);
  dst[i] = float(double(a[i]) + (b ? double(b[i]) : 0.0) - m / 64);
}
// dst = W beta + bias (W [64][64] row-major): what a linear layer makes of the LayerNorm's beta in front of it
__device__ __forceinline__ void k_pack_affine(int i, const float* __restrict__ W, const float* __restrict__ beta,
                                              const float* __restrict__ bias, float* __restrict__ dst) {
  if (i >= 64) return;
  double acc = bias[i];
  for (int c = 0; c < 64; ++c) acc += double(W[i * 64 + c]) * beta[c];
  dst[i] = float(acc);
}

// closed-form Linear(2,64)->LayerNorm block (layouts.hpp In2L) from W [64][2], b [64], gamma [64]: thread f < 64 writes
// the gamma-scaled centred columns, thread 0 the Cholesky factor of their Gram matrix / 64; sums in double
__device__ __forceinline__ void k_pack_ln2(int i, const float* __restrict__ W, const float* __restrict__ b,
                                           const float* __restrict__ gamma, float* __restrict__ dst) {
  if (i >= 64) return;
  double m0 = 0, m1 = 0, mb = 0;
  for (int f = 0; f < 64; ++f) { m0 += W[2 * f]; m1 += W[2 * f + 1]; mb += b[f]; }
  m0 /= 64; m1 /= 64; mb /= 64;
  dst[In2L::GW0 + i] = float(double(gamma[i]) * (W[2 * i] - m0));
  dst[In2L::GW1 + i] = float(double(gamma[i]) * (W[2 * i + 1] - m1));
  dst[In2L::GB + i] = float(double(gamma[i]) * (b[i] - mb));
  if (i == 0) {
    double g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
    for (int f = 0; f < 64; ++f) {
      const double a = W[2 * f] - m0, c = W[2 * f + 1] - m1, d = b[f] - mb;
      g00 += a * a; g01 += a * c; g02 += a * d; g11 += c * c; g12 += c * d; g22 += d * d;
    }
    g00 /= 64; g01 /= 64; g02 /= 64; g11 /= 64; g12 /= 64; g22 /= 64;
    // G = L L^T (lower), rank-deficient columns (e.g. a zero bias) give zero pivots: their rows of L are zero
    const double l00 = g00 > 0 ? sqrt(g00) : 0, l10 = l00 > 0 ? g01 / l00 : 0, l20 = l00 > 0 ? g02 / l00 : 0;
    const double p11 = g11 - l10 * l10, l11 = p11 > 0 ? sqrt(p11) : 0, l21 = l11 > 0 ? (g12 - l20 * l10) / l11 : 0;
    const double p22 = g22 - l20 * l20 - l21 * l21, l22 = p22 > 0 ? sqrt(p22) : 0;
    // |L^T z|^2 with z = (x0, x1, 1):  (l00 x0 + l10 x1 + l20)^2 + (l11 x1 + l21)^2 + l22^2
    float* ch = dst + In2L::CH;
    ch[0] = float(l00); ch[1] = float(l10); ch[2] = float(l20); ch[3] = float(l11); ch[4] = float(l21); ch[5] = float(l22);
    ch[6] = ch[7] = 0.f;
  }
}

// matrix-core fragments of a closed-form Linear(2,64)->LayerNorm block (layouts.hpp IN2F) from the In2L block `in2` packed by
// an earlier pass and the LayerNorm's beta; dst (16-bit) [jo][lane][8]
__device__ __forceinline__ void k_pack_in2f(int i, const float* __restrict__ in2, const float* __restrict__ beta,
                                            unsigned short* __restrict__ dst) {
  if (i >= 4 * 512) return;
  const int j = i & 7, lane = (i >> 3) & 63, jo = i >> 9;
  const int f = 16 * jo + (lane & 15), g = lane >> 4;
  const float term[4] = {in2[In2L::GW0 + f], in2[In2L::GW1 + f], in2[In2L::GB + f], beta[f]};
  float x = 0.f;
  bool low = false;
  if (g == 0) {                                            // GW0_h GW1_h GW0_h GW1_h GB_h GB_h be_h be_l
    x = term[j < 4 ? (j & 1) : j < 6 ? 2 : 3];
    low = j == 7;
  } else if (g == 1 && (j < 2 || j == 4)) {                // GW0_l GW1_l 0 0 GB_l 0 0 0
    x = term[j < 2 ? j : 2];
    low = true;
  }
#if TSDE_SPLIT_H3
  range_note(fabsf(x), RS_WEIGHT);
#endif
  const _Float16 h = _Float16(x);
  const _Float16 l = _Float16(x - float(h));
  dst[i] = __builtin_bit_cast(unsigned short, low ? l : h);
}

// element i of job J
__device__ __forceinline__ void pack_element(const PackJob& J, int i) {
  switch (J.kind) {
    case PK_VEC: k_pack_vec(i, J.src, J.dst, J.count, J.p0, J.p1); break;
    case PK_COL: k_pack_col(i, J.src, J.dst, J.count, J.p0, J.p1, J.p2); break;
    case PK_MAT: k_pack_mat(i, J.src, J.dst, J.p0, J.p1, J.p2, J.p3); break;
    case PK_MAT_PAD: k_pack_mat_pad(i, J.src, J.dst, J.p0, J.p1, J.p2, J.p3); break;
    case PK_MATT: k_pack_matT(i, J.src, J.dst, J.p0, J.p1, J.p2, J.p3, J.p4); break;
    case PK_MAT6: k_pack_mat6(i, J.src, reinterpret_cast<unsigned short*>(J.dst), J.p0, J.p1, J.p2, J.p3, J.p4); break;
    case PK_MAT6_STACK2: k_pack_mat6_stack2(i, J.src, J.src2, J.src3, reinterpret_cast<unsigned short*>(J.dst)); break;
    case PK_MAT6_CENTRED: k_pack_mat6_centred(i, J.src, reinterpret_cast<unsigned short*>(J.dst)); break;
    case PK_VEC_CENTRED: k_pack_vec_centred(i, J.src, J.src2, J.dst); break;
    case PK_AFFINE: k_pack_affine(i, J.src, J.src2, J.src3, J.dst); break;
    case PK_MAT32: k_pack_mat32(i, J.src, J.src2, J.src3, reinterpret_cast<unsigned short*>(J.dst), J.p0, J.p1); break;
    case PK_LN2: k_pack_ln2(i, J.src, J.src2, J.src3, J.dst); break;
    case PK_IN2F: k_pack_in2f(i, J.src, J.src2, reinterpret_cast<unsigned short*>(J.dst)); break;
    case PK_SPLIT: k_pack_split(i, J.src, reinterpret_cast<unsigned short*>(J.dst), J.p0, J.p1, J.p2, J.p3, J.p4 & 0x3FFFFFFF, J.p4 >> 30); break;
  }
}

// one thread per element; blockIdx.y = job
__global__ __launch_bounds__(256) void k_pack_jobs(const PackJobs jobs) {
  const PackJob& J = jobs.j[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= J.count) return;
  pack_element(J, i);
}

// The same over a job table in DEVICE memory (trajsde_pack_weights_many: the tables of all the stages of a training step, uploaded
// once and re-used while the parameter and blob addresses stay): jobs [first, first + gridDim.y).
__global__ __launch_bounds__(256) void k_pack_jobs_table(const PackJob* __restrict__ table, int first) {
  const PackJob J = table[first + blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= J.count) return;
  pack_element(J, i);
}
// blobs to zero: job q = (dst, count); gridDim.x workgroups stride over a blob in float4 steps
__global__ __launch_bounds__(256) void k_pack_zero_table(const PackJob* __restrict__ table) {
  const PackJob J = table[blockIdx.y];
  float* __restrict__ d = J.dst;
  const int n4 = J.count >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) reinterpret_cast<float4*>(d)[i] = float4{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && int(threadIdx.x) < (J.count & 3)) d[4 * n4 + threadIdx.x] = 0.f;
}

struct Packer {
  bool dry;
  float scale = 0.f;                       // != 0: vec / col / mat6 multiply what they pack by it (pack_scaled); recipes set and clear it
  int scale_bits() const { return scale != 0.f ? __builtin_bit_cast(int, scale) : 0; }
  std::vector<PackJob> jobs;               // wet: filled by the recipe, launched once by trajsde_pack_weights
  void emit(int kind, int count, const float* s, float* d, int p0 = 0, int p1 = 0, int p2 = 0, int p3 = 0, int p4 = 0,
            const float* s2 = nullptr, int pass = 0) {
    jobs.push_back(PackJob{kind, count, pass, p0, p1, p2, p3, p4, s, s2, nullptr, d});
  }
  // raw-pointer forms for recipes that slice a parameter themselves
  void vec_raw(const float* s, float* d, int count) { emit(PK_VEC, count, s, d, 0); }
  // general matrix images: fp32 fragment order (bf16x6 build) or, in the fp16x3 build, split-precision planes of the
  // same size (tile.hpp linear_acc); jti counts 16-column input tiles and is always even
  void split_raw(const float* s, float* d, int jto, int jti, int ld, int col0, int rows_valid, bool transposed) {
    const int rv = rows_valid > 0x3FFFFFFF ? 0x3FFFFFFF : rows_valid;
    emit(PK_SPLIT, jto * (jti / 2) * 512, s, d, jto, jti / 2, ld, col0, rv | (transposed ? 1 << 30 : 0));
  }
  void mat_raw(const float* s, float* d, int jto, int jti, int ld, int col0) {
    if (TSDE_SPLIT_H3) split_raw(s, d, jto, jti, ld, col0, 1 << 30, false);
    else emit(PK_MAT, jto * jti * 256, s, d, jto, jti, ld, col0);
  }
  void matT_raw(const float* s, float* d, int jto, int jti, int ld, int col0, int rows_valid) {
    if (TSDE_SPLIT_H3) split_raw(s, d, jto, jti, ld, col0, rows_valid, true);
    else emit(PK_MATT, jto * jti * 256, s, d, jto, jti, ld, col0, rows_valid);
  }
  void mat6_raw(const float* s, float* d, int jto, int ks, int ld, int col0) { emit(PK_MAT6, jto * ks * 512, s, d, jto, ks, ld, col0, scale_bits()); }
  std::vector<std::string> names;          // dry: collected in order of first use
  const float* const* params = nullptr;    // wet
  float* blob = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  int index(const std::string& n) {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == n) return int(i);
    names.push_back(n);
    return int(names.size()) - 1;
  }
  const float* src(const std::string& n) {
    const int i = index(n);
    return dry ? nullptr : params[i];
  }
  void vec(const std::string& n, int dst, int count, bool accumulate = false) {
    const float* s = src(n);
    if (dry) return;
    emit(PK_VEC, count, s, blob + dst, accumulate ? 1 : 0, scale_bits(), 0, 0, 0, nullptr, accumulate ? 1 : 0);
  }
  void col(const std::string& n, int dst, int rows, int ld, int c) {
    const float* s = src(n);
    if (dry) return;
    emit(PK_COL, rows, s, blob + dst, ld, c, scale_bits());
  }
  // rows x cols sub-matrix starting at column col0 of a row-major [rows x ld] weight
  void mat(const std::string& n, int dst, int rows, int cols, int ld, int col0 = 0) {
    const float* s = src(n);
    if (dry) return;
    const int jto = rows / 16, jti = cols / 16;
    mat_raw(s, blob + dst, jto, jti, ld, col0);
  }
  void mat_pad(const std::string& n, int dst, int jto, int jti, int ld, int rows_valid) {
    const float* s = src(n);
    if (dry) return;
    if (TSDE_SPLIT_H3) split_raw(s, blob + dst, jto, jti, ld, 0, rows_valid, false);
    else emit(PK_MAT_PAD, jto * jti * 256, s, blob + dst, jto, jti, ld, rows_valid);
  }
  // transposed image of the [16*jti x 16*jto] block of W starting at column col0 (default: a 64x64 block)
  void matT(const std::string& n, int dst, int ld, int col0 = 0, int jto = 4, int jti = 4, int rows_valid = 1 << 30) {
    const float* s = src(n);
    if (dry) return;
    matT_raw(s, blob + dst, jto, jti, ld, col0, rows_valid);
  }
  void mat6(const std::string& n, int dst, int rows, int cols, int ld, int col0 = 0) {
    const float* s = src(n);
    if (dry) return;
    const int jto = rows / 16, ks = cols / 32;
    mat6_raw(s, blob + dst, jto, ks, ld, col0);
  }
  // closed-form block of Sequential(Linear(2,64), LayerNorm(64)) `p.0`, `p.1`
  void ln2(const std::string& p, int dst) {
    const float* w = src(p + ".0.weight");
    const float* b = src(p + ".0.bias");
    const float* g = src(p + ".1.weight");
    if (dry) return;
    emit(PK_LN2, 64, w, blob + dst, 0, 0, 0, 0, 0, b);
    jobs.back().src3 = g;
  }
  // ... and its matrix-core fragments (layouts.hpp IN2F) from that block at `in2` and the LayerNorm's beta: second pass
  void in2f(const std::string& p, int in2, int dst) {
    const float* be = src(p + ".1.bias");
    if (dry) return;
    emit(PK_IN2F, 4 * 512, blob + in2, blob + dst, 0, 0, 0, 0, 0, be, /*pass=*/1);
  }
  void lin(const std::string& p, int w, int b, int rows = 64, int cols = 64) {
    mat(p + ".weight", w, rows, cols, cols);
    vec(p + ".bias", b, rows);
  }
  void ln(const std::string& p, int g, int e) {
    vec(p + ".weight", g, 64);
    vec(p + ".bias", e, 64);
  }
};

static void recipe_edge_embed(Packer& P, const std::string& p, int base) {   // MultipleInputEmbedding
  using E = EdgeL;
  P.vec(p + ".module_list.0.0.weight", base + E::A_W0, 128);
  P.vec(p + ".module_list.0.0.bias", base + E::A_B0, 64);
  P.ln(p + ".module_list.0.1", base + E::A_G, base + E::A_E);
  P.vec(p + ".module_list.1.0.weight", base + E::B_W0, 128);
  P.vec(p + ".module_list.1.0.bias", base + E::B_B0, 64);
  P.ln(p + ".module_list.1.1", base + E::B_G, base + E::B_E);
  P.mat(p + ".module_list.0.3.weight", base + E::WA3, 64, 64, 64);
  P.mat(p + ".module_list.1.3.weight", base + E::WB3, 64, 64, 64);
  P.vec(p + ".module_list.0.3.bias", base + E::B3, 64);
  P.vec(p + ".module_list.1.3.bias", base + E::B3, 64, /*accumulate=*/true);
  P.ln(p + ".aggr_embed.0", base + E::AG0, base + E::AE0);
  P.lin(p + ".aggr_embed.2", base + E::W2, base + E::B2);
  P.ln(p + ".aggr_embed.3", base + E::AG3, base + E::AE3);
}
static void recipe_edge_embed6(Packer& P, const std::string& p, int base) {   // split-precision twin of the above
  using E = EdgeL6;
  P.vec(p + ".module_list.0.0.weight", base + E::A_W0, 128);
  P.vec(p + ".module_list.0.0.bias", base + E::A_B0, 64);
  P.ln(p + ".module_list.0.1", base + E::A_G, base + E::A_E);
  P.vec(p + ".module_list.1.0.weight", base + E::B_W0, 128);
  P.vec(p + ".module_list.1.0.bias", base + E::B_B0, 64);
  P.ln(p + ".module_list.1.1", base + E::B_G, base + E::B_E);
  P.mat6(p + ".module_list.0.3.weight", base + E::WA3, 64, 64, 64);
  P.mat6(p + ".module_list.1.3.weight", base + E::WB3, 64, 64, 64);
  P.vec(p + ".module_list.0.3.bias", base + E::B3, 64);
  P.vec(p + ".module_list.1.3.bias", base + E::B3, 64, /*accumulate=*/true);
  P.ln2(p + ".module_list.0", base + E::A_C);
  P.ln2(p + ".module_list.1", base + E::B_C);
  P.in2f(p + ".module_list.0", base + E::A_C, base + E::A_F);
  P.in2f(p + ".module_list.1", base + E::B_C, base + E::B_F);
  P.ln(p + ".aggr_embed.0", base + E::AG0, base + E::AE0);
  P.mat6(p + ".aggr_embed.2.weight", base + E::W2, 64, 64, 64);
  P.vec(p + ".aggr_embed.2.bias", base + E::B2, 64);
  P.ln(p + ".aggr_embed.3", base + E::AG3, base + E::AE3);
}
// image of the fused edge-attention kernel (layouts.hpp EdgeL6F): the embedding of recipe_edge_embed6 with the LayerNorm
// algebra folded into the matrices, lin_k | lin_v scaled by the last LayerNorm's gamma, no k / v bias rows
static void recipe_edge_fused(Packer& P, const std::string& p, const std::string& att, int base, bool tile32 = false) {
  using E = EdgeL6F;
  const float* wa = P.src(p + ".module_list.0.3.weight");
  const float* wb = P.src(p + ".module_list.1.3.weight");
  const float* ba = P.src(p + ".module_list.0.3.bias");
  const float* bb = P.src(p + ".module_list.1.3.bias");
  const float* w2 = P.src(p + ".aggr_embed.2.weight");
  const float* b2 = P.src(p + ".aggr_embed.2.bias");
  const float* g3 = P.src(p + ".aggr_embed.3.weight");
  const float* e3 = P.src(p + ".aggr_embed.3.bias");
  const float* wk = P.src(att + ".lin_k.weight");
  const float* wv = P.src(att + ".lin_v.weight");
  const float* bk = P.src(att + ".lin_k.bias");
  const float* bv = P.src(att + ".lin_v.bias");
  P.vec(p + ".module_list.0.1.bias", base + E::A_E, 64);
  P.vec(p + ".module_list.1.1.bias", base + E::B_E, 64);
  P.ln2(p + ".module_list.0", base + E::A_C);
  P.ln2(p + ".module_list.1", base + E::B_C);
  P.in2f(p + ".module_list.0", base + E::A_C, base + E::A_F);
  P.in2f(p + ".module_list.1", base + E::B_C, base + E::B_F);
  P.ln(p + ".aggr_embed.0", base + E::AG0, base + E::AE0);
  P.ln(p + ".aggr_embed.3", base + E::AG3, base + E::AE3);
  if (P.dry) return;
  float* d = P.blob + base;
  if (tile32) {                                            // the same matrices in the fragment order of the 32x32x16 instruction
    P.emit(PK_MAT32, 2 * 4 * 512, wa, d + E::WA3, 2, 1);
    P.emit(PK_MAT32, 2 * 4 * 512, wb, d + E::WB3, 2, 1);
    P.emit(PK_MAT32, 2 * 4 * 512, w2, d + E::W2, 2, 1);
    P.emit(PK_MAT32, 4 * 4 * 512, wk, d + E::WKV, 4, 0, 0, 0, 0, wv);
    P.jobs.back().src3 = g3;
  } else {
    P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, wa, d + E::WA3);
    P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, wb, d + E::WB3);
    P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, w2, d + E::W2);
    P.emit(PK_MAT6_STACK2, 8 * 2 * 512, wk, d + E::WKV, 0, 0, 0, 0, 0, wv);
    P.jobs.back().src3 = g3;
  }
  P.emit(PK_VEC_CENTRED, 64, ba, d + E::B3, 0, 0, 0, 0, 0, bb);
  P.emit(PK_VEC_CENTRED, 64, b2, d + E::B2);
  P.emit(PK_AFFINE, 64, wk, d + E::CK, 0, 0, 0, 0, 0, e3);
  P.jobs.back().src3 = bk;
  P.emit(PK_AFFINE, 64, wv, d + E::CV, 0, 0, 0, 0, 0, e3);
  P.jobs.back().src3 = bv;
}
// the embedding half of recipe_edge_fused on the EdgeL6G image (no key / value part): the global interactor's rel_embed
static void recipe_edge_fused_embed(Packer& P, const std::string& p, int base) {
  using E = EdgeL6G;
  const float* wa = P.src(p + ".module_list.0.3.weight");
  const float* wb = P.src(p + ".module_list.1.3.weight");
  const float* ba = P.src(p + ".module_list.0.3.bias");
  const float* bb = P.src(p + ".module_list.1.3.bias");
  const float* w2 = P.src(p + ".aggr_embed.2.weight");
  const float* b2 = P.src(p + ".aggr_embed.2.bias");
  P.vec(p + ".module_list.0.1.bias", base + E::A_E, 64);
  P.vec(p + ".module_list.1.1.bias", base + E::B_E, 64);
  P.ln2(p + ".module_list.0", base + E::A_C);
  P.ln2(p + ".module_list.1", base + E::B_C);
  P.in2f(p + ".module_list.0", base + E::A_C, base + E::A_F);
  P.in2f(p + ".module_list.1", base + E::B_C, base + E::B_F);
  P.ln(p + ".aggr_embed.0", base + E::AG0, base + E::AE0);
  P.ln(p + ".aggr_embed.3", base + E::AG3, base + E::AE3);
  if (P.dry) return;
  float* d = P.blob + base;
  P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, wa, d + E::WA3);
  P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, wb, d + E::WB3);
  P.emit(PK_MAT6_CENTRED, 4 * 2 * 512, w2, d + E::W2);
  P.emit(PK_VEC_CENTRED, 64, ba, d + E::B3, 0, 0, 0, 0, 0, bb);
  P.emit(PK_VEC_CENTRED, 64, b2, d + E::B2);
}
static void pack_kv6(Packer& P, const std::string& k, const std::string& v, int w, int b) {
  const float* wk = P.src(k + ".weight");
  const float* wv = P.src(v + ".weight");
  if (!P.dry) P.emit(PK_MAT6_STACK2, 8 * 2 * 512, wk, P.blob + w, 0, 0, 0, 0, 0, wv);
  P.vec(k + ".bias", b, 64);
  P.vec(v + ".bias", b + 64, 64);
}

static void recipe_upd_ffn(Packer& P, const std::string& p, int upd, int ffn) {
  P.lin(p + ".lin_ih", upd + UpdL::WIH, upd + UpdL::BIH);
  P.lin(p + ".lin_hh", upd + UpdL::WHH, upd + UpdL::BHH);
  P.lin(p + ".lin_self", upd + UpdL::WSELF, upd + UpdL::BSELF);
  P.lin(p + ".out_proj", upd + UpdL::WOUT, upd + UpdL::BOUT);
  P.ln(p + ".norm2", upd + UpdL::N2G, upd + UpdL::N2B);
  P.lin(p + ".mlp.0", ffn + FfnL::W1, ffn + FfnL::B1, 256, 64);
  P.lin(p + ".mlp.3", ffn + FfnL::W2, ffn + FfnL::B2, 64, 256);
}
static void recipe_upd_ffn6(Packer& P, const std::string& p, int upd, int ffn) {
  using U = UpdL6;
  P.mat6(p + ".lin_ih.weight", upd + U::WIH, 64, 64, 64);
  P.vec(p + ".lin_ih.bias", upd + U::BIH, 64);
  P.mat6(p + ".lin_hh.weight", upd + U::WHH, 64, 64, 64);
  P.vec(p + ".lin_hh.bias", upd + U::BHH, 64);
  P.mat6(p + ".lin_self.weight", upd + U::WSELF, 64, 64, 64);
  P.vec(p + ".lin_self.bias", upd + U::BSELF, 64);
  P.mat6(p + ".out_proj.weight", upd + U::WOUT, 64, 64, 64);
  P.vec(p + ".out_proj.bias", upd + U::BOUT, 64);
  P.ln(p + ".norm2", upd + U::N2G, upd + U::N2B);
  // FFN halves: hidden units [128*hf, 128*hf+128): rows of mlp.0, columns of mlp.3
  const float* w1 = P.src(p + ".mlp.0.weight");     // [256, 64]
  const float* b1 = P.src(p + ".mlp.0.bias");
  const float* w2 = P.src(p + ".mlp.3.weight");     // [64, 256]
  const float* b2 = P.src(p + ".mlp.3.bias");
  if (!P.dry)
    for (int hf = 0; hf < 2; ++hf) {
      const int base = ffn + hf * FfnL6::HALF;
      P.mat6_raw(w1 + hf * 128 * 64, P.blob + base + FfnL6::W1, 8, 2, 64, 0);
      P.vec_raw(b1 + hf * 128, P.blob + base + FfnL6::B1, 128);
      P.mat6_raw(w2, P.blob + base + FfnL6::W2, 4, 4, 256, hf * 128);
      P.vec_raw(b2, P.blob + base + FfnL6::B2, 64);
    }
}

static void recipe_drift(Packer& P, const std::string& p, int base) {
  using L = DriftL;
  P.mat(p + ".net.0.weight", base + L::W0, 64, 64, 66, 0);
  P.col(p + ".net.0.weight", base + L::WS, 64, 66, 64);
  P.col(p + ".net.0.weight", base + L::WC, 64, 66, 65);
  P.vec(p + ".net.0.bias", base + L::B0, 64);
  P.lin(p + ".net.2", base + L::W2, base + L::B2);
  P.lin(p + ".net.4", base + L::W4, base + L::B4);
}
static void recipe_diff(Packer& P, const std::string& p, int base) {
  using L = DiffL;
  P.mat(p + ".net.0.weight", base + L::W0, 64, 64, 66, 0);
  P.col(p + ".net.0.weight", base + L::WS, 64, 66, 64);
  P.col(p + ".net.0.weight", base + L::WC, 64, 66, 65);
  P.vec(p + ".net.0.bias", base + L::B0, 64);
  P.lin(p + ".net.2", base + L::W2, base + L::B2);
  P.vec(p + ".net.4.weight", base + L::W4, 64);
  P.vec(p + ".net.4.bias", base + L::B4, 1);
}

static void recipe_drift6(Packer& P, const std::string& p, int base) {
  using L = DriftL6;
  P.mat6(p + ".net.0.weight", base + L::W0, 64, 64, 66, 0);
  P.col(p + ".net.0.weight", base + L::WS, 64, 66, 64);
  P.col(p + ".net.0.weight", base + L::WC, 64, 66, 65);
  P.vec(p + ".net.0.bias", base + L::B0, 64);
  P.mat6(p + ".net.2.weight", base + L::W2, 64, 64, 64);
  P.vec(p + ".net.2.bias", base + L::B2, 64);
  P.mat6(p + ".net.4.weight", base + L::W4, 64, 64, 64);
  P.vec(p + ".net.4.bias", base + L::B4, 64);
}
static void recipe_diff6(Packer& P, const std::string& p, int base) {
  using L = DiffL6;
  P.mat6(p + ".net.0.weight", base + L::W0, 64, 64, 66, 0);
  P.col(p + ".net.0.weight", base + L::WS, 64, 66, 64);
  P.col(p + ".net.0.weight", base + L::WC, 64, 66, 65);
  P.vec(p + ".net.0.bias", base + L::B0, 64);
  P.mat6(p + ".net.2.weight", base + L::W2, 64, 64, 64);
  P.vec(p + ".net.2.bias", base + L::B2, 64);
  P.vec(p + ".net.4.weight", base + L::W4, 64);
  P.vec(p + ".net.4.bias", base + L::B4, 1);
}

// AAEncoder + ALEncoder images (shared by the SDE encoder and the vanilla one)
static void recipe_encoder_attention(Packer& P) {
  using B = EncBlob;
  {  // AAEncoder node path
    using L = AaCenterL;
    const std::string c = "aa_encoder.center_embed.embed.";
    P.vec(c + "0.weight", B::AA_CENTER + L::W0, 128);
    P.vec(c + "0.bias", B::AA_CENTER + L::B0, 64);
    P.ln(c + "1", B::AA_CENTER + L::G1, B::AA_CENTER + L::E1);
    P.lin(c + "3", B::AA_CENTER + L::W3, B::AA_CENTER + L::B3);
    P.ln(c + "4", B::AA_CENTER + L::G4, B::AA_CENTER + L::E4);
    P.lin(c + "6", B::AA_CENTER + L::W6, B::AA_CENTER + L::B6);
    P.ln(c + "7", B::AA_CENTER + L::G7, B::AA_CENTER + L::E7);
    P.vec("aa_encoder.bos_token", B::AA_CENTER + L::BOS, 21 * 64);
    P.ln("aa_encoder.norm1", B::AA_CENTER + L::N1G, B::AA_CENTER + L::N1B);
    P.lin("aa_encoder.lin_q", B::AA_CENTER + L::WQ, B::AA_CENTER + L::BQ);
  }
  recipe_edge_embed(P, "aa_encoder.nbr_embed", B::AA_EDGE);
  P.mat("aa_encoder.lin_k.weight", B::AA_EDGE + EdgeL::WKV, 64, 64, 64);
  P.mat("aa_encoder.lin_v.weight", B::AA_EDGE + EdgeL::WKV + MAT64, 64, 64, 64);
  P.vec("aa_encoder.lin_k.bias", B::AA_EDGE + EdgeL::BKV, 64);
  P.vec("aa_encoder.lin_v.bias", B::AA_EDGE + EdgeL::BKV + 64, 64);
  recipe_upd_ffn(P, "aa_encoder", B::AA_UPD, B::AA_FFN);
  P.ln("al_encoder.norm1", B::AL_Q + NodeProjL<1>::N1G, B::AL_Q + NodeProjL<1>::N1B);
  P.lin("al_encoder.lin_q", B::AL_Q + NodeProjL<1>::W, B::AL_Q + NodeProjL<1>::B);
  recipe_edge_embed(P, "al_encoder.lane_embed", B::AL_EDGE);
  P.mat("al_encoder.lin_k.weight", B::AL_EDGE + EdgeL::WKV, 64, 64, 64);
  P.mat("al_encoder.lin_v.weight", B::AL_EDGE + EdgeL::WKV + MAT64, 64, 64, 64);
  P.vec("al_encoder.lin_k.bias", B::AL_EDGE + EdgeL::BKV, 64);
  P.vec("al_encoder.lin_v.bias", B::AL_EDGE + EdgeL::BKV + 64, 64);
  recipe_upd_ffn(P, "al_encoder", B::AL_UPD, B::AL_FFN);
  recipe_edge_embed6(P, "aa_encoder.nbr_embed", B::AA_EDGE6);
  pack_kv6(P, "aa_encoder.lin_k", "aa_encoder.lin_v", B::AA_EDGE6 + EdgeL6::WKV, B::AA_EDGE6 + EdgeL6::BKV);
  recipe_edge_embed6(P, "al_encoder.lane_embed", B::AL_EDGE6);
  pack_kv6(P, "al_encoder.lin_k", "al_encoder.lin_v", B::AL_EDGE6 + EdgeL6::WKV, B::AL_EDGE6 + EdgeL6::BKV);
  recipe_edge_fused(P, "aa_encoder.nbr_embed", "aa_encoder", B::AA_EDGE6F);
  recipe_edge_fused(P, "al_encoder.lane_embed", "al_encoder", B::AL_EDGE6F);
  recipe_edge_fused(P, "aa_encoder.nbr_embed", "aa_encoder", B::AA_EDGE6F + EdgeL6F::SIZE, true);
  recipe_edge_fused(P, "al_encoder.lane_embed", "al_encoder", B::AL_EDGE6F + EdgeL6F::SIZE, true);
  recipe_upd_ffn6(P, "aa_encoder", B::AA_UPD6, B::AA_FFN6);
  recipe_upd_ffn6(P, "al_encoder", B::AL_UPD6, B::AL_FFN6);
  // training path: wave-per-target attention over stored embedding rows (k_global_attn<.., NODE = false> and its backward)
  const char* enc[2] = {"aa_encoder", "al_encoder"};
  const int at[2] = {B::AA_ATTN, B::AL_ATTN};
  for (int i = 0; i < 2; ++i) {
    const std::string e = enc[i];
    P.vec(e + ".lin_k.weight", at[i] + GAttnL::WKE, MAT64);
    P.vec(e + ".lin_k.bias", at[i] + GAttnL::BKE, 64);
    P.vec(e + ".lin_v.weight", at[i] + GAttnL::WVE, MAT64);
    P.vec(e + ".lin_v.bias", at[i] + GAttnL::BVE, 64);
  }
}

static void recipe_encoder(Packer& P) {
  using B = EncBlob;
  recipe_encoder_attention(P);
  recipe_drift(P, "lsde_func.f_func", B::SDE + EncSdeL::F);
  recipe_diff(P, "lsde_func.g_nus", B::SDE + EncSdeL::GN);
  recipe_diff(P, "lsde_func.g_argo", B::SDE + EncSdeL::GA);
  {  // GRU_Unit
    using L = EncGruL;
    const int g = B::GRU;
    P.mat("gru_unit.update_gate.0.weight", g + L::WUR_H, 64, 64, 128, 0);
    P.mat("gru_unit.reset_gate.0.weight", g + L::WUR_H + MAT64, 64, 64, 128, 0);
    P.mat("gru_unit.update_gate.0.weight", g + L::WUR_X, 64, 64, 128, 64);
    P.mat("gru_unit.reset_gate.0.weight", g + L::WUR_X + MAT64, 64, 64, 128, 64);
    P.vec("gru_unit.update_gate.0.bias", g + L::BUR, 64);
    P.vec("gru_unit.reset_gate.0.bias", g + L::BUR + 64, 64);
    P.lin("gru_unit.update_gate.2", g + L::WU2, g + L::BU2);
    P.lin("gru_unit.reset_gate.2", g + L::WR2, g + L::BR2);
    P.mat("gru_unit.new_state_net.0.weight", g + L::WN_X, 64, 64, 128, 0);
    P.mat("gru_unit.new_state_net.0.weight", g + L::WN_H, 64, 64, 128, 64);
    P.vec("gru_unit.new_state_net.0.bias", g + L::BN0, 64);
    P.lin("gru_unit.new_state_net.2", g + L::WN2, g + L::BN2);
  }
  P.vec("hidden", B::HIDDEN, 64);
  {  // the same matrices once more as split-precision images for k_enc_recur_coop
    using C = EncCoopL6;
    auto at = [](int m) { return B::COOP6 + m * MAT64X6; };
    // the layers in front of a tanh carry its 2 / ln 2, those in front of a sigmoid its -1 / ln 2 (tile.hpp tanh_prescaled /
    // sigmoid_prescaled: one multiply less per activation); k_enc_recur_coop scales the matching biases when it copies them to LDS
    P.scale = TANH_PRESCALE;
    P.mat6("lsde_func.f_func.net.0.weight", at(C::F0), 64, 64, 66, 0);
    P.mat6("lsde_func.f_func.net.2.weight", at(C::F2), 64, 64, 64);
    P.mat6("lsde_func.g_nus.net.0.weight", at(C::N0), 64, 64, 66, 0);
    P.mat6("lsde_func.g_nus.net.2.weight", at(C::N2), 64, 64, 64);
    P.mat6("lsde_func.g_argo.net.0.weight", at(C::A0), 64, 64, 66, 0);
    P.mat6("lsde_func.g_argo.net.2.weight", at(C::A2), 64, 64, 64);
    P.mat6("gru_unit.update_gate.0.weight", at(C::UH), 64, 64, 128, 0);
    P.mat6("gru_unit.reset_gate.0.weight", at(C::RH), 64, 64, 128, 0);
    P.mat6("gru_unit.update_gate.0.weight", at(C::UX), 64, 64, 128, 64);
    P.mat6("gru_unit.reset_gate.0.weight", at(C::RX), 64, 64, 128, 64);
    P.mat6("gru_unit.new_state_net.0.weight", at(C::NX), 64, 64, 128, 0);
    P.mat6("gru_unit.new_state_net.0.weight", at(C::NH), 64, 64, 128, 64);
    P.scale = SIGMOID_PRESCALE;
    P.mat6("gru_unit.update_gate.2.weight", at(C::U2), 64, 64, 64);
    P.mat6("gru_unit.reset_gate.2.weight", at(C::R2), 64, 64, 64);
    P.scale = 0.f;
    P.mat6("lsde_func.f_func.net.4.weight", at(C::F4), 64, 64, 64);
    P.mat6("gru_unit.new_state_net.2.weight", at(C::N2G), 64, 64, 64);
  }
}

// vanilla LocalEncoder (GENC:52-93): AA / AL as above + TemporalEncoder with `nl` layers
static void recipe_encoder_grid(Packer& P, int nl) {
  recipe_encoder_attention(P);
  const std::string t = "temporal_encoder.";
  const int tok = EncGridBlob::TOK;
  P.vec(t + "padding_token", tok + EncGridBlob::TOK_PAD, 21 * 64);
  P.vec(t + "cls_token", tok + EncGridBlob::TOK_CLS, 64);
  P.vec(t + "pos_embed", tok + EncGridBlob::TOK_POS, 22 * 64);
  for (int i = 0; i < nl; ++i) {
    const std::string l = t + "transformer_encoder.layers." + std::to_string(i);
    const int b = EncGridBlob::layer(i);
    using Q = NodeProjL<3>;
    P.ln(l + ".norm1", b + TrLayerL::QKV + Q::N1G, b + TrLayerL::QKV + Q::N1B);
    P.mat(l + ".self_attn.in_proj_weight", b + TrLayerL::QKV + Q::W, 192, 64, 64);     // rows q | k | v
    P.vec(l + ".self_attn.in_proj_bias", b + TrLayerL::QKV + Q::B, 192);
    P.lin(l + ".self_attn.out_proj", b + TrLayerL::OUT + TrOutL::WOUT, b + TrLayerL::OUT + TrOutL::BOUT);
    P.ln(l + ".norm2", b + TrLayerL::OUT + TrOutL::N2G, b + TrLayerL::OUT + TrOutL::N2B);
    P.lin(l + ".linear1", b + TrLayerL::FFN + FfnL::W1, b + TrLayerL::FFN + FfnL::B1, 256, 64);
    P.lin(l + ".linear2", b + TrLayerL::FFN + FfnL::W2, b + TrLayerL::FFN + FfnL::B2, 64, 256);
  }
  P.ln(t + "transformer_encoder.norm", EncGridBlob::norm(nl), EncGridBlob::norm(nl) + 64);
}

// MLPDecoder (GDEC:11-63) for T future steps (2T <= 128 outputs per head)
static void recipe_decoder_mlp(Packer& P, int T) {
  using I = MlpInitL;
  using H = MlpHeadsL;
  const int b = MlpDecBlob::INIT, h = MlpDecBlob::HEADS;
  P.mat("aggr_embed.0.weight", b + I::WA_G, 64, 64, 128, 0);      // cat(global, local)
  P.mat("aggr_embed.0.weight", b + I::WA_L, 64, 64, 128, 64);
  P.vec("aggr_embed.0.bias", b + I::BA, 64);
  P.ln("aggr_embed.1", b + I::AG, b + I::AE);
  P.mat("pi.0.weight", b + I::WP_L, 64, 64, 128, 0);              // cat(local, global)
  P.mat("pi.0.weight", b + I::WP_G, 64, 64, 128, 64);
  P.vec("pi.0.bias", b + I::BP, 64);
  P.ln("pi.1", b + I::PG, b + I::PE);
  P.lin("pi.3", b + I::WP3, b + I::BP3);
  P.ln("pi.4", b + I::PG4, b + I::PE4);
  P.vec("pi.6.weight", b + I::WP6, 64);
  P.vec("pi.6.bias", b + I::BP6, 1);
  const char* heads[2] = {"loc", "scale"};
  const int w0[2] = {H::L_W0, H::S_W0}, b0[2] = {H::L_B0, H::S_B0}, g[2] = {H::L_G, H::S_G}, e[2] = {H::L_E, H::S_E},
            w3[2] = {H::L_W3, H::S_W3}, b3[2] = {H::L_B3, H::S_B3};
  for (int k = 0; k < 2; ++k) {
    const std::string p = heads[k];
    P.lin(p + ".0", h + w0[k], h + b0[k]);
    P.ln(p + ".1", h + g[k], h + e[k]);
    P.mat_pad(p + ".3.weight", h + w3[k], 8, 4, 64, 2 * T);       // [2T, 64] zero-padded to 128 rows
    P.vec(p + ".3.bias", h + b3[k], 2 * T);
  }
}

static void recipe_aggregator(Packer& P, int nl, int K) {
  recipe_edge_embed(P, "rel_embed", AggBlob::REL);
  recipe_edge_embed6(P, "rel_embed", AggBlob::REL6);
#if TSDE_SPLIT_H3
  recipe_edge_fused_embed(P, "rel_embed", AggBlob::REL6G);
#endif
  for (int i = 0; i < nl; ++i) {
    const std::string p = "global_interactor_layers." + std::to_string(i);
    const int b = AggBlob::layer(i);
    using Q = NodeProjL<3>;
    P.ln(p + ".norm1", b + AggLayerL::QKV + Q::N1G, b + AggLayerL::QKV + Q::N1B);
    const char* qkv[3] = {".lin_q_node", ".lin_k_node", ".lin_v_node"};
    for (int j = 0; j < 3; ++j) {
      P.mat(p + qkv[j] + ".weight", b + AggLayerL::QKV + Q::W + j * MAT64, 64, 64, 64);
      P.vec(p + qkv[j] + ".bias", b + AggLayerL::QKV + Q::B + j * 64, 64);
    }
    P.mat(p + ".lin_k_edge.weight", b + AggLayerL::EDGE + GEdgeL::WKV, 64, 64, 64);
    P.mat(p + ".lin_v_edge.weight", b + AggLayerL::EDGE + GEdgeL::WKV + MAT64, 64, 64, 64);
    P.vec(p + ".lin_k_edge.bias", b + AggLayerL::EDGE + GEdgeL::BKV, 64);
    P.vec(p + ".lin_v_edge.bias", b + AggLayerL::EDGE + GEdgeL::BKV + 64, 64);
    recipe_upd_ffn(P, p, b + AggLayerL::UPD, b + AggLayerL::FFN);
    pack_kv6(P, p + ".lin_k_edge", p + ".lin_v_edge", b + AggLayerL::EDGE6 + GEdgeL6::WKV, b + AggLayerL::EDGE6 + GEdgeL6::BKV);
    recipe_upd_ffn6(P, p, b + AggLayerL::UPD6, b + AggLayerL::FFN6);
    P.vec(p + ".lin_k_edge.weight", b + AggLayerL::ATTN + GAttnL::WKE, MAT64);
    P.vec(p + ".lin_k_edge.bias", b + AggLayerL::ATTN + GAttnL::BKE, 64);
    P.vec(p + ".lin_v_edge.weight", b + AggLayerL::ATTN + GAttnL::WVE, MAT64);
    P.vec(p + ".lin_v_edge.bias", b + AggLayerL::ATTN + GAttnL::BVE, 64);
  }
  P.ln("norm", AggBlob::norm(nl), AggBlob::norm(nl) + 64);
  // multihead_proj [K*64, 64]: mode k owns rows 64k..64k+63 (AGG:56 view(-1, K, 64))
  const float* w = P.src("multihead_proj.weight");
  const float* bsrc = P.src("multihead_proj.bias");
  if (!P.dry)
    for (int k = 0; k < K; ++k) {
      P.mat_raw(w + int64_t(k) * MAT64, P.blob + AggBlob::proj(nl, k), 4, 4, 64, 0);
      P.vec_raw(bsrc + k * 64, P.blob + AggBlob::proj(nl, k) + MAT64, 64);
    }
}

static void recipe_head(Packer& P, const std::string& p, int base) {
  P.lin(p + ".0", base + HeadL::W0, base + HeadL::B0);
  P.ln(p + ".1", base + HeadL::G, base + HeadL::E);
  P.vec(p + ".3.weight", base + HeadL::W3, 128);
  P.vec(p + ".3.bias", base + HeadL::B3, 2);
}
static void recipe_decoder(Packer& P) {
  using I = DecInitL;
  const int b = DecBlob::INIT;
  P.mat("aggr_embed.0.weight", b + I::WA_G, 64, 64, 128, 0);     // cat(global, local): DEC:82
  P.mat("aggr_embed.0.weight", b + I::WA_L, 64, 64, 128, 64);
  P.vec("aggr_embed.0.bias", b + I::BA, 64);
  P.ln("aggr_embed.1", b + I::AG, b + I::AE);
  P.mat("pi.0.weight", b + I::WP_L, 64, 64, 128, 0);             // cat(local, global): DEC:93-94
  P.mat("pi.0.weight", b + I::WP_G, 64, 64, 128, 64);
  P.vec("pi.0.bias", b + I::BP, 64);
  P.ln("pi.1", b + I::PG, b + I::PE);
  P.vec("pi.3.weight", b + I::WP3, 64);
  P.vec("pi.3.bias", b + I::BP3, 1);
  recipe_drift(P, "lsde_func.f_func", DecBlob::SDE + DecSdeL::F);
  recipe_diff(P, "lsde_func.g_func", DecBlob::SDE + DecSdeL::G);
  recipe_head(P, "decoder", DecBlob::SDE + DecSdeL::LOC);
  recipe_head(P, "scale", DecBlob::SDE + DecSdeL::SCALE);
#if TSDE_SPLIT_H3
  {
    using D = DecSdeL6;
    const int d = DecBlob::SDE6;
    const std::string f = "lsde_func.f_func.net.", g = "lsde_func.g_func.net.";
    // the four layers in front of a tanh carry its 2 / ln 2 (tile.hpp TANH_PRESCALE, tanh_prescaled): one multiply less per tanh
    P.scale = TANH_PRESCALE;
    P.mat6(f + "0.weight", d + D::W0FG, 64, 64, 66, 0);                 // jo-major planes: the two row blocks concatenate
    P.mat6(g + "0.weight", d + D::W0FG + MAT64X6, 64, 64, 66, 0);
    P.vec(f + "0.bias", d + D::B0FG, 64);
    P.vec(g + "0.bias", d + D::B0FG + 64, 64);
    P.col(f + "0.weight", d + D::WSFG, 64, 66, 64);
    P.col(g + "0.weight", d + D::WSFG + 64, 64, 66, 64);
    P.col(f + "0.weight", d + D::WCFG, 64, 66, 65);
    P.col(g + "0.weight", d + D::WCFG + 64, 64, 66, 65);
    P.mat6(f + "2.weight", d + D::F_W2, 64, 64, 64);
    P.vec(f + "2.bias", d + D::F_B2, 64);
    P.mat6(g + "2.weight", d + D::G_W2, 64, 64, 64);
    P.vec(g + "2.bias", d + D::G_B2, 64);
    P.scale = 0.f;
    P.mat6(f + "4.weight", d + D::F_W4, 64, 64, 64);
    P.vec(f + "4.bias", d + D::F_B4, 64);
    P.vec(g + "4.weight", d + D::G_W4, 64);
    P.vec(g + "4.bias", d + D::G_B4, 1);
  }
#else
  recipe_drift6(P, "lsde_func.f_func", DecBlob::SDE6 + DecSdeL6::F);
  recipe_diff6(P, "lsde_func.g_func", DecBlob::SDE6 + DecSdeL6::G);
#endif
#if TSDE_SPLIT_H3
  {
    using HP = HeadPairL6;
    const int h = DecBlob::SDE6 + DecSdeL6::LOC;
    pack_kv6(P, "decoder.0", "scale.0", h + HP::W0, h + HP::B0);      // stacked first layers + both biases
    P.ln("decoder.1", h + HP::G_LOC, h + HP::E_LOC);
    P.vec("decoder.3.weight", h + HP::W3_LOC, 128);
    P.vec("decoder.3.bias", h + HP::B3_LOC, 2);
    P.ln("scale.1", h + HP::G_SC, h + HP::E_SC);
    P.vec("scale.3.weight", h + HP::W3_SC, 128);
    P.vec("scale.3.bias", h + HP::B3_SC, 2);
  }
#else
  recipe_head(P, "decoder", DecBlob::SDE6 + DecSdeL6::LOC);
  recipe_head(P, "scale", DecBlob::SDE6 + DecSdeL6::SCALE);
#endif
}

static void recipe_aggr_embed_bwd(Packer& P, int b) {      // aggr_embed of either decoder: forward halves + transposes
  using I = InitBwdL;
  P.mat("aggr_embed.0.weight", b + I::WA_G, 64, 64, 128, 0);
  P.mat("aggr_embed.0.weight", b + I::WA_L, 64, 64, 128, 64);
  P.vec("aggr_embed.0.bias", b + I::BA, 64);
  P.ln("aggr_embed.1", b + I::AG, b + I::AE);
  P.matT("aggr_embed.0.weight", b + I::WA_GT, 128, 0);
  P.matT("aggr_embed.0.weight", b + I::WA_LT, 128, 64);
}
// MLPDecoder backward (L2 on loc): loc head + aggr_embed; `scale` and `pi` get no gradient from that loss
static void recipe_decoder_mlp_bwd(Packer& P, int T) {
  using H = MlpHeadBwdL;
  const int h = MlpDecBwdBlob::HEAD;
  P.lin("loc.0", h + H::W0, h + H::B0);
  P.ln("loc.1", h + H::G, h + H::E);
  P.mat_pad("loc.3.weight", h + H::W3, 8, 4, 64, 2 * T);
  P.vec("loc.3.bias", h + H::B3, 2 * T);
  P.matT("loc.3.weight", h + H::W3T, 64, 0, 4, 8, 2 * T);          // (W3 [2T,64])^T as a 64 x 128 image
  P.matT("loc.0.weight", h + H::W0T, 64);
  recipe_aggr_embed_bwd(P, MlpDecBwdBlob::INIT);
}

// backward images of the decoder stage (loc head, drift/diffusion nets, aggr_embed); `pi` and `scale` receive no
// gradient from the L2 regression loss and are not packed
static void recipe_decoder_bwd(Packer& P) {
  const int s = DecBwdBlob::SWEEP, h = DecBwdBlob::HEAD, b = DecBwdBlob::INIT;
  P.matT("lsde_func.f_func.net.0.weight", s + SweepL::F_W0T, 66);
  P.matT("lsde_func.f_func.net.2.weight", s + SweepL::F_W2T, 64);
  P.matT("lsde_func.f_func.net.4.weight", s + SweepL::F_W4T, 64);
  P.matT("lsde_func.g_func.net.0.weight", s + SweepL::G_W0T, 66);
  P.matT("lsde_func.g_func.net.2.weight", s + SweepL::G_W2T, 64);
  P.vec("lsde_func.g_func.net.4.weight", s + SweepL::G_W4, 64);
  recipe_head(P, "decoder", h + HeadBwdL::FWD);
  P.matT("decoder.0.weight", h + HeadBwdL::W0T, 64);
  recipe_aggr_embed_bwd(P, b);
  // not packed, but they receive a gradient: listed so that the gradient slots of trajsde_decoder_l2_backward
  // follow this parameter table
  for (const char* n : {"lsde_func.f_func.net.0.bias", "lsde_func.f_func.net.2.bias", "lsde_func.f_func.net.4.bias",
                        "lsde_func.g_func.net.0.bias", "lsde_func.g_func.net.2.bias", "lsde_func.g_func.net.4.bias"})
    P.index(n);
}

// Laplace NLL: the L2 table and images, then the scale head's (its gradient slots follow the L2 ones)
static void recipe_decoder_nll_bwd(Packer& P) {
  recipe_decoder_bwd(P);
  const int h = DecNllBwdBlob::HEAD_SC;
  recipe_head(P, "scale", h + HeadBwdL::FWD);
  P.matT("scale.0.weight", h + HeadBwdL::W0T, 64);
}

// node-level backward images of one attention block (layouts.hpp NodeBlockBwdL); `p` = the block's parameter prefix
static void recipe_node_block_bwd(Packer& P, const std::string& p, int base) {
  const int a = base + NodeBlockBwdL::FFN_A, b = base + NodeBlockBwdL::FFN_B, u = base + NodeBlockBwdL::UPD;
  P.lin(p + ".mlp.0", a + FfnBwdAL::W1, a + FfnBwdAL::B1, 256, 64);
  P.matT(p + ".mlp.3.weight", a + FfnBwdAL::W2T, 256, 0, 16, 4);         // (W2 [64,256])^T as a 256x64 image
  P.matT(p + ".mlp.0.weight", b + FfnBwdBL::W1T, 64, 0, 4, 16);          // (W1 [256,64])^T as a 64x256 image
  P.vec(p + ".norm2.weight", b + FfnBwdBL::N2G, 64);
  P.lin(p + ".lin_ih", u + UpdBwdL::WIH, u + UpdBwdL::BIH);
  P.lin(p + ".lin_hh", u + UpdBwdL::WHH, u + UpdBwdL::BHH);
  P.lin(p + ".lin_self", u + UpdBwdL::WSELF, u + UpdBwdL::BSELF);
  P.matT(p + ".out_proj.weight", u + UpdBwdL::WOUT_T, 64);
  P.matT(p + ".lin_ih.weight", u + UpdBwdL::WIH_T, 64);
  P.matT(p + ".lin_hh.weight", u + UpdBwdL::WHH_T, 64);
  P.matT(p + ".lin_self.weight", u + UpdBwdL::WSELF_T, 64);
  for (const char* n : {".mlp.3.bias", ".out_proj.bias", ".norm2.bias"}) P.index(p + n);
}
static void recipe_edge_embed_bwd(Packer& P, const std::string& p, int base) {
  recipe_edge_embed6(P, p, base + EdgeBwdL::FWD);
  P.matT(p + ".aggr_embed.2.weight", base + EdgeBwdL::W2T, 64);
  P.matT(p + ".module_list.0.3.weight", base + EdgeBwdL::WA3T, 64);
  P.matT(p + ".module_list.1.3.weight", base + EdgeBwdL::WB3T, 64);
}
static void recipe_aggregator_bwd(Packer& P, int nl, int K) {
  recipe_edge_embed_bwd(P, "rel_embed", AggBwdBlob::REL);
  for (int i = 0; i < nl; ++i) {
    const std::string p = "global_interactor_layers." + std::to_string(i);
    const int b = AggBwdBlob::layer(i);
    recipe_node_block_bwd(P, p, b + AggLayerBwdL::NODE);
    using Q = ProjBwdL<3>;
    P.ln(p + ".norm1", b + AggLayerBwdL::PROJ + Q::N1G, b + AggLayerBwdL::PROJ + Q::N1B);
    const char* qkv[3] = {".lin_q_node", ".lin_k_node", ".lin_v_node"};
    for (int j = 0; j < 3; ++j) {
      P.matT(p + qkv[j] + ".weight", b + AggLayerBwdL::PROJ + Q::WT + j * MAT64, 64);
      P.index(p + qkv[j] + ".bias");
    }
    P.vec(p + ".lin_k_edge.weight", b + AggLayerBwdL::ATTN + GAttnL::WKE, MAT64);
    P.vec(p + ".lin_k_edge.bias", b + AggLayerBwdL::ATTN + GAttnL::BKE, 64);
    P.vec(p + ".lin_v_edge.weight", b + AggLayerBwdL::ATTN + GAttnL::WVE, MAT64);
    P.vec(p + ".lin_v_edge.bias", b + AggLayerBwdL::ATTN + GAttnL::BVE, 64);
  }
  P.ln("norm", AggBwdBlob::norm(nl), AggBwdBlob::norm(nl) + 64);
  const float* w = P.src("multihead_proj.weight");
  P.index("multihead_proj.bias");
  if (!P.dry)
    for (int k = 0; k < K; ++k)
      P.matT_raw(w + int64_t(k) * MAT64, P.blob + AggBwdBlob::proj(nl, k), 4, 4, 64, 0, 1 << 30);
}

static void recipe_edge_kv_bwd(Packer& P, const std::string& p, const std::string& embed, int kv, int emb) {
  recipe_edge_embed6(P, p + "." + embed, kv + EdgeKvBwdL::FWD);
  P.mat6(p + ".lin_k.weight", kv + EdgeKvBwdL::WK6, 64, 64, 64);
  P.vec(p + ".lin_k.bias", kv + EdgeKvBwdL::BK, 64);
  P.index(p + ".lin_v.bias");
  P.matT(p + ".lin_k.weight", kv + EdgeKvBwdL::WKT, 64);
  P.matT(p + ".lin_v.weight", kv + EdgeKvBwdL::WVT, 64);
  recipe_edge_embed_bwd(P, p + "." + embed, emb);
}
static void recipe_proj1_bwd(Packer& P, const std::string& p, int base) {
  P.ln(p + ".norm1", base + ProjBwdL<1>::N1G, base + ProjBwdL<1>::N1B);
  P.matT(p + ".lin_q.weight", base + ProjBwdL<1>::WT, 64);
  P.index(p + ".lin_q.bias");
}
// AA / AL backward images (shared by the SDE encoder and the vanilla one)
static void recipe_encoder_attention_bwd(Packer& P) {
  using B = EncBwdBlob;
  recipe_edge_kv_bwd(P, "aa_encoder", "nbr_embed", B::AA_EDGEKV, B::AA_EDGEEMB);
  recipe_node_block_bwd(P, "aa_encoder", B::AA_NODE);
  recipe_proj1_bwd(P, "aa_encoder", B::AA_PROJ);
  {
    using L = AaCenterL;
    const std::string c = "aa_encoder.center_embed.embed.";
    const int t = B::AA_CTAIL, h = B::AA_CHEAD;
    P.vec(c + "0.weight", t + L::W0, 128);
    P.vec(c + "0.bias", t + L::B0, 64);
    P.ln(c + "1", t + L::G1, t + L::E1);
    P.lin(c + "3", t + L::W3, t + L::B3);
    P.ln(c + "4", t + L::G4, t + L::E4);
    P.lin(c + "6", t + L::W6, t + L::B6);
    P.ln(c + "7", t + L::G7, t + L::E7);
    P.matT(c + "6.weight", t + CenterTailL::W6T, 64);
    P.vec(c + "0.weight", h + EdgeL::A_W0, 128);
    P.vec(c + "0.bias", h + EdgeL::A_B0, 64);
    P.ln(c + "1", h + EdgeL::A_G, h + EdgeL::A_E);
    P.matT(c + "3.weight", h + EdgeBwdL::WA3T, 64);
    P.index("aa_encoder.bos_token");
  }
  recipe_proj1_bwd(P, "al_encoder", B::AL_PROJ);
  recipe_edge_kv_bwd(P, "al_encoder", "lane_embed", B::AL_EDGEKV, B::AL_EDGEEMB);
  recipe_node_block_bwd(P, "al_encoder", B::AL_NODE);
}

static void recipe_encoder_bwd(Packer& P) {
  using B = EncBwdBlob;
  recipe_encoder_attention_bwd(P);
  {
    using L = EncSdeBwdL;
    const int s = B::SDE;
    P.matT("lsde_func.f_func.net.0.weight", s + L::F_W0T, 66);
    P.matT("lsde_func.f_func.net.2.weight", s + L::F_W2T, 64);
    P.matT("lsde_func.f_func.net.4.weight", s + L::F_W4T, 64);
    P.matT("lsde_func.g_nus.net.0.weight", s + L::GN_W0T, 66);
    P.matT("lsde_func.g_nus.net.2.weight", s + L::GN_W2T, 64);
    P.matT("lsde_func.g_argo.net.0.weight", s + L::GA_W0T, 66);
    P.matT("lsde_func.g_argo.net.2.weight", s + L::GA_W2T, 64);
    P.vec("lsde_func.g_nus.net.4.weight", s + L::GN_W4, 64);
    P.vec("lsde_func.g_argo.net.4.weight", s + L::GA_W4, 64);
    for (const char* net : {"f_func", "g_nus", "g_argo"})
      for (const char* l : {"0", "2", "4"}) P.index(std::string("lsde_func.") + net + ".net." + l + ".bias");
  }
  {
    using L = GruBwdL;
    const int g = B::GRU;
    P.matT("gru_unit.new_state_net.2.weight", g + L::WN2T, 64);
    P.matT("gru_unit.new_state_net.0.weight", g + L::WNXT, 128, 0);
    P.matT("gru_unit.new_state_net.0.weight", g + L::WNHT, 128, 64);
    P.matT("gru_unit.update_gate.2.weight", g + L::WU2T, 64);
    P.matT("gru_unit.reset_gate.2.weight", g + L::WR2T, 64);
    P.matT("gru_unit.update_gate.0.weight", g + L::UHT, 128, 0);
    P.matT("gru_unit.reset_gate.0.weight", g + L::RHT, 128, 0);
    P.matT("gru_unit.update_gate.0.weight", g + L::UXT, 128, 64);
    P.matT("gru_unit.reset_gate.0.weight", g + L::RXT, 128, 64);
    for (const char* n : {"update_gate.0", "update_gate.2", "reset_gate.0", "reset_gate.2", "new_state_net.0", "new_state_net.2"})
      P.index(std::string("gru_unit.") + n + ".bias");
  }
  P.index("hidden");
}

// vanilla LocalEncoder backward: AA / AL as above + the temporal transformer
static void recipe_encoder_grid_bwd(Packer& P, int nl) {
  recipe_encoder_attention_bwd(P);
  const std::string t = "temporal_encoder.";
  for (int i = 0; i < nl; ++i) {
    const std::string l = t + "transformer_encoder.layers." + std::to_string(i);
    const int b = EncGridBwdBlob::layer(i);
    const int a = b + TrLayerBwdL::FFN_A, bb = b + TrLayerBwdL::FFN_B, pr = b + TrLayerBwdL::PROJ;
    P.lin(l + ".linear1", a + FfnBwdAL::W1, a + FfnBwdAL::B1, 256, 64);
    P.matT(l + ".linear2.weight", a + FfnBwdAL::W2T, 256, 0, 16, 4);
    P.matT(l + ".linear1.weight", bb + FfnBwdBL::W1T, 64, 0, 4, 16);
    P.vec(l + ".norm2.weight", bb + FfnBwdBL::N2G, 64);
    P.matT(l + ".self_attn.out_proj.weight", b + TrLayerBwdL::WOUT_T, 64);
    P.ln(l + ".norm1", pr + ProjBwdL<3>::N1G, pr + ProjBwdL<3>::N1B);
    // in_proj_weight [192,64] = q | k | v row blocks: three transposed 64x64 images
    const float* w = P.src(l + ".self_attn.in_proj_weight");
    if (!P.dry)
      for (int j = 0; j < 3; ++j)
        P.matT_raw(w + int64_t(j) * MAT64, P.blob + pr + ProjBwdL<3>::WT + j * MAT64, 4, 4, 64, 0, 1 << 30);
    for (const char* n : {".self_attn.in_proj_bias", ".self_attn.out_proj.bias", ".linear2.bias", ".norm2.bias"}) P.index(l + n);
  }
  P.ln(t + "transformer_encoder.norm", EncGridBwdBlob::norm(nl), EncGridBwdBlob::norm(nl) + 64);
  for (const char* n : {"padding_token", "cls_token", "pos_embed"}) P.index(t + n);
}

static bool run_recipe(Packer& P, int stage, int nl, int K) {
  switch (stage) {
    case TRAJSDE_STAGE_ENCODER: recipe_encoder(P); return true;
    case TRAJSDE_STAGE_AGGREGATOR: recipe_aggregator(P, nl, K); return true;
    case TRAJSDE_STAGE_DECODER: recipe_decoder(P); return true;
    case TRAJSDE_STAGE_DECODER_BWD: recipe_decoder_bwd(P); return true;
    case TRAJSDE_STAGE_DECODER_NLL_BWD: recipe_decoder_nll_bwd(P); return true;
    case TRAJSDE_STAGE_AGGREGATOR_BWD: recipe_aggregator_bwd(P, nl, K); return true;
    case TRAJSDE_STAGE_ENCODER_BWD: recipe_encoder_bwd(P); return true;
    case TRAJSDE_STAGE_ENCODER_GRID: recipe_encoder_grid(P, nl); return true;
    case TRAJSDE_STAGE_DECODER_MLP: recipe_decoder_mlp(P, nl); return true;
    case TRAJSDE_STAGE_DECODER_MLP_BWD: recipe_decoder_mlp_bwd(P, nl); return true;
    case TRAJSDE_STAGE_ENCODER_GRID_BWD: recipe_encoder_grid_bwd(P, nl); return true;
  }
  return false;
}

std::vector<std::string> stage_param_names(int stage, int num_layers, int num_modes) {
  Packer P{true};
  if (!run_recipe(P, stage, num_layers, num_modes)) return {};
  return P.names;
}

// ---- event profiler (see common.hpp)
struct ProfRec {
  std::string tag;
  bool dominant;
  std::vector<hipEvent_t> beg, end;
};
static int g_prof_mode = 0;
static std::vector<ProfRec> g_prof;
int profile_mode() { return g_prof_mode; }
static ProfRec& prof_rec(const char* tag, bool dominant) {
  for (auto& r : g_prof)
    if (r.tag == tag) return r;
  g_prof.push_back(ProfRec{tag, dominant, {}, {}});
  return g_prof.back();
}
void profile_begin(const char* tag, hipStream_t st, bool dominant) {
  ProfRec& r = prof_rec(tag, dominant);
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  r.beg.push_back(e);
}
void profile_end(const char* tag, hipStream_t st, bool dominant) {
  ProfRec& r = prof_rec(tag, dominant);
  if (r.end.size() >= r.beg.size()) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  r.end.push_back(e);
}

}  // namespace tsde

using namespace tsde;

extern "C" {

int trajsde_profile_mode(int mode) {
  g_prof_mode = mode;
  return TRAJSDE_OK;
}

// "tag count total_ms dominant\n" per kernel tag; waits for the recorded events, then forgets them
int64_t trajsde_profile_report(char* buf, int64_t cap) {
  std::string out;
  for (auto& r : g_prof) {
    double total = 0;
    size_t n = r.end.size() < r.beg.size() ? r.end.size() : r.beg.size();
    for (size_t i = 0; i < n; ++i) {
      float ms = 0;
      if (hipEventSynchronize(r.end[i]) == hipSuccess && hipEventElapsedTime(&ms, r.beg[i], r.end[i]) == hipSuccess) total += ms;
    }
    for (auto e : r.beg) (void)hipEventDestroy(e);
    for (auto e : r.end) (void)hipEventDestroy(e);
    char line[256];
    snprintf(line, sizeof(line), "%s %zu %.6f %d\n", r.tag.c_str(), n, total, r.dominant ? 1 : 0);
    out += line;
  }
  g_prof.clear();
  if (buf && cap > 0) {
    const size_t n = out.size() < size_t(cap - 1) ? out.size() : size_t(cap - 1);
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return int64_t(out.size());
}

const char* trajsde_last_error(void) { return last_error_ref().c_str(); }
int trajsde_split_products(void) { return TSDE_SPLIT_H3 ? 3 : 6; }

int trajsde_state_storage(int mode) {
  const int prev = tsde::g_state_bf16.exchange(mode == 1 ? 1 : 0);
  return prev;
}

int trajsde_range_status(int reset, uint32_t* sites_out, void* stream) {
  static const char* const names[tsde::RS_SITES] = {"decoder SDE state", "decoder embedding inputs", "encoder latent state",
                                                    "aa_out rows entering the GRU", "attention aggregate / gated update",
                                                    "FFN hidden units", "a weight (no fp16 image)"};
  unsigned all = 0;
  for (tsde::RangeReader r : tsde::range_readers()) {
    unsigned w = 0;
    TS_HIP(r(&w, reset != 0, static_cast<hipStream_t>(stream)));
    all |= w;
  }
  if (sites_out) *sites_out = all;
  if (all == 0) return TRAJSDE_OK;
  std::string msg = "fp16x3 split-precision range exceeded (|value| >= 65504) in: ";
  bool first = true;
  for (unsigned i = 0; i < tsde::RS_SITES; ++i)
    if (all & (1u << i)) { msg += (first ? "" : ", "); msg += names[i]; first = false; }
  msg += " -- the results of the affected launches are not valid; rebuild with TRAJSDE_SPLIT=bf16x6 for fp32's exponent range";
  return tsde::fail(TRAJSDE_ERR_UNSUPPORTED, msg);
}
int trajsde_abi_version(void) { return 10; }

int trajsde_param_count(int stage, int num_layers, int num_modes) {
  Packer P{true};
  if (!run_recipe(P, stage, num_layers, num_modes)) return fail(TRAJSDE_ERR_INVALID, "unknown stage");
  return int(P.names.size());
}

const char* trajsde_param_name(int stage, int index, int num_layers, int num_modes) {
  static thread_local std::string out;
  Packer P{true};
  if (!run_recipe(P, stage, num_layers, num_modes) || index < 0 || index >= int(P.names.size())) return nullptr;
  out = P.names[index];
  return out.c_str();
}

int64_t trajsde_blob_floats(int stage, int num_layers, int num_modes) {
  switch (stage) {
    case TRAJSDE_STAGE_ENCODER: return EncBlob::SIZE;
    case TRAJSDE_STAGE_AGGREGATOR: return AggBlob::size(num_layers, num_modes);
    case TRAJSDE_STAGE_DECODER: return DecBlob::SIZE;
    case TRAJSDE_STAGE_DECODER_BWD: return DecBwdBlob::SIZE;
    case TRAJSDE_STAGE_DECODER_NLL_BWD: return DecNllBwdBlob::SIZE;
    case TRAJSDE_STAGE_AGGREGATOR_BWD: return AggBwdBlob::size(num_layers, num_modes);
    case TRAJSDE_STAGE_ENCODER_BWD: return EncBwdBlob::SIZE;
    case TRAJSDE_STAGE_ENCODER_GRID: return EncGridBlob::size(num_layers);
    case TRAJSDE_STAGE_DECODER_MLP: return MlpDecBlob::SIZE;
    case TRAJSDE_STAGE_DECODER_MLP_BWD: return MlpDecBwdBlob::SIZE;
    case TRAJSDE_STAGE_ENCODER_GRID_BWD: return EncGridBwdBlob::size(num_layers);
  }
  return fail(TRAJSDE_ERR_INVALID, "unknown stage");
}

}  // extern "C"
namespace {
struct PackLaunch {
  PackJobs tab;
  int n, max_count;
};
struct PackPlan {
  int stage, nl, K;
  int64_t need_floats;
  std::vector<const float*> params;
  std::vector<PackLaunch> tables;          // built against a stand-in blob address (pack_standin_blob): re-based per call
  std::vector<PackJob> jobs;               // the same jobs as one list (trajsde_pack_weights_many merges the stages' lists)
  int standin = 0;                         // which stand-in range (one that holds none of the parameters)
};
// an address range no allocation lives in (2^44 .. 2^44 + blob size): the recipe's `blob + offset` arithmetic stays ordinary pointer
// arithmetic on a non-null base, and what points into the blob is recognisable afterwards
float* pack_standin_blob(int slot = 0) { return reinterpret_cast<float*>(uintptr_t(1) << (44 - slot)); }
// the launch tables of one stage's packing: the recipe run against `params`, destinations relative to a null blob
void build_pack_tables(int stage, int num_layers, int num_modes, const std::vector<std::string>& names, const float* const* params,
                       std::vector<PackLaunch>& out, std::vector<PackJob>& flat, int standin) {
  Packer wet{false};
  wet.names = names;
  wet.params = params;
  wet.blob = pack_standin_blob(standin);
  wet.stream = nullptr;
  run_recipe(wet, stage, num_layers, num_modes);
  flat = wet.jobs;
  for (int pass = 0; pass < 2; ++pass) {
    PackLaunch cur;
    cur.n = cur.max_count = 0;
    for (const PackJob& j : wet.jobs) {
      if (j.pass != pass) continue;
      cur.tab.j[cur.n++] = j;
      cur.max_count = j.count > cur.max_count ? j.count : cur.max_count;
      if (cur.n == PACK_JOBS_PER_LAUNCH) {
        out.push_back(cur);
        cur.n = cur.max_count = 0;
      }
    }
    if (cur.n) out.push_back(cur);
  }
}

// A training loop re-packs its blobs after every optimizer step from the SAME parameter addresses: the launch tables of a
// (stage, sizes, parameter addresses) triple are built once -- two runs of the recipe, whose name lookups are a linear search over
// ~130 strings each, were ~0.1 ms of host time a call -- and re-used with the destination re-based onto the call's blob.
std::mutex& plan_mutex() {
  static std::mutex mu;
  return mu;
}
int get_pack_plan(int stage, int num_layers, int num_modes, const float* const* params, int n_params, PackPlan& plan_copy) {
  static std::vector<PackPlan> plans;
  {
    std::lock_guard<std::mutex> lk(plan_mutex());
    for (const PackPlan& pl : plans)
      if (pl.stage == stage && pl.nl == num_layers && pl.K == num_modes && int(pl.params.size()) == n_params &&
          std::equal(pl.params.begin(), pl.params.end(), params)) {
        plan_copy = pl;
        return TRAJSDE_OK;
      }
  }
  Packer dry{true};
  if (!run_recipe(dry, stage, num_layers, num_modes)) return fail(TRAJSDE_ERR_INVALID, "unknown stage");
  TS_REQUIRE(n_params == int(dry.names.size()), "pack_weights: parameter count does not match trajsde_param_count");
  for (int i = 0; i < n_params; ++i) TS_REQUIRE(params[i] != nullptr, "pack_weights: null parameter " + dry.names[i]);
  plan_copy = PackPlan();
  plan_copy.stage = stage; plan_copy.nl = num_layers; plan_copy.K = num_modes;
  plan_copy.params.assign(params, params + n_params);
  plan_copy.need_floats = trajsde_blob_floats(stage, num_layers, num_modes);
  for (int slot = 0; slot < 8; ++slot) {                // a stand-in range that no parameter lives in
    const uintptr_t lo = reinterpret_cast<uintptr_t>(pack_standin_blob(slot)), hi = lo + uintptr_t(plan_copy.need_floats) * sizeof(float);
    bool clash = false;
    for (int i = 0; i < n_params; ++i) {
      const uintptr_t u = reinterpret_cast<uintptr_t>(params[i]);
      clash = clash || (u + (uintptr_t(1) << 32) >= lo && u < hi);
    }
    plan_copy.standin = slot;
    if (!clash) break;
  }
  build_pack_tables(stage, num_layers, num_modes, dry.names, params, plan_copy.tables, plan_copy.jobs, plan_copy.standin);
  std::lock_guard<std::mutex> lk(plan_mutex());
  if (plans.size() >= 64) plans.clear();
  plans.push_back(plan_copy);
  return TRAJSDE_OK;
}
// every address inside the plan's stand-in blob -- destinations, and the sources of second-pass jobs that read first-pass results --
// moves onto `blob`
void rebase_job(PackJob& j, const PackPlan& pl, float* blob) {
  const uintptr_t F = reinterpret_cast<uintptr_t>(pack_standin_blob(pl.standin)), Fend = F + uintptr_t(pl.need_floats) * sizeof(float);
  auto rebase = [&](const float* q) -> const float* {
    const uintptr_t u = reinterpret_cast<uintptr_t>(q);
    return (u >= F && u < Fend) ? blob + (u - F) / sizeof(float) : q;
  };
  j.dst = const_cast<float*>(rebase(j.dst));
  j.src = rebase(j.src);
  j.src2 = rebase(j.src2);
  j.src3 = rebase(j.src3);
}

// number of jobs of a stage's recipe (independent of the addresses): the recipe run wet over made-up parameter addresses
int pack_job_count(int stage, int num_layers, int num_modes) {
  Packer dry{true};
  if (!run_recipe(dry, stage, num_layers, num_modes)) return -1;
  std::vector<const float*> fake(dry.names.size());
  for (size_t i = 0; i < fake.size(); ++i) fake[i] = reinterpret_cast<const float*>((uintptr_t(1) << 40) + (uintptr_t(i) << 28));
  Packer wet{false};
  wet.names = dry.names;
  wet.params = fake.data();
  wet.blob = pack_standin_blob(0);
  run_recipe(wet, stage, num_layers, num_modes);
  return int(wet.jobs.size());
}

// the merged table of several stages (trajsde_pack_weights_many), as last uploaded to `table_dev`
struct ManyPlan {
  std::vector<trajsde_pack_item> items;
  std::vector<std::vector<const float*>> params;
  void* table_dev = nullptr;
  std::vector<PackJob> table;              // [zero jobs | first-pass jobs | second-pass jobs]
  int n_zero = 0, n0 = 0, max0 = 0, n1 = 0, max1 = 0;
  uint64_t serial = 0;
  bool matches(const trajsde_pack_item* it, int n, void* dev) const {
    if (int(items.size()) != n || dev != table_dev) return false;
    for (int i = 0; i < n; ++i) {
      const trajsde_pack_item& a = items[i];
      if (a.stage != it[i].stage || a.num_layers != it[i].num_layers || a.num_modes != it[i].num_modes || a.n_params != it[i].n_params ||
          a.blob != it[i].blob || a.blob_floats != it[i].blob_floats)
        return false;
      if (!std::equal(params[i].begin(), params[i].end(), it[i].params)) return false;
    }
    return true;
  }
};
}  // namespace
extern "C" {

int trajsde_pack_weights(int stage, int num_layers, int num_modes, const float* const* params, int n_params,
                         float* blob, int64_t blob_floats, void* stream) {
  TS_REQUIRE(params && blob, "pack_weights: null pointer");
  TS_REQUIRE(n_params >= 0 && n_params < (1 << 20), "pack_weights: bad parameter count");
  PackPlan plan_copy;
  if (int rc = get_pack_plan(stage, num_layers, num_modes, params, n_params, plan_copy)) return rc;
  TS_REQUIRE(blob_floats >= plan_copy.need_floats, "pack_weights: blob too small");
  TS_HIP(hipMemsetAsync(blob, 0, size_t(blob_floats) * sizeof(float), static_cast<hipStream_t>(stream)));
  for (PackLaunch& t : plan_copy.tables) {
    for (int q = 0; q < t.n; ++q) rebase_job(t.tab.j[q], plan_copy, blob);
    k_pack_jobs<<<dim3(cdiv(t.max_count, 256), t.n), 256, 0, static_cast<hipStream_t>(stream)>>>(t.tab);
  }
  TS_LAUNCH_CHECK("pack");
  return TRAJSDE_OK;
}

int64_t trajsde_pack_many_table_bytes(const trajsde_pack_item* items, int n) {
  if (!items || n < 0 || n > 64) return -1;
  int64_t jobs = 0;
  for (int i = 0; i < n; ++i) {
    const int c = pack_job_count(items[i].stage, items[i].num_layers, items[i].num_modes);
    if (c < 0) return -1;
    jobs += c + 1;                                         // + the blob's zero job
  }
  return jobs * int64_t(sizeof(PackJob));
}

// The weight images of SEVERAL stages in three launches (zero the blobs | first-pass jobs | second-pass jobs) over ONE job table in
// device memory: a training step re-packs six blobs (three stages, forward and backward images) after every optimizer step, which
// through trajsde_pack_weights is 26 launches and 6 fills of ~5 us each.  The merged table is built when the arguments are first seen,
// written to `table_host` (pinned host memory of the caller) and copied to `table_dev` on `stream`; while the stage list, parameter
// addresses, blob addresses and `table_dev` stay the same, later calls only launch.  `fresh` != 0 forces the upload (the caller
// re-allocated or wrote `table_dev`).  Both tables: trajsde_pack_many_table_bytes(items, n) bytes, owned by the caller, not to be
// written by it while calls with these arguments continue.  Results: bit-identical to n calls of trajsde_pack_weights.
int trajsde_pack_weights_many(const trajsde_pack_item* items, int n, void* table_host, void* table_dev, int64_t table_bytes, int fresh,
                              void* stream) {
  TS_REQUIRE(items && table_host && table_dev, "pack_weights_many: null pointer");
  TS_REQUIRE(n > 0 && n <= 64, "pack_weights_many: bad stage count");
  for (int i = 0; i < n; ++i) {
    TS_REQUIRE(items[i].params && items[i].blob, "pack_weights_many: null pointer in an item");
    TS_REQUIRE(items[i].n_params >= 0 && items[i].n_params < (1 << 20), "pack_weights_many: bad parameter count");
    for (int k = 0; k < i; ++k) TS_REQUIRE(items[k].blob != items[i].blob, "pack_weights_many: two items share a blob");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  static std::mutex mu;
  static std::vector<ManyPlan> plans;
  static uint64_t next_serial = 1;
  static std::vector<std::pair<void*, uint64_t>> resident;       // table_dev -> serial of the plan it holds
  struct { int n_zero, n0, max0, n1, max1; uint64_t serial; int64_t bytes; } run{};
  bool upload = fresh != 0;
  auto note = [&](const ManyPlan& pl) {
    run.n_zero = pl.n_zero; run.n0 = pl.n0; run.max0 = pl.max0; run.n1 = pl.n1; run.max1 = pl.max1; run.serial = pl.serial;
    run.bytes = int64_t(pl.table.size()) * int64_t(sizeof(PackJob));
  };
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const ManyPlan& pl : plans)
      if (pl.matches(items, n, table_dev)) note(pl);
  }
  if (run.serial == 0) {
    ManyPlan mp;
    mp.items.assign(items, items + n);
    mp.table_dev = table_dev;
    std::vector<PackJob> zero, p0, p1;
    for (int i = 0; i < n; ++i) {
      const trajsde_pack_item& it = items[i];
      PackPlan pl;
      if (int rc = get_pack_plan(it.stage, it.num_layers, it.num_modes, it.params, it.n_params, pl)) return rc;
      TS_REQUIRE(it.blob_floats >= pl.need_floats, "pack_weights_many: blob too small");
      TS_REQUIRE(it.blob_floats < (int64_t(1) << 31), "pack_weights_many: blob too large");
      TS_REQUIRE((reinterpret_cast<uintptr_t>(it.blob) & 15) == 0, "pack_weights_many: blob must be 16-byte aligned");
      mp.params.emplace_back(it.params, it.params + it.n_params);
      mp.items[i].params = nullptr;                                   // (the key holds its own copy of the addresses)
      PackJob z{};
      z.count = int(it.blob_floats);
      z.dst = it.blob;
      zero.push_back(z);
      for (PackJob j : pl.jobs) {
        rebase_job(j, pl, it.blob);
        (j.pass == 0 ? p0 : p1).push_back(j);
      }
    }
    mp.n_zero = int(zero.size());
    mp.n0 = int(p0.size());
    mp.n1 = int(p1.size());
    for (const PackJob& j : p0) mp.max0 = j.count > mp.max0 ? j.count : mp.max0;
    for (const PackJob& j : p1) mp.max1 = j.count > mp.max1 ? j.count : mp.max1;
    mp.table = zero;
    mp.table.insert(mp.table.end(), p0.begin(), p0.end());
    mp.table.insert(mp.table.end(), p1.begin(), p1.end());
    TS_REQUIRE(mp.n0 < 65536 && mp.n1 < 65536, "pack_weights_many: too many jobs for one launch");
    std::lock_guard<std::mutex> lk(mu);
    mp.serial = next_serial++;
    if (plans.size() >= 16) plans.clear();
    plans.push_back(mp);
    note(mp);
    upload = true;
  }
  TS_REQUIRE(table_bytes >= run.bytes, "pack_weights_many: table too small (trajsde_pack_many_table_bytes)");
  {
    std::lock_guard<std::mutex> lk(mu);
    bool seen = false;
    for (auto& r : resident)
      if (r.first == table_dev) {
        seen = true;
        upload = upload || r.second != run.serial;
        r.second = run.serial;
      }
    if (!seen) {
      if (resident.size() >= 64) resident.clear();
      resident.emplace_back(table_dev, run.serial);
      upload = true;
    }
  }
  if (upload) {
    // (rare) `table_host` may still be the source of an earlier upload on this stream
    TS_HIP(hipStreamSynchronize(st));
    {
      std::lock_guard<std::mutex> lk(mu);
      const ManyPlan* src = nullptr;
      for (const ManyPlan& pl : plans)
        if (pl.serial == run.serial) src = &pl;
      TS_REQUIRE(src != nullptr, "pack_weights_many: plan evicted while in use");
      std::memcpy(table_host, src->table.data(), size_t(run.bytes));
    }
    TS_HIP(hipMemcpyAsync(table_dev, table_host, size_t(run.bytes), hipMemcpyHostToDevice, st));
  }
  const PackJob* tab = static_cast<const PackJob*>(table_dev);
  k_pack_zero_table<<<dim3(32, run.n_zero), 256, 0, st>>>(tab);
  if (run.n0) k_pack_jobs_table<<<dim3(cdiv(run.max0, 256), run.n0), 256, 0, st>>>(tab, run.n_zero);
  if (run.n1) k_pack_jobs_table<<<dim3(cdiv(run.max1, 256), run.n1), 256, 0, st>>>(tab, run.n_zero + run.n0);
  TS_LAUNCH_CHECK("pack_many");
  return TRAJSDE_OK;
}

}  // extern "C"
