// optim.hip -- the end of a training step in two launches: the stages' gradient buffers gathered into the training loop's flat
// gradient tensor, and AdamW (MODEL:204-207: torch.optim.AdamW(lr, weight_decay), defaults otherwise) over the flat parameter tensor.
// Both are element-wise and HBM-bound (2 MB of parameters: 5 us); what they replace is ~14 element-wise launches of ~5 us each and
// their ~0.2 ms of host time (profiles/r05_train_step_timeline.txt).
#include "common.hpp"

namespace tsde {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int GATHER_MAX_ITEMS = 8;
struct GatherItems {
  trajsde_gather_item it[GATHER_MAX_ITEMS];
};

// dst[i] += src[index[i]] * (scale * mult): grid.y = item (product and sum rounded separately, as torch's addcmul_ with value 1)
__global__ __launch_bounds__(256) void k_grad_gather_add(const GatherItems items, const float* __restrict__ scale) {
#pragma clang fp contract(off)
  const trajsde_gather_item& J = items.it[blockIdx.y];
  const float s = (scale ? scale[0] : 1.f) * J.mult;
  const int64_t n = J.n;
  const int64_t* __restrict__ idx = J.index;
  const float* __restrict__ src = J.src;
  float* __restrict__ dst = J.dst;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const float t = src[idx[i]] * s;
    dst[i] = dst[i] + t;
  }
}

// torch.optim.AdamW's single-tensor update, operation by operation (torch/optim/adamw.py _single_tensor_adamw, amsgrad = maximize =
// False), with its scalars formed by the caller the way torch forms them (Python floats, rounded to fp32 where a tensor op takes them):
//   param.mul_(1 - lr * weight_decay)                                    decay
//   exp_avg.lerp_(grad, 1 - beta1)                                       w1   (|w| < 0.5: a + w (b - a), ATen/native/Lerp.h)
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)         beta2, w2
//   denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2^step)).add_(eps)         bias2, eps  (torch divides a tensor by a host scalar as a
//                                                                        product with the scalar's reciprocal, formed in double and
//                                                                        rounded to fp32: `divide` = 0, bias2 = that reciprocal)
// The multi-tensor form (`foreach=True`, what AdamW(model.parameters()) runs on a GPU: MODEL:205) differs in ONE operation: its
// _foreach_div_ by the scalar list is a true division (`divide` = 1, bias2 = sqrt(1 - beta2^step) itself).  Measured on this torch,
// element by element over 2e5 values: both forms reproduced exactly (tests/test_gpu_step_launches.py).
//   param.addcdiv_(exp_avg, denom, value=-(lr / (1 - beta1^step)))       neg_step
struct AdamScalars {
  float decay, w1, beta2, w2, bias2, eps, neg_step;
};
// (each torch op rounds its result to fp32; inside one op the multiply-add is fused, as hipcc contracts it in torch's kernels.  Written
//  with contraction OFF and the fused operations spelled out: the __f*_rn spellings are plain operators to this compiler and were
//  contracted across the op boundaries; __fsqrt_rn is the bare 1-ulp v_sqrt_f32, sqrtf the correctly rounded sequence torch uses)
template <bool DIVIDE>
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamScalars& c) {
#pragma clang fp contract(off)
  p = p * c.decay;
  const float diff = g - m;
  m = c.w1 < 0.5f ? __builtin_fmaf(c.w1, diff, m) : __builtin_fmaf(-diff, 1.f - c.w1, g);
  v = v * c.beta2;
  const float gg = g * g;
  v = __builtin_fmaf(c.w2, gg, v);                                     // addcmul: a + value * (b * c)
  float denom = DIVIDE ? sqrtf(v) / c.bias2 : sqrtf(v) * c.bias2;
  denom = denom + c.eps;
  const float q = m / denom;
  p = __builtin_fmaf(c.neg_step, q, p);
}
template <bool DIVIDE>
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                               float* __restrict__ exp_avg_sq, int64_t n, AdamScalars c, int vec) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (vec) {
    const int64_t i = 4 * t;
    if (i + 3 < n) {
      f4 p = *reinterpret_cast<const f4*>(param + i), m = *reinterpret_cast<const f4*>(exp_avg + i), v = *reinterpret_cast<const f4*>(exp_avg_sq + i);
      const f4 g = *reinterpret_cast<const f4*>(grad + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adamw_one<DIVIDE>(pk, g[k], mk, vk, c);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      *reinterpret_cast<f4*>(param + i) = p;
      *reinterpret_cast<f4*>(exp_avg + i) = m;
      *reinterpret_cast<f4*>(exp_avg_sq + i) = v;
    } else {
      for (int64_t j = i; j < n; ++j) adamw_one<DIVIDE>(param[j], grad[j], exp_avg[j], exp_avg_sq[j], c);
    }
  } else if (t < n) {
    adamw_one<DIVIDE>(param[t], grad[t], exp_avg[t], exp_avg_sq[t], c);
  }
}

}  // namespace tsde

using namespace tsde;

extern "C" {

int trajsde_grad_gather_add(const trajsde_gather_item* items, int n_items, const float* scale, void* stream) {
  TS_REQUIRE(items, "grad_gather_add: null pointer");
  TS_REQUIRE(n_items > 0 && n_items <= GATHER_MAX_ITEMS, "grad_gather_add: 1 .. 8 items a call");
  GatherItems tab;
  int64_t longest = 0;
  for (int i = 0; i < n_items; ++i) {
    TS_REQUIRE(items[i].n >= 0, "grad_gather_add: negative length");
    TS_REQUIRE(items[i].n == 0 || (items[i].dst && items[i].src && items[i].index), "grad_gather_add: null pointer in an item");
    tab.it[i] = items[i];
    longest = items[i].n > longest ? items[i].n : longest;
  }
  if (longest == 0) return TRAJSDE_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t blocks = (longest + 255) / 256;
  TS_LAUNCH(k_grad_gather_add, dim3(unsigned(blocks > 2048 ? 2048 : blocks), n_items), 256, 0, st, tab, scale);
  return TRAJSDE_OK;
}

int trajsde_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float decay, float w1,
                       float beta2, float w2, float bias2, int divide, float eps, float neg_step, void* stream) {
  TS_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adamw_step: null pointer");
  TS_REQUIRE(n >= 0, "adamw_step: negative length");
  TS_REQUIRE(bias2 > 0.f && bias2 < 3.0e38f, "adamw_step: the bias-correction scalar must be positive and finite (step >= 1)");
  if (n == 0) return TRAJSDE_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uintptr_t all = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                        reinterpret_cast<uintptr_t>(exp_avg_sq);
  const int vec = (all & 15) == 0;
  const int64_t threads = vec ? (n + 3) / 4 : n;
  const AdamScalars c{decay, w1, beta2, w2, bias2, eps, neg_step};
  if (divide) TS_LAUNCH(k_adamw<true>, dim3(unsigned((threads + 255) / 256)), 256, 0, st, param, grad, exp_avg, exp_avg_sq, n, c, vec);
  else TS_LAUNCH(k_adamw<false>, dim3(unsigned((threads + 255) / 256)), 256, 0, st, param, grad, exp_avg, exp_avg_sq, n, c, vec);
  return TRAJSDE_OK;
}

}  // extern "C"
