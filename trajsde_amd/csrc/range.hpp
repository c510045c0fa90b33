// range.hpp -- the fp16 range guard of the split-precision products (tile.hpp).
//
// fp16x3 represents an fp32 operand as two fp16 pieces.  That is fp32-accurate for magnitudes in roughly [6e-5 * 2^-11,
// 65504]: below, the low piece underflows and the product degrades gracefully to an ABSOLUTE error of ~6e-8 per operand
// (harmless at the path's 1e-4 tolerance); above 65504 the high piece saturates and the result would be silently wrong.
// Most operands of the path are outputs of a LayerNorm, a tanh or a sigmoid and cannot get there.  The ones that can are the
// data-dependent, unnormalised tensors: the SDE states (integrated noise), the rows entering the recurrence from the
// attention block, the decoder's embedding inputs, the attention aggregates and the FFN hidden units.  Each kernel that
// feeds such a tensor to a split product notes its magnitude here -- one v_max3 per two values and one compare per tile --
// and a value >= 65504 (or a NaN) sets a sticky bit in a per-device flag word.  The host reads the word at its natural
// synchronisation points through trajsde_range_status(): TRAJSDE_ERR_UNSUPPORTED instead of a silently saturated result.
// Weights are checked once, when their fp16 images are packed.
//
// The flag is a `static __device__` word per translation unit (no relocatable device code needed); every unit that includes
// this header registers a reader for its word with the registry in pack.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "tile.hpp"

namespace tsde {

enum RangeSite : unsigned {
  RS_DEC_STATE = 0,    // decoder SDE state y (DEC:88 solve)
  RS_DEC_INPUT = 1,    // local / global embedding rows entering aggr_embed / pi (DEC:82, 93)
  RS_ENC_STATE = 2,    // encoder latent state h (ENC:140-182)
  RS_ENC_INPUT = 3,    // aa_out rows entering the GRU (ENC:176)
  RS_NODE_AGG = 4,     // attention aggregate / gated update entering lin_ih, out_proj (ENC:595-609, AGG:119-131)
  RS_FFN_HIDDEN = 5,   // ReLU(mlp.0(..)) entering mlp.3
  RS_WEIGHT = 6,       // a weight whose fp16 image would overflow
  RS_SITES = 7
};
constexpr float FP16_SPLIT_LIMIT = 65504.0f;

static __device__ unsigned g_range_flag;     // one word per translation unit; bit = RangeSite

__device__ __forceinline__ float absmax4(const f4& a) { return fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))); }
template <int JT>
__device__ __forceinline__ float absmax(const f4 (&a)[JT]) {
  float m = 0.f;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) m = fmaxf(m, absmax4(a[jt]));
  return m;
}
// `m`: the lane's largest magnitude of a tensor about to be split.  `!(m < limit)` is also true for a NaN (fmaxf drops
// NaNs, so a NaN operand is caught by the products' consumers, not here -- this guard is about saturation).
__device__ __forceinline__ void range_note(float m, RangeSite site) {
#if TSDE_SPLIT_H3
  if (__builtin_expect(!(m < FP16_SPLIT_LIMIT), 0)) atomicOr(&g_range_flag, 1u << unsigned(site));
#else
  (void)m; (void)site;                     // bf16 pieces have fp32's exponent range: nothing to guard
#endif
}

// host side: registry of the per-unit flag words (defined in pack.hip)
using RangeReader = hipError_t (*)(unsigned* out, bool reset, hipStream_t st);
void register_range_reader(RangeReader r);
std::vector<RangeReader>& range_readers();

static hipError_t range_reader_of_this_unit(unsigned* out, bool reset, hipStream_t st) {
  hipError_t e = hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(g_range_flag), sizeof(unsigned), 0, hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess || !reset || *out == 0) return e;
  const unsigned zero = 0;
  e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_range_flag), &zero, sizeof(unsigned), 0, hipMemcpyHostToDevice, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}
static const int g_range_reader_registered = (register_range_reader(&range_reader_of_this_unit), 0);

}  // namespace tsde
