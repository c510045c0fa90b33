// common.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <string>

#include "../../include/trajsde_hip.h"

namespace tsde {

std::string& last_error_ref();
int fail(int code, const std::string& msg);

#define TS_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return ::tsde::fail(TRAJSDE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
  } while (0)

#define TS_LAUNCH_CHECK(name)                                                                         \
  do {                                                                                                \
    hipError_t _e = hipGetLastError();                                                                \
    if (_e != hipSuccess)                                                                             \
      return ::tsde::fail(TRAJSDE_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(_e)); \
  } while (0)

#define TS_REQUIRE(cond, msg)                                                 \
  do {                                                                        \
    if (!(cond)) return ::tsde::fail(TRAJSDE_ERR_INVALID, std::string(msg));  \
  } while (0)

constexpr int MAX_DEVICES = 64;
inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < MAX_DEVICES) ? d : 0;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
// (addresses are formed in integer arithmetic: the size queries carve from a null base, and pointer arithmetic on a null pointer
//  is undefined behaviour -- found by the sanitizer build of the host side, tests/test_cabi_cpu.py)
struct Carver {
  uintptr_t base;
  int64_t size, off;
  bool ok;
  Carver(void* p, int64_t n) : base(reinterpret_cast<uintptr_t>(p)), size(n), off(0), ok(true) {}
  template <typename T>
  T* take(int64_t count) {
    off = align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + uintptr_t(off));
    off += count * int64_t(sizeof(T));
    if (base != 0 && off > size) ok = false;
    return p;
  }
};

// p + n in integer arithmetic (workspace layouts are also computed from a null base by the size queries)
template <typename T>
inline T* ptr_add(T* p, int64_t n) { return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + uintptr_t(n) * sizeof(T)); }

inline int cdiv(int64_t a, int64_t b) { return int((a + b - 1) / b); }
// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin to the 8 XCDs (workgroup b runs on XCD b % 8) and
// each XCD has its own 4 MB L2.  Kernels whose neighbouring workgroups read the same rows (the targets of one scene gather
// the same node rows) launch xcd_grid(blocks) workgroups and work on logical block xcd_block(): XCD x then owns the contiguous
// range [x * per, (x + 1) * per), so a scene's rows live in one L2 instead of being streamed through all eight.
inline int xcd_grid(int64_t blocks) { return 8 * cdiv(blocks, 8); }
#ifdef __HIPCC__
__device__ __forceinline__ int64_t xcd_block() { return int64_t(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
#endif

// grid for a tile kernel whose workgroups each hold an LDS weight image: as many workgroups as fit on the
// chip at once (LDS- and thread-limited), grid-stride beyond that, never more than the work needs
// same, for kernels that flush per-wave vector-gradient partials: capped so that the partial slab stays small
inline int tile_grid(int64_t ntiles, int threads, int lds_bytes);
inline int env_threads(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v) return dflt;
  const int t = atoi(v);
  return (t >= 64 && t <= 1024 && t % 64 == 0) ? t : dflt;
}
inline int vec_grid(int64_t ntiles, int threads, int lds_bytes) {
  const int g = tile_grid(ntiles, threads, lds_bytes);
  return g > 512 ? 512 : g;
}
inline int tile_grid(int64_t ntiles, int threads, int lds_bytes) {
  const int waves = threads / 64;
  int per_cu = int((160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1));
  const int by_threads = 2048 / threads;
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu < 1) per_cu = 1;
  int64_t want = (ntiles + waves - 1) / waves;
  const int64_t cap = 256 * int64_t(per_cu);
  if (want > cap) want = cap;
  return int(want < 1 ? 1 : want);
}


// storage of the [rows][64] activations that stay inside a stage (trajsde_state_storage): false = fp32, true = bf16
bool state_bf16();

// ---- optional per-kernel timing with HIP events on the launch stream (trajsde_profile_mode / _report).
// mode 0: off (no events are created or recorded); 1: only launches tagged as "dominant"; 2: every launch.
int profile_mode();
void profile_begin(const char* tag, hipStream_t st, bool dominant);
void profile_end(const char* tag, hipStream_t st, bool dominant);
struct ProfScope {
  const char* tag;
  hipStream_t st;
  bool dom, on;
  ProfScope(const char* t, hipStream_t s, bool dominant = false) : tag(t), st(s), dom(dominant) {
    const int m = profile_mode();
    on = m == 2 || (m == 1 && dominant);
    if (on) profile_begin(tag, st, dom);
  }
  ~ProfScope() {
    if (on) profile_end(tag, st, dom);
  }
};

// launch a kernel that needs `lds` bytes of dynamic LDS.  The >64 KB opt-in is raised whenever a call site asks for more
// than it has asked before ON THE CURRENT DEVICE (a kernel whose LDS size depends on the batch, e.g. k_enc_recur_coop, must
// not stay latched at its first call's size; the attribute is per device).  `_attr_lds[dev]` holds bytes + 1, 0 = unset.
#define TS_LAUNCH_TAG(tag, dominant, kern, grid, threads, lds, st, ...)                                           \
  do {                                                                                                            \
    static std::atomic<int> _attr_lds[::tsde::MAX_DEVICES];                                                       \
    const int _dev = ::tsde::current_device();                                                                    \
    if (int(lds) + 1 > _attr_lds[_dev].load(std::memory_order_relaxed)) {                                         \
      TS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));       \
      _attr_lds[_dev].store(int(lds) + 1, std::memory_order_relaxed);                                             \
    }                                                                                                             \
    {                                                                                                             \
      ::tsde::ProfScope _ps(tag, st, dominant);                                                                   \
      kern<<<grid, threads, lds, st>>>(__VA_ARGS__);                                                              \
    }                                                                                                             \
    TS_LAUNCH_CHECK(tag);                                                                                         \
  } while (0)
#define TS_LAUNCH(kern, grid, threads, lds, st, ...) TS_LAUNCH_TAG(#kern, false, kern, grid, threads, lds, st, __VA_ARGS__)

}  // namespace tsde
