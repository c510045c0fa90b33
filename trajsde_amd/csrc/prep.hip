// prep.hip -- everything the reference does with index tensors before the attention blocks, on device:
//   MODEL:75-85   rotate_mat, y @ rotate_mat
//   ENC:68-71     lane feature = last valid lane point - first lane point
//   ENC:73-74,103 nus_mask;  ENC:88-103 fake copies of the target agents (x + 2*randn, duplicated in-edges)
//   ENC:107-118   21 x subgraph(valid at t) + DistanceDropEdge(radius) (UTIL:83-92)
//   AGG:41-51     subgraph(valid at t=H-1), relative pose;   ENC:198 radius drop on lane-actor edges
// Instead of materialising 21 boolean-masked edge lists, edges are sorted once by target (counting sort
// -> CSR), every (t, edge) candidate gets a 1-byte flag, one prefix sum gives the compacted position
// of every survivor, and the segment pointer of snapshot node (t, i) is read off the same prefix sum.
// The compacted lists carry the pre-rotated 2-d geometry the edge kernels consume (16 B per edge).

#include <algorithm>
#include <cmath>
#include <limits>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "philox.hpp"
#include "tile.hpp"

namespace tsde {

__global__ void k_rotate(const float* __restrict__ ang, int N, const float* __restrict__ y, int F,
                         float* __restrict__ rot, float* __restrict__ y_rot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float s = sinf(ang[i]), c = cosf(ang[i]);
  f4 r = {c, -s, s, c};
  *reinterpret_cast<f4*>(rot + 4 * i) = r;
  if (y != nullptr)
    for (int t = 0; t < F; ++t) {
      const float a = y[(int64_t(i) * F + t) * 2], b = y[(int64_t(i) * F + t) * 2 + 1];
      y_rot[(int64_t(i) * F + t) * 2] = a * c + b * s;        // [a b] @ [[c,-s],[s,c]]
      y_rot[(int64_t(i) * F + t) * 2 + 1] = b * c - a * s;
    }
}

// ---------------------------------------------------------------------------------------------- CSR by target
// counting sort on the target id (deterministic counts via integer atomics), then every row is put into canonical
// order (ascending source / edge id) by an in-LDS bitonic sort, so the result does not depend on atomic arrival
// order -- nor on the order of the input edge list.
__global__ void k_degree(const int64_t* __restrict__ ei, int E, int32_t* __restrict__ deg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) atomicAdd(&deg[int(ei[int64_t(E) + e])], 1);                 // row 1 = target
}
// a row's slots are handed out by counting its degree back down (any order: the rows are sorted afterwards), so no copy of the
// row pointers is needed as a cursor
__global__ void k_scatter(const int64_t* __restrict__ ei, int E, const int32_t* __restrict__ rowptr, int32_t* __restrict__ deg,
                          int32_t* __restrict__ out, int store_edge_id) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int d = int(ei[int64_t(E) + e]);
  const int p = rowptr[d] + atomicSub(&deg[d], 1) - 1;
  out[p] = store_edge_id ? e : int32_t(ei[e]);                             // row 0 = source
}
// lane-actor edges: value = (lane id << 32) | edge id, so that rows sort by lane first (canonical under permutations
// of the input list) and the edge id is still at hand for the vector lookup
__global__ void k_scatter_lane(const int64_t* __restrict__ lai, int E, const int32_t* __restrict__ rowptr, int32_t* __restrict__ deg,
                               int64_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int d = int(lai[int64_t(E) + e]);
  const int p = rowptr[d] + atomicSub(&deg[d], 1) - 1;
  out[p] = (lai[e] << 32) | int64_t(e);
}
// ascending sort of the long CSR rows (more than 256 entries; second phase of k_row_sort), one workgroup per row (grid-stride).
// Rows of up to 4096 entries: bitonic network in LDS on the row padded to a power of two with +inf.  Longer rows: in place in global memory with the "flip" form of the network
// (first sub-step of stage k pairs i with i ^ (k - 1), the others i with i ^ j), in which EVERY compare-exchange puts the
// minimum at the lower index -- so the virtual +inf padding above the row's end never moves and a pair whose upper index lies
// beyond the row is a no-op: correct for any row length without materialising the padding.
template <typename T>
__device__ __forceinline__ void sort_long_rows(const int32_t* __restrict__ rowptr, int n_rows, T* __restrict__ vals, T* buf /* LDS, 4096 */,
                                               int32_t* __restrict__ low_out, int32_t* __restrict__ row_out) {
  const T INF = sizeof(T) == 8 ? T(INT64_MAX) : T(INT32_MAX);
  for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const int beg = rowptr[row], n = rowptr[row + 1] - beg;
    if (n <= 256) continue;                                 // (sorted by a wave in the first phase)
    if (row_out != nullptr)
      for (int i = threadIdx.x; i < n; i += blockDim.x) row_out[beg + i] = row;
    int P = 2;
    while (P < n) P <<= 1;
    if (P <= 4096) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) buf[i] = i < n ? vals[beg + i] : INF;
      __syncthreads();
      for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = threadIdx.x; i < P; i += blockDim.x) {
            const int l = i ^ j;
            if (l > i) {
              const T vi = buf[i], vl = buf[l];
              const bool up = (i & k) == 0;
              if ((vi > vl) == up) {
                buf[i] = vl;
                buf[l] = vi;
              }
            }
          }
          __syncthreads();
        }
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        vals[beg + i] = buf[i];
        if (low_out != nullptr) low_out[beg + i] = int32_t(int64_t(buf[i]) & 0xFFFFFFFFll);
      }
      __syncthreads();
    } else {
      T* a = vals + beg;
      for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          const int mask = (j == (k >> 1)) ? k - 1 : j;      // flip on the first sub-step of a stage
          for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int l = i ^ mask;
            if (l > i && l < n) {
              const T vi = a[i], vl = a[l];
              if (vi > vl) {
                a[i] = vl;
                a[l] = vi;
              }
            }
          }
          __threadfence_block();
          __syncthreads();
        }
      if (low_out != nullptr)
        for (int i = threadIdx.x; i < n; i += blockDim.x) low_out[beg + i] = int32_t(int64_t(a[i]) & 0xFFFFFFFFll);
    }
  }
}

// Ascending sort of every CSR row (canonical order: the result does not depend on the atomics' arrival order nor on the order of
// the input list).  Rows of up to 256 entries -- every row of the usual batches -- are sorted by ONE WAVE in registers: element
// i = 4 lane + r, so the compare-exchanges at distance 1 and 2 are register swaps and the others one cross-lane exchange per
// register; no LDS image, no workgroup barrier per sub-step (the workgroup form spends 36 barriers on a 256-entry row: 47 us
// for the 8 192 rows of the metric workload against 12 us here).  Longer rows: second phase, sort_long_rows.
template <typename T>
__device__ __forceinline__ T lane_xor(T v, int mask) {
  if constexpr (sizeof(T) == 8) {
    const unsigned lo = unsigned(__shfl_xor(int(unsigned(v & 0xFFFFFFFFll)), mask)), hi = unsigned(__shfl_xor(int(unsigned((v >> 32) & 0xFFFFFFFFll)), mask));
    return T((int64_t(hi) << 32) | int64_t(lo));
  } else {
    return T(__shfl_xor(int(v), mask));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_row_sort(const int32_t* __restrict__ rowptr, int n_rows, T* __restrict__ vals,
                                                  int32_t* __restrict__ low_out /* or null: the low words of the sorted values (edge ids of packed lane keys) */,
                                                  int32_t* __restrict__ row_out /* or null: the row of every position (its target) */) {
  __shared__ T buf[4096];
  constexpr int EPL = 4, PMAX = 64 * EPL;
  const T INF = sizeof(T) == 8 ? T(INT64_MAX) : T(INT32_MAX);
  const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
  for (int row = blockIdx.x * waves + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * waves) {
    const int beg = rowptr[row], n = rowptr[row + 1] - beg;
    if (n > PMAX) continue;
    if (row_out != nullptr) {
#pragma unroll
      for (int r = 0; r < EPL; ++r)
        if (64 * r + lane < n) row_out[beg + 64 * r + lane] = row;
    }
    if (n <= 1) {
      if (n == 1 && low_out != nullptr && lane == 0) low_out[beg] = int32_t(int64_t(vals[beg]) & 0xFFFFFFFFll);
      continue;
    }
    T v[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) v[r] = EPL * lane + r < n ? vals[beg + EPL * lane + r] : INF;
#pragma unroll
    for (int k = 2; k <= PMAX; k <<= 1) {
      if ((k >> 1) >= n) break;                             // the row's power of two is done: the padding above it never moves
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        if (j < EPL) {                                      // partner in this lane
#pragma unroll
          for (int r = 0; r < EPL; ++r) {
            const int l = r ^ j;
            if (l > r) {
              const bool up = ((EPL * lane + r) & k) == 0;
              const T a = v[r], b = v[l];
              const bool sw = (a > b) == up;
              v[r] = sw ? b : a;
              v[l] = sw ? a : b;
            }
          }
        } else {                                            // partner in lane ^ (j / EPL), same register
          const int lj = j / EPL;
          const bool lower = (lane & lj) == 0;
#pragma unroll
          for (int r = 0; r < EPL; ++r) {
            const T o = lane_xor<T>(v[r], lj);
            const bool up = ((EPL * lane + r) & k) == 0;
            const bool keep_min = lower == up;
            v[r] = keep_min ? (o < v[r] ? o : v[r]) : (o > v[r] ? o : v[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < EPL; ++r)
      if (EPL * lane + r < n) {
        vals[beg + EPL * lane + r] = v[r];
        if (low_out != nullptr) low_out[beg + EPL * lane + r] = int32_t(int64_t(v[r]) & 0xFFFFFFFFll);
      }
  }
  sort_long_rows<T>(rowptr, n_rows, vals, buf, low_out, row_out);    // rows of more than 256 entries, a workgroup per row
}

// per extended node: original actor, source mask, recurrence iteration to keep; slots of the agent rows
__device__ __forceinline__ void ext_nodes_body(int i, int N, int A, int H, const int64_t* __restrict__ agent_index,
                                               const int64_t* __restrict__ batch, const int64_t* __restrict__ source,
                                               const uint8_t* __restrict__ bos, int32_t* __restrict__ orig, uint8_t* __restrict__ nus,
                                               int32_t* __restrict__ eos, int32_t* __restrict__ pick_slot) {
  if (i >= N + A) return;
  const int o = i < N ? i : int(agent_index[i - N]);
  orig[i] = o;
  nus[i] = (i < N ? source[batch[i]] : source[i - N]) == 0;        // ENC:73-74, 103
  int first = 0;                                                     // torch.argmax of an all-false row is 0
  for (int t = H - 1; t >= 0; --t)
    if (bos[int64_t(o) * H + t]) first = t;
  eos[i] = (H - 1) - first;                                          // ENC:187 (ref_time = H-1)
  pick_slot[i] = i < N ? -1 : A + (i - N);
}
__global__ void k_agent_slots(int A, const int64_t* __restrict__ agent_index, int32_t* __restrict__ pick_slot) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < A) pick_slot[agent_index[k]] = k;
}

// x_fake[k,t,:] = x[agent_k,t,:] + 2*z   (ENC:94-95)
__device__ __forceinline__ void fake_x_body(int idx /* one thread per (k, quad of 4 columns) */, int A, int H, const float* __restrict__ x,
                                            const int64_t* __restrict__ agent_index, const NoiseArg& na, float* __restrict__ x_fake) {
  const int quads = (2 * H + 3) / 4;
  if (idx >= A * quads) return;
  const int k = idx / quads, q = idx % quads;
  f4 z;
  if (na.z != nullptr) {
    for (int c = 0; c < 4; ++c) z[c] = (4 * q + c < 2 * H) ? na.z[int64_t(k) * 2 * H + 4 * q + c] : 0.f;
  } else {
    z = philox_normal4(noise_key(na), STREAM_FAKE_AGENT, 0u, na.row_ids ? uint32_t(na.row_ids[k]) : uint32_t(k), uint32_t(q));
  }
  const int64_t a = agent_index[k];
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * q + c;
    if (col < 2 * H) x_fake[int64_t(k) * 2 * H + col] = x[a * 2 * H + col] + 2.0f * z[c];
  }
}

// ---------------------------------------------------------------------------------------------- 21 snapshots
// The agent-agent list of snapshot t keeps the in-edge (j -> i) when both ends are valid at t (ENC:108) and closer than the
// radius (UTIL:88).  Three steps, none of which evaluates a candidate twice or leaves lanes idle on non-survivors:
//   k_aa_ballots   a wave tests 64 consecutive CSR positions (in-edges of REAL actors) at every t and stores, per t, the 64-bit
//                  ballot of the survivors -- 8 H bytes per 64 candidates x H.  The rows of the fake agents alias their actors'
//                  (same positions, same padding), so their candidates are the same bits.
//   k_aa_count     one thread per snapshot node (t, i): popcounts of its row's stretch of the ballots -> segment lengths; a prefix
//                  sum (k_scan_chained) turns them into the segment pointers.
//   k_aa_fill      one wave per extended node: for every t the survivors of the row are handed to the lanes by rank (lane r takes the
//                  r-th set bit of the row's ballots), so a segment's records are computed and written 64 at a time, contiguously
//                  -- the pass is as long as the output, not as long as the candidate list.
// Results are identical to evaluating every (t, in-edge) candidate in row order: same predicate, same order (ascending CSR
// position = ascending sender), same arithmetic for the geometry.
// The reference's test is  sqrt(dx^2 + dy^2) < radius  in float32 (UTIL:88).  sqrtf is correctly rounded and monotone, so that is
// exactly  dx^2 + dy^2 < T  with T the smallest float whose square root reaches the radius (radius2_threshold, host): same
// survivors for every input, without a 25-instruction IEEE square root per candidate.
// The sum of squares is formed the way torch.norm forms it on the reference's side (measured: acc = fl(dx * dx), then
// fma(dy, dy, acc) -- 0 mismatches in 2 M random pairs, against 8 % for the other contraction order), spelled out so that the
// compiler's choice of contraction cannot move a borderline pair.
__device__ __forceinline__ float norm2_sq(float dx, float dy) { return fmaf(dy, dy, __fmul_rn(dx, dx)); }
__device__ __forceinline__ bool within_radius2(float dx, float dy, float thr2) { return norm2_sq(dx, dy) < thr2; }
static float radius2_threshold(float radius) {
  if (!(radius > 0.f)) return 0.f;                                       // sqrt(x) < r <= 0 never holds (x >= 0)
  if (std::isinf(radius)) return radius;
  float t = radius * radius;
  if (std::isinf(t)) t = std::numeric_limits<float>::max();
  while (t > 0.f && std::sqrt(t) >= radius) t = std::nextafter(t, 0.f);                                            // sqrt(t) < radius
  while (std::sqrt(t) < radius) t = std::nextafter(t, std::numeric_limits<float>::infinity());                      // smallest t with sqrt(t) >= radius
  return t;
}

// vmask[i]: bit t set when actor i is valid (not padded) at history step t
__device__ __forceinline__ void valid_mask_body(int i, int N, int H, int TT, const uint8_t* __restrict__ pad, uint32_t* __restrict__ vmask) {
  if (i >= N) return;
  uint32_t m = 0;
  uint8_t v[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) v[t] = pad[int64_t(i) * TT + (t < H ? t : H - 1)];     // all loads in flight (H <= 32)
#pragma unroll
  for (int t = 0; t < 32; ++t) m |= uint32_t(t < H && !v[t]) << t;
  vmask[i] = m;
}

// A wave owns 64 consecutive CSR positions.  Phase 1, lane = (candidate, t) the way the rows lie in memory: the senders' position
// rows are read coalesced (64 / H rows of 8 H bytes per instruction; one lane per candidate looping over t would make every load
// a 64-line gather) and parked in wave-private LDS.  Phase 2, lane = candidate: for every t the lane tests its pair (sender row
// from LDS, target row from the one or two rows the block's candidates share) and the wave's ballot IS the stored word.
__global__ __launch_bounds__(256) void k_aa_ballots(int E, int H, int TT, const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                                                    const uint32_t* __restrict__ vmask, const float* __restrict__ pos, float thr2,
                                                    unsigned long long* __restrict__ bal) {
  extern __shared__ __attribute__((aligned(16))) float2 rows[];                            // [4 waves][64 rows][H | 1]: odd stride, 2-way LDS conflicts at most
  const int AA_ROW = H | 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t blk = xcd_block() * (blockDim.x >> 6) + wv;                               // 64 consecutive CSR positions (xcd_grid launch)
  if (blk * 64 >= E) return;                                                               // (uniform)
  const int per = 64 / H, j = lane / H, t = lane - j * H;
  const float2* pos2 = reinterpret_cast<const float2*>(pos);
  const int64_t pl = blk * 64 + lane;
  const bool live = pl < E;
  const int s_all = live ? csr_src[pl] : 0, o_all = live ? csr_dst[pl] : 0;
  const uint32_t m = live ? (vmask[s_all] & vmask[o_all]) : 0u;
  float2* mine_rows = rows + wv * 64 * AA_ROW;
  for (int q0 = 0; q0 < 64; q0 += 8 * per) {                                               // eight row groups in flight
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * per + j;
      const int su = __shfl(s_all, q & 63);
      v[u] = pos2[int64_t(su) * TT + t];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * per + j;
      if (j < per && q < 64) mine_rows[q * AA_ROW + t] = v[u];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float2* pd = pos2 + int64_t(o_all) * TT;
  unsigned long long mine = 0;
  for (int t0 = 0; t0 < H; t0 += 8) {
    float2 d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = pd[t0 + u < H ? t0 + u : H - 1];                     // one or two distinct rows per wave: broadcast
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int tt = t0 + u;
      if (tt >= H) break;                                                                  // (uniform)
      const float2 q = mine_rows[lane * AA_ROW + tt];
      const bool ok = ((m >> tt) & 1u) && within_radius2(q.x - d[u].x, q.y - d[u].y, thr2);
      const unsigned long long B = __ballot(ok);
      if (lane == tt) mine = B;
    }
  }
  if (lane < H) bal[blk * H + lane] = mine;
}

// bits [lo, hi) of a 64-bit word
__device__ __forceinline__ unsigned long long bit_range(int lo, int hi) {
  const unsigned long long upto_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
  return upto_hi & ~((1ull << lo) - 1ull);
}
// the part of ballot block b that belongs to the CSR row [beg, end)
__device__ __forceinline__ unsigned long long row_part(int b, int beg, int end) {
  const int lo = beg > 64 * b ? beg - 64 * b : 0, hi = end < 64 * b + 64 ? end - 64 * b : 64;
  return bit_range(lo, hi);
}

__global__ void k_aa_count(int Nt, int H, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ orig,
                           const unsigned long long* __restrict__ bal, int32_t* __restrict__ cnt) {
  const int64_t id = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;                       // = t * Nt + node
  if (id >= int64_t(H) * Nt) return;
  if (id == 0) cnt[int64_t(H) * Nt] = 0;                                                   // the scan's extra element (total at the end)
  const int t = int(id / Nt), node = int(id - int64_t(t) * Nt), o = orig[node];
  const int beg = rowptr[o], end = rowptr[o + 1];
  int c = 0;
  if (end > beg)
    for (int b = beg >> 6; b <= (end - 1) >> 6; ++b) c += __popcll(bal[int64_t(b) * H + t] & row_part(b, beg, end));
  cnt[id] = c;
}

// index of the r-th (0-based) set bit of w (r < popcount(w))
__device__ __forceinline__ int nth_set_bit(unsigned long long w, int r) {
  int posn = 0;
#pragma unroll
  for (int width = 32; width >= 1; width >>= 1) {
    const int c = __popcll((w >> posn) & ((1ull << width) - 1ull));
    if (r >= c) { r -= c; posn += width; }
  }
  return posn;
}

__global__ __launch_bounds__(256) void k_aa_fill(int N, int Nt, int H, int TT, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ csr_src, const int32_t* __restrict__ orig,
                                                 const unsigned long long* __restrict__ bal, const int32_t* __restrict__ segptr,
                                                 const float* __restrict__ pos, const float* __restrict__ x, const float* __restrict__ rot,
                                                 int32_t* __restrict__ aa_dst, int32_t* __restrict__ aa_src, float* __restrict__ geom) {
  // one wave per extended node; for every t its survivors are handed out to the lanes by rank (lane r takes the r-th set bit of
  // the row's ballots), so the records of segment (t, node) are written 64 at a time, contiguously
  const int lane = threadIdx.x & 63;
  const int node = __builtin_amdgcn_readfirstlane(int(xcd_block() * (blockDim.x >> 6) + (threadIdx.x >> 6)));   // xcd_grid launch
  (void)N;
  if (node >= Nt) return;
  const int o = orig[node];
  const int beg = rowptr[o], end = rowptr[o + 1];
  if (end <= beg) return;
  const int b0 = beg >> 6, b1 = (end - 1) >> 6;
  const f4 R = *reinterpret_cast<const f4*>(rot + 4 * o);
  const float2* pd_row = reinterpret_cast<const float2*>(pos) + int64_t(o) * TT;
  for (int t = blockIdx.y; t < H; t += gridDim.y) {                        // gridDim.y waves share a node's snapshots: more gathers in flight
    const int seg = t * Nt + node;
    const int base = segptr[seg], n = segptr[seg + 1] - base;
    if (n == 0) continue;                                                  // (uniform)
    const float2 pd = pd_row[t];
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int r = c0 + lane;
      unsigned long long selw = 0;
      int selb = b0, rr = 0, cum = 0;
      for (int b = b0; b <= b1; ++b) {                                     // the row's ballot words are wave-uniform
        const unsigned long long w = bal[int64_t(b) * H + t] & row_part(b, beg, end);
        const int c = __popcll(w);
        if (r >= cum && r < cum + c) { selw = w; selb = b; rr = r - cum; }
        cum += c;
        if (cum >= c0 + 64) break;                                         // (uniform) every lane of this chunk is served
      }
      if (r < n) {
        const int sdr = csr_src[64 * selb + nth_set_bit(selw, rr)];        // senders are real actors
        const float2 ps = reinterpret_cast<const float2*>(pos)[int64_t(sdr) * TT + t];
        const float dx = ps.x - pd.x, dy = ps.y - pd.y;
        const float x0 = x[(int64_t(sdr) * H + t) * 2], x1 = x[(int64_t(sdr) * H + t) * 2 + 1];
        const f4 g = {x0 * R[0] + x1 * R[2], x0 * R[1] + x1 * R[3], dx * R[0] + dy * R[2], dx * R[1] + dy * R[3]};   // v @ R_i (ENC:584-585)
        const int64_t k = int64_t(base) + r;
        *reinterpret_cast<f4*>(geom + 4 * k) = g;
        aa_dst[k] = seg;
        if (aa_src != nullptr) aa_src[k] = sdr;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- global / lane edges
// global interactor edges: both endpoints valid at the reference step (AGG:41)
__global__ void k_g_flags(int E, int TT, int tref, const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                          const uint8_t* __restrict__ pad, uint8_t* __restrict__ flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > E) return;
  uint8_t keep = 0;
  if (p < E) keep = !pad[int64_t(csr_src[p]) * TT + tref] && !pad[int64_t(csr_dst[p]) * TT + tref];   // csr_dst: written by the row sort
  flags[p] = keep;
}
__global__ void k_g_compact(int E, int TT, int tref, const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                            const float* __restrict__ pos, const float* __restrict__ rot, const float* __restrict__ ang,
                            const uint8_t* __restrict__ flags, const int32_t* __restrict__ cpos, int32_t* __restrict__ g_src,
                            int32_t* __restrict__ g_dst, float* __restrict__ geom) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E || !flags[p]) return;
  const int s = csr_src[p], i = csr_dst[p], q = cpos[p];
  const f4 R = *reinterpret_cast<const f4*>(rot + 4 * i);
  const float dx = pos[(int64_t(s) * TT + tref) * 2] - pos[(int64_t(i) * TT + tref) * 2];
  const float dy = pos[(int64_t(s) * TT + tref) * 2 + 1] - pos[(int64_t(i) * TT + tref) * 2 + 1];
  const float th = ang[s] - ang[i];                                                          // AGG:48-50
  f4 g = {dx * R[0] + dy * R[2], dx * R[1] + dy * R[3], cosf(th), sinf(th)};
  *reinterpret_cast<f4*>(geom + 4 * int64_t(q)) = g;
  g_src[q] = s;
  g_dst[q] = i;
}

// lane feature (ENC:68-71); torch's negative index wraps when a lane is fully padded
__device__ __forceinline__ void lane_feat_body(int l, int L, int P, const float* __restrict__ lp, const float* __restrict__ pad,
                                               float* __restrict__ feat) {
  if (l >= L) return;
  float len = 0.f;
  for (int j = 0; j < P; ++j) len += 1.0f - pad[int64_t(l) * P + j];
  int last = int(len - 1.0f);
  if (last < 0) last += P;
  feat[2 * l] = lp[(int64_t(l) * P + last) * 2] - lp[int64_t(l) * P * 2];
  feat[2 * l + 1] = lp[(int64_t(l) * P + last) * 2 + 1] - lp[int64_t(l) * P * 2 + 1];
}
// The four per-input passes that depend on nothing but the batch -- extended-node table, fake agents' inputs, per-actor validity
// masks, lane features -- as ONE launch: blocks [0, b1) / [b1, b2) / [b2, b3) / [b3, ..) take one pass each (a 71-kernel forward
// spends 5 % of its time in launches of a few microseconds).
struct InputPassArgs {
  int N, A, H, TT, L, P, b1, b2, b3;
  const int64_t *agent_index, *batch, *source;
  const uint8_t *bos, *pad;
  const float *x, *lane_pos, *lane_pad;
  int32_t *orig, *eos, *pick_slot;
  uint8_t* nus;
  float *x_fake, *lane_feat;
  uint32_t* vmask;
  NoiseArg na;
};
__global__ __launch_bounds__(256) void k_input_passes(const InputPassArgs a) {
  const int blk = blockIdx.x;
  if (blk < a.b1) ext_nodes_body(blk * 256 + threadIdx.x, a.N, a.A, a.H, a.agent_index, a.batch, a.source, a.bos, a.orig, a.nus, a.eos, a.pick_slot);
  else if (blk < a.b2) fake_x_body((blk - a.b1) * 256 + threadIdx.x, a.A, a.H, a.x, a.agent_index, a.na, a.x_fake);
  else if (blk < a.b3) valid_mask_body((blk - a.b2) * 256 + threadIdx.x, a.N, a.H, a.TT, a.pad, a.vmask);
  else lane_feat_body((blk - a.b3) * 256 + threadIdx.x, a.L, a.P, a.lane_pos, a.lane_pad, a.lane_feat);
}
__global__ void k_la_flags(int E_al, const int32_t* __restrict__ eid, const float* __restrict__ vec, float radius, uint8_t* __restrict__ flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > E_al) return;
  uint8_t keep = 0;
  if (p < E_al) {                                          // (the actor of every position: written by the row sort)
    const float vx = vec[2 * int64_t(eid[p])], vy = vec[2 * int64_t(eid[p]) + 1];
    keep = sqrtf(norm2_sq(vx, vy)) < radius;                                                  // ENC:198
  }
  flags[p] = keep;
}
__global__ void k_la_compact(int E_al, const int32_t* __restrict__ actor, const int32_t* __restrict__ eid,
                             const int64_t* __restrict__ la_index, const float* __restrict__ vec,
                             const float* __restrict__ lane_feat, const float* __restrict__ rot,
                             const uint8_t* __restrict__ flags, const int32_t* __restrict__ cpos, int32_t* __restrict__ la_dst,
                             int32_t* __restrict__ la_lane, float* __restrict__ geom) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E_al || !flags[p]) return;
  const int i = actor[p], e = eid[p], q = cpos[p];
  const int lane = int(la_index[e]);                                   // row 0 of lane_actor_index
  const f4 R = *reinterpret_cast<const f4*>(rot + 4 * i);
  const float fx = lane_feat[2 * lane], fy = lane_feat[2 * lane + 1], vx = vec[2 * int64_t(e)], vy = vec[2 * int64_t(e) + 1];
  f4 g = {fx * R[0] + fy * R[2], fx * R[1] + fy * R[3], vx * R[0] + vy * R[2], vx * R[1] + vy * R[3]};   // ENC:763-764
  *reinterpret_cast<f4*>(geom + 4 * int64_t(q)) = g;
  la_dst[q] = i;
  if (la_lane != nullptr) la_lane[q] = lane;
}
// the last launch of the graph stage: the segment pointers of the global and lane lists (positions of the row starts among the
// survivors) and the three list lengths
__global__ void k_collect_counts(int64_t n_aa, int E, int E_al, const int32_t* __restrict__ aa_segptr, const int32_t* __restrict__ cpos_g,
                                 const int32_t* __restrict__ cpos_la, float radius, int32_t* __restrict__ counts, int N,
                                 const int32_t* __restrict__ rowptr, const int32_t* __restrict__ la_rowptr, int32_t* __restrict__ g_segptr,
                                 int32_t* __restrict__ la_segptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= N) {
    g_segptr[i] = cpos_g[rowptr[i]];
    la_segptr[i] = cpos_la[la_rowptr[i]];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[0] = 0;
    reinterpret_cast<float*>(counts)[4] = radius;
    counts[1] = aa_segptr[n_aa];
    counts[2] = cpos_g[E];
    counts[3] = cpos_la[E_al];
  }
}

// ---- exclusive prefix sums of the graph stage: one launch each, no library.  (Rounds 1-3 called hipcub::DeviceScan: two launches
// per scan -- init_lookback_scan_state + the scan -- and rocPRIM on the hot path for what are 8 K .. 2 M element scans.)
// "Chained scan with decoupled look-back": a workgroup takes a ticket (so that a workgroup only ever waits for workgroups that have
// already started), scans its chunk in registers, publishes its sum, then walks back over the published words of its predecessors
// -- a wave looks at 64 of them at a time -- until it meets one that carries a complete prefix; it then publishes its own.  One
// 64-bit word per workgroup holds (state << 32 | value), read and written with agent-scope atomics; the words and the ticket are
// zeroed by the memset that already clears the degree counters (PrepWs::zeroed).  Values are counts < 2^31.
constexpr int SCAN_MAX_BLOCKS = 512;
constexpr int SCAN_WORDS = SCAN_MAX_BLOCKS + 1;            // per scan instance: the ticket + one word per workgroup
constexpr int SCAN_INSTANCES = 5;                          // agent-agent segments, global flags, lane flags, the two CSR row pointers
template <typename IN, int PER>                            // PER elements per thread, PER x 1024 per workgroup
__global__ __launch_bounds__(1024) void k_scan_chained(const IN* __restrict__ in, int32_t* __restrict__ out, int n,
                                                       unsigned long long* __restrict__ st) {
  __shared__ int32_t wsum[16];
  __shared__ int s_bid;
  __shared__ int32_t s_prefix;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) s_bid = int(__hip_atomic_fetch_add(&st[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
  const int b = s_bid;
  const int64_t first = (int64_t(b) * 1024 + t) * PER;
  int32_t v[PER];
  int32_t s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = first + i < n ? int32_t(in[first + i]) : 0;
    s += v[i];
  }
  int32_t inc = s;                                          // inclusive scan of the threads' sums inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int32_t wbase = 0;
  for (int w = 0; w < wv; ++w) wbase += wsum[w];
  if (wv == 0) {
    int32_t total = lane < 16 ? wsum[lane] : 0;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) total += __shfl_xor(total, d);
    total = __shfl(total, 0);
    unsigned long long* words = st + 1;
    if (lane == 0)
      __hip_atomic_store(&words[b], ((b == 0 ? 2ull : 1ull) << 32) | (unsigned long long)(unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int32_t prefix = 0;
    if (b > 0) {
      int j = b - 1;
      for (long spins = 0;; ++spins) {
        if (spins > (1L << 28)) __builtin_trap();          // a predecessor never published: fail loudly, never hang
        const int k = j - lane;
        const unsigned long long w = k >= 0 ? __hip_atomic_load(&words[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 32);
        const unsigned flag = unsigned(w >> 32);
        const unsigned long long done = __ballot(flag == 2u), none = __ballot(flag == 0u);
        const int upto = done ? __builtin_ctzll(done) : 63;    // the nearest predecessor with a complete prefix ends the walk
        const unsigned long long need = upto == 63 ? ~0ull : ((1ull << (upto + 1)) - 1);
        if (none & need) continue;                         // one of the words we need is not there yet: look again
        int32_t part = lane <= upto ? int32_t(unsigned(w)) : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) part += __shfl_xor(part, d);
        prefix += part;
        if (done) break;
        j -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&words[b], (2ull << 32) | (unsigned long long)(unsigned)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  int32_t run = s_prefix + wbase + inc - s;                 // exclusive prefix of this thread's first element
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (first + i < n) out[first + i] = run;
    run += v[i];
  }
}
// out[i] = sum of in[0 .. i), i < n; `st`: this scan's SCAN_WORDS zeroed words; in == out is allowed
template <typename IN>
static int scan_exclusive(const IN* in, int32_t* out, int64_t n, unsigned long long* st, hipStream_t stream) {
  TS_REQUIRE(n <= int64_t(SCAN_MAX_BLOCKS) * 65536, "graph stage: more than 33 M elements in a prefix sum");
  const int64_t per = (n + int64_t(SCAN_MAX_BLOCKS) * 1024 - 1) / (int64_t(SCAN_MAX_BLOCKS) * 1024);      // smallest class that needs <= 512 workgroups
  if (per <= 8) k_scan_chained<IN, 8><<<cdiv(n, 8 * 1024), 1024, 0, stream>>>(in, out, int(n), st);
  else if (per <= 16) k_scan_chained<IN, 16><<<cdiv(n, 16 * 1024), 1024, 0, stream>>>(in, out, int(n), st);
  else if (per <= 32) k_scan_chained<IN, 32><<<cdiv(n, 32 * 1024), 1024, 0, stream>>>(in, out, int(n), st);
  else k_scan_chained<IN, 64><<<cdiv(n, 64 * 1024), 1024, 0, stream>>>(in, out, int(n), st);
  TS_LAUNCH_CHECK("k_scan_chained");
  return TRAJSDE_OK;
}

// phase-1 workspace layout, derived from the batch sizes alone (so both phases agree on it)
struct PrepWs {
  int32_t *deg, *rowptr, *csr_src, *csr_dst, *orig, *eos, *pick_slot, *counts;
  int32_t *la_deg, *la_rowptr, *la_eid, *la_actor;
  int64_t* la_pack;
  int32_t *aa_segptr, *g_segptr, *la_segptr, *cpos_g, *cpos_la;
  uint8_t *nus, *flags_g, *flags_la;
  uint32_t* vmask;               // per actor: bit t = valid at history step t
  unsigned long long* bal;       // [ceil(E / 64)][H]: survivors of 64 consecutive CSR positions at step t (k_aa_ballots)
  float *x_fake, *lane_feat;
  unsigned long long* scan_st;   // SCAN_INSTANCES x SCAN_WORDS zeroed words (k_scan_chained), directly behind the degree counters
  int64_t zeroed_bytes, n_aa;
  int64_t total;
  bool ok;
  PrepWs(const trajsde_batch* b, void* ws, int64_t ws_bytes) {
    Carver c(ws, ws_bytes);
    const int64_t N = b->N, A = b->A, E = b->E, Nt = N + A, H = b->H, Ea = b->E_al;
    n_aa = H * Nt;
    // both degree arrays and the scans' state words: ONE block, one memset (graph_prepare)
    const int64_t deg_ints = (2 * (N + 1) + 1) / 2 * 2;
    deg = c.take<int32_t>(deg_ints + 2 * int64_t(SCAN_INSTANCES) * SCAN_WORDS); la_deg = ptr_add(deg, N + 1);
    scan_st = reinterpret_cast<unsigned long long*>(ptr_add(deg, deg_ints));
    zeroed_bytes = (deg_ints + 2 * int64_t(SCAN_INSTANCES) * SCAN_WORDS) * int64_t(sizeof(int32_t));
    rowptr = c.take<int32_t>(N + 1);
    csr_src = c.take<int32_t>(E + 1); csr_dst = c.take<int32_t>(E + 1);
    orig = c.take<int32_t>(Nt); eos = c.take<int32_t>(Nt); pick_slot = c.take<int32_t>(Nt); counts = c.take<int32_t>(8);
    la_rowptr = c.take<int32_t>(N + 1);
    la_eid = c.take<int32_t>(Ea + 1); la_actor = c.take<int32_t>(Ea + 1); la_pack = c.take<int64_t>(Ea + 1);
    aa_segptr = c.take<int32_t>(n_aa + 1); g_segptr = c.take<int32_t>(N + 1); la_segptr = c.take<int32_t>(N + 1);
    cpos_g = c.take<int32_t>(E + 1); cpos_la = c.take<int32_t>(Ea + 1);
    nus = c.take<uint8_t>(Nt); flags_g = c.take<uint8_t>(E + 1); flags_la = c.take<uint8_t>(Ea + 1);
    x_fake = c.take<float>(A * H * 2 + 4); lane_feat = c.take<float>(int64_t(b->L) * 2 + 4);
    vmask = c.take<uint32_t>(N + 1);
    bal = c.take<unsigned long long>(((E + 63) / 64 + 1) * H);
    total = c.off + 256;
    ok = c.ok;
  }
};

struct EdgeWs {
  float *aa_geom, *g_geom, *la_geom;
  int32_t *aa_dst, *aa_src, *g_src, *g_dst, *la_dst, *la_lane;
  int64_t total;
  bool ok;
  EdgeWs(const trajsde_batch* b, const trajsde_graph* g, void* ws, int64_t ws_bytes) {
    Carver c(ws, ws_bytes);
    (void)b;
    const int64_t Eaa = g->E_aa, Eg = g->E_g, Ela = g->E_la;            // (int64: E + 1 overflows int32 at the bound the ABI admits)
    aa_geom = c.take<float>(4 * Eaa + 4); aa_dst = c.take<int32_t>(Eaa + 1); aa_src = c.take<int32_t>(Eaa + 1);
    g_geom = c.take<float>(4 * Eg + 4); g_src = c.take<int32_t>(Eg + 1); g_dst = c.take<int32_t>(Eg + 1);
    la_geom = c.take<float>(4 * Ela + 4); la_dst = c.take<int32_t>(Ela + 1); la_lane = c.take<int32_t>(Ela + 1);
    total = c.off + 256;
    ok = c.ok;
  }
};

static int check_batch(const trajsde_batch* b) {
  TS_REQUIRE(b != nullptr, "batch: null");
  TS_REQUIRE(b->N > 0 && b->A >= 0 && b->H > 0 && b->H <= 32 && b->TT >= b->H, "batch: bad sizes (H <= 32)");
  TS_REQUIRE(b->E >= 0 && b->E_al >= 0 && b->L >= 0, "batch: negative sizes");
  TS_REQUIRE(int64_t(b->H) * b->E < (int64_t(1) << 31) - 2, "batch: too many (t, edge) candidates for int32 positions");
  TS_REQUIRE(b->x && b->positions && b->padding_mask && b->bos_mask && b->rotate_angles && b->batch && b->source,
             "batch: null actor tensor");
  TS_REQUIRE(b->A == 0 || b->agent_index, "batch: null agent_index");   // A == 0: no fake agents (forward_ood, ENC:204-370)
  TS_REQUIRE(b->E == 0 || b->edge_index, "batch: null edge_index");
  TS_REQUIRE(b->E_al == 0 || (b->lane_actor_index && b->lane_actor_vectors && b->lane_positions && b->lane_paddings),
             "batch: null lane tensor");
  return TRAJSDE_OK;
}

// CSR by target: degree histogram -> exclusive scan -> scatter -> canonical row order
// exclusive prefix sum of up to 32 768 int32 by ONE workgroup (the row pointers of a batch: N + 1 counts) -- one launch of a few
// microseconds where the device-wide scan is two
__global__ __launch_bounds__(1024) void k_scan_small(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n) {
  __shared__ int32_t wsum[16];
  const int per = (n + 1023) / 1024, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int beg = t * per, end = beg + per < n ? beg + per : n;
  int32_t s = 0;
  for (int i = beg; i < end; ++i) s += in[i];
  int32_t inc = s;                                          // inclusive scan of the threads' sums inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int32_t base = 0;
  for (int w = 0; w < wv; ++w) base += wsum[w];
  int32_t run = base + inc - s;                             // exclusive prefix of this thread's first element
  for (int i = beg; i < end; ++i) {
    const int32_t v = in[i];
    out[i] = run;
    run += v;
  }
}
// (deg arrives zeroed)
// dst_out (or null): the target of every CSR position (the row it lies in), for the passes that walk positions, not rows
static int build_csr(const int64_t* ei, int E, int N, int32_t* deg, int32_t* rowptr, int32_t* out, int64_t* lane_pack,
                     unsigned long long* scan_words, hipStream_t st, int32_t* dst_out = nullptr) {
  if (E > 0) k_degree<<<cdiv(E, 256), 256, 0, st>>>(ei, E, deg);
  if (N + 1 <= 32768) {
    k_scan_small<<<1, 1024, 0, st>>>(deg, rowptr, N + 1);
  } else if (int rc = scan_exclusive(deg, rowptr, int64_t(N) + 1, scan_words, st)) {
    return rc;
  }
  if (E > 0 && lane_pack == nullptr) {
    k_scatter<<<cdiv(E, 256), 256, 0, st>>>(ei, E, rowptr, deg, out, 0);
    k_row_sort<int32_t><<<cdiv(N, 4) < 8192 ? cdiv(N, 4) : 8192, 256, 0, st>>>(rowptr, N, out, nullptr, dst_out);
  } else if (E > 0) {
    k_scatter_lane<<<cdiv(E, 256), 256, 0, st>>>(ei, E, rowptr, deg, lane_pack);
    k_row_sort<int64_t><<<cdiv(N, 4) < 8192 ? cdiv(N, 4) : 8192, 256, 0, st>>>(rowptr, N, lane_pack, out, dst_out);   // also writes the edge ids
  }
  TS_LAUNCH_CHECK("build_csr");
  return TRAJSDE_OK;
}

}  // namespace tsde

using namespace tsde;

static std::atomic<int> g_export_senders{0};

extern "C" {

float trajsde_radius2_threshold(float radius) { return radius2_threshold(radius); }

int trajsde_export_senders(int on) {
  const int prev = g_export_senders.exchange(on ? 1 : 0);
  return prev;
}

int trajsde_rotate(const float* rotate_angles, int32_t N, const float* y, int32_t F, float* rotate_mat, float* y_rot, void* stream) {
  TS_REQUIRE(rotate_angles && rotate_mat && N > 0, "rotate: bad argument");
  TS_REQUIRE(y == nullptr || y_rot != nullptr, "rotate: y given without y_rot");
  k_rotate<<<cdiv(N, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(rotate_angles, N, y, F, rotate_mat, y_rot);
  TS_LAUNCH_CHECK("k_rotate");
  return TRAJSDE_OK;
}

int64_t trajsde_graph_ws_bytes(const trajsde_batch* b) {
  if (check_batch(b) != TRAJSDE_OK) return -1;
  PrepWs w(b, nullptr, 0);
  return w.total;
}

static int graph_prepare(const trajsde_batch* b, const float* rot, float radius, const trajsde_noise* fake_noise, void* ws,
                         int64_t ws_bytes, trajsde_graph* out, void* stream_, bool sync) {
  if (int rc = check_batch(b)) return rc;
  TS_REQUIRE(rot && ws && out, "graph_prepare: null pointer");
  PrepWs w(b, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "graph_prepare: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int N = b->N, A = b->A, E = b->E, H = b->H, TT = b->TT, Ea = b->E_al, Nt = N + A;
  NoiseArg na{0, nullptr, nullptr};
  if (fake_noise) { na.seed = fake_noise->seed; na.z = fake_noise->z; na.row_ids = fake_noise->row_ids; na.seed_dev = fake_noise->seed_dev; }

  TS_HIP(hipMemsetAsync(w.deg, 0, size_t(w.zeroed_bytes), st));       // actor and lane degree counters, the scans' state words
  {
    ProfScope ps("build_csr[actors]", st);
    if (int rc = build_csr(b->edge_index, E, N, w.deg, w.rowptr, w.csr_src, nullptr, w.scan_st + 3 * SCAN_WORDS, st, w.csr_dst)) return rc;
  }
  {
    InputPassArgs ia;
    ia.N = N; ia.A = A; ia.H = H; ia.TT = TT; ia.L = b->L; ia.P = b->lane_pts;
    ia.b1 = cdiv(Nt, 256);
    ia.b2 = ia.b1 + (A > 0 ? cdiv(A * ((2 * H + 3) / 4), 256) : 0);
    ia.b3 = ia.b2 + (E > 0 ? cdiv(N, 256) : 0);
    const int blocks = ia.b3 + (b->L > 0 ? cdiv(b->L, 256) : 0);
    ia.agent_index = b->agent_index; ia.batch = b->batch; ia.source = b->source; ia.bos = b->bos_mask; ia.pad = b->padding_mask;
    ia.x = b->x; ia.lane_pos = b->lane_positions; ia.lane_pad = b->lane_paddings;
    ia.orig = w.orig; ia.eos = w.eos; ia.pick_slot = w.pick_slot; ia.nus = w.nus; ia.x_fake = w.x_fake; ia.lane_feat = w.lane_feat;
    ia.vmask = w.vmask; ia.na = na;
    k_input_passes<<<blocks, 256, 0, st>>>(ia);
  }
  if (A > 0) k_agent_slots<<<cdiv(A, 256), 256, 0, st>>>(A, b->agent_index, w.pick_slot);    // after the table: overrides its -1 entries
  // global interactor edges (also names the target of every CSR position: csr_dst)
  k_g_flags<<<cdiv(E + 1, 256), 256, 0, st>>>(E, TT, H - 1, w.csr_src, w.csr_dst, b->padding_mask, w.flags_g);
  // 21 snapshots: survivor ballots -> segment lengths -> prefix sum -> segment pointers
  if (E > 0) {
    const int lds_b = 4 * 64 * (H | 1) * int(sizeof(float2));              // (67 KB at H = 32: TS_LAUNCH raises the dynamic-LDS limit)
    TS_LAUNCH(k_aa_ballots, xcd_grid(cdiv(cdiv(E, 64), 4)), 256, lds_b, st, E, H, TT, w.csr_src, w.csr_dst, w.vmask, b->positions, radius2_threshold(radius),
              w.bal);
  }
  { ProfScope ps("k_aa_count", st);
  k_aa_count<<<cdiv(int64_t(H) * Nt, 256), 256, 0, st>>>(Nt, H, w.rowptr, w.orig, w.bal, w.aa_segptr); }
  if (int rc = scan_exclusive(w.aa_segptr, w.aa_segptr, w.n_aa + 1, w.scan_st + 0 * SCAN_WORDS, st)) return rc;
  if (int rc = scan_exclusive(w.flags_g, w.cpos_g, int64_t(E) + 1, w.scan_st + 1 * SCAN_WORDS, st)) return rc;
  // lane-actor edges grouped by actor
  {
    ProfScope ps("build_csr[lanes]", st);
    if (int rc = build_csr(b->lane_actor_index, Ea, N, w.la_deg, w.la_rowptr, w.la_eid, w.la_pack, w.scan_st + 4 * SCAN_WORDS, st, w.la_actor)) return rc;
  }
  k_la_flags<<<cdiv(Ea + 1, 256), 256, 0, st>>>(Ea, w.la_eid, b->lane_actor_vectors, radius, w.flags_la);
  if (int rc = scan_exclusive(w.flags_la, w.cpos_la, int64_t(Ea) + 1, w.scan_st + 2 * SCAN_WORDS, st)) return rc;
  k_collect_counts<<<cdiv(N + 1, 256), 256, 0, st>>>(w.n_aa, E, Ea, w.aa_segptr, w.cpos_g, w.cpos_la, radius, w.counts, N, w.rowptr, w.la_rowptr,
                                                     w.g_segptr, w.la_segptr);
  TS_LAUNCH_CHECK("graph_prepare kernels");
  int32_t h[4] = {0, 0, 0, 0};
  if (sync) {
    TS_HIP(hipMemcpyAsync(h, w.counts, sizeof(h), hipMemcpyDeviceToHost, st));
    TS_HIP(hipStreamSynchronize(st));
  } else {
    // upper bounds: every (t, in-edge) candidate of every extended node survives.  The fake rows alias their agents' CSR
    // rows, whose in-degrees sum to at most E (repeated input edges count, so A * (N - 1) is NOT a bound)
    const int64_t aa_max = int64_t(H) * 2 * int64_t(E);
    (void)A;
    TS_REQUIRE(aa_max < (int64_t(1) << 31) - 2, "graph_prepare_async: too many (t, edge) candidates for int32 positions");
    h[1] = int32_t(aa_max); h[2] = E; h[3] = Ea;
  }
  std::memset(out, 0, sizeof(*out));
  out->counts = w.counts; out->exact = sync ? 1 : 0;
  out->Nt = Nt; out->E_ext = E; out->E_aa = h[1]; out->E_g = h[2]; out->E_la = h[3];
  out->orig = w.orig; out->nus_mask = w.nus; out->eos_idx = w.eos; out->pick_slot = w.pick_slot; out->x_fake = w.x_fake;
  out->aa_segptr = w.aa_segptr; out->g_segptr = w.g_segptr; out->la_segptr = w.la_segptr;
  return TRAJSDE_OK;
}

int trajsde_graph_prepare(const trajsde_batch* b, const float* rot, float radius, const trajsde_noise* fake_noise, void* ws,
                          int64_t ws_bytes, trajsde_graph* out, void* stream_) {
  return graph_prepare(b, rot, radius, fake_noise, ws, ws_bytes, out, stream_, true);
}

int trajsde_graph_prepare_async(const trajsde_batch* b, const float* rot, float radius, const trajsde_noise* fake_noise, void* ws,
                                int64_t ws_bytes, trajsde_graph* out, void* stream_) {
  return graph_prepare(b, rot, radius, fake_noise, ws, ws_bytes, out, stream_, false);
}

int64_t trajsde_graph_edges_ws_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EdgeWs e(b, g, nullptr, 0);
  return e.total;
}

int trajsde_graph_compact(const trajsde_batch* b, const float* rot, void* ws, int64_t ws_bytes, void* edges_ws,
                          int64_t edges_ws_bytes, trajsde_graph* out, void* stream_) {
  if (int rc = check_batch(b)) return rc;
  TS_REQUIRE(rot && ws && edges_ws && out, "graph_compact: null pointer");
  PrepWs w(b, ws, ws_bytes);
  EdgeWs e(b, out, edges_ws, edges_ws_bytes);
  if (!w.ok || !e.ok) return fail(TRAJSDE_ERR_WORKSPACE, "graph_compact: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int N = b->N, A = b->A, E = b->E, H = b->H, TT = b->TT, Ea = b->E_al, Nt = N + A;
  const bool want_src = g_export_senders.load() != 0;                      // sender ids are for checking the index work only
  { ProfScope ps("k_aa_fill", st);
  static const int fill_parts = []() { const char* e = getenv("TRAJSDE_FILL_PARTS"); const int v = e ? atoi(e) : 7; return v >= 1 && v <= 32 ? v : 7; }();   // 94 us at 1, 80 at 3, 79 at 7, 84 at 21 (32 x 256 agents)
  k_aa_fill<<<dim3(xcd_grid(cdiv(Nt, 4)), fill_parts), 256, 0, st>>>(N, Nt, H, TT, w.rowptr, w.csr_src, w.orig, w.bal, w.aa_segptr, b->positions, b->x, rot, e.aa_dst,
                                   want_src ? e.aa_src : nullptr, e.aa_geom); }
  if (E > 0)
    k_g_compact<<<cdiv(E, 256), 256, 0, st>>>(E, TT, H - 1, w.csr_src, w.csr_dst, b->positions, rot, b->rotate_angles, w.flags_g,
                                              w.cpos_g, e.g_src, e.g_dst, e.g_geom);
  if (Ea > 0)
    k_la_compact<<<cdiv(Ea, 256), 256, 0, st>>>(Ea, w.la_actor, w.la_eid, b->lane_actor_index, b->lane_actor_vectors, w.lane_feat,
                                                rot, w.flags_la, w.cpos_la, e.la_dst, want_src ? e.la_lane : nullptr, e.la_geom);
  TS_LAUNCH_CHECK("graph_compact kernels");
  out->aa_geom = e.aa_geom; out->aa_dst = e.aa_dst;
  out->aa_src = want_src ? e.aa_src : nullptr; out->la_lane = want_src ? e.la_lane : nullptr;
  out->g_geom = e.g_geom; out->g_src = e.g_src; out->g_dst = e.g_dst;
  out->la_geom = e.la_geom; out->la_dst = e.la_dst;
  return TRAJSDE_OK;
}

}  // extern "C"
