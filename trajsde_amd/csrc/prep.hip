// prep.hip -- everything the reference does with index tensors before the attention blocks, on device:
//   MODEL:75-85   rotate_mat, y @ rotate_mat
//   ENC:68-71     lane feature = last valid lane point - first lane point
//   ENC:73-74,103 nus_mask;  ENC:88-103 fake copies of the target agents (x + 2*randn, duplicated in-edges)
//   ENC:107-118   21 x subgraph(valid at t) + DistanceDropEdge(radius) (UTIL:83-92)
//   AGG:41-51     subgraph(valid at t=H-1), relative pose;   ENC:198 radius drop on lane-actor edges
// Instead of materialising 21 boolean-masked edge lists, edges are sorted once by target (hipcub radix
// sort -> CSR), every (t, edge) candidate gets a 1-byte flag, one prefix sum gives the compacted position
// of every survivor, and the segment pointer of snapshot node (t, i) is read off the same prefix sum.
// The compacted lists carry the pre-rotated 2-d geometry the edge kernels consume (16 B per edge).
#include <hipcub/hipcub.hpp>

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

__global__ void k_split_edges(const int64_t* __restrict__ ei, int E, int row_key, int32_t* __restrict__ keys,
                              int32_t* __restrict__ vals, int vals_are_ids) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  keys[e] = int32_t(ei[int64_t(row_key) * E + e]);
  vals[e] = vals_are_ids ? e : int32_t(ei[int64_t(1 - row_key) * E + e]);
}

// rowptr[i] = first position whose key >= i   (i in [0, n])
__global__ void k_rowptr(const int32_t* __restrict__ keys, int E, int n, int32_t* __restrict__ rowptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int lo = 0, hi = E;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < i) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = lo;
}

// per extended node: original actor, source mask, recurrence iteration to keep; slots of the agent rows
__global__ void k_ext_nodes(int N, int A, int H, const int64_t* __restrict__ agent_index, const int64_t* __restrict__ batch,
                            const int64_t* __restrict__ source, const uint8_t* __restrict__ bos,
                            int32_t* __restrict__ orig, uint8_t* __restrict__ nus, int32_t* __restrict__ eos,
                            int32_t* __restrict__ pick_slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
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
__global__ void k_fake_x(int A, int H, const float* __restrict__ x, const int64_t* __restrict__ agent_index, NoiseArg na,
                         float* __restrict__ x_fake) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (k, quad of 4 columns)
  const int quads = (2 * H + 3) / 4;
  if (idx >= A * quads) return;
  const int k = idx / quads, q = idx % quads;
  f4 z;
  if (na.z != nullptr) {
    for (int c = 0; c < 4; ++c) z[c] = (4 * q + c < 2 * H) ? na.z[int64_t(k) * 2 * H + 4 * q + c] : 0.f;
  } else {
    z = philox_normal4(na.seed, STREAM_FAKE_AGENT, 0u, na.row_ids ? uint32_t(na.row_ids[k]) : uint32_t(k), uint32_t(q));
  }
  const int64_t a = agent_index[k];
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * q + c;
    if (col < 2 * H) x_fake[int64_t(k) * 2 * H + col] = x[a * 2 * H + col] + 2.0f * z[c];
  }
}

// ext_rowptr: rows 0..N as rowptr, rows of the fake agents appended (copies of their agents' in-edges)
__global__ void k_ext_rowptr(int N, int A, int E, const int32_t* __restrict__ rowptr, const int64_t* __restrict__ agent_index,
                             int32_t* __restrict__ ext_rowptr, int32_t* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= N) ext_rowptr[i] = rowptr[i];
  if (i == 0) {
    int acc = E;
    for (int k = 0; k < A; ++k) {
      const int a = int(agent_index[k]);
      acc += rowptr[a + 1] - rowptr[a];
      ext_rowptr[N + k + 1] = acc;
    }
    counts[0] = acc;   // E_ext
  }
}

struct EdgeRef { int i, o, src; };
// candidate p of the extended CSR -> (extended dst node i, its original actor o, source actor)
__device__ __forceinline__ EdgeRef resolve_ext_edge(int p, int N, int A, int E, const int32_t* csr_src, const int32_t* csr_dst,
                                                    const int32_t* rowptr, const int32_t* ext_rowptr, const int32_t* orig) {
  EdgeRef r;
  if (p < E) {
    r.i = csr_dst[p]; r.o = r.i; r.src = csr_src[p];
  } else {
    int lo = N, hi = N + A - 1;                       // last fake row whose start <= p
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (ext_rowptr[mid] <= p) lo = mid; else hi = mid - 1;
    }
    r.i = lo; r.o = orig[lo];
    r.src = csr_src[rowptr[r.o] + (p - ext_rowptr[lo])];
  }
  return r;
}

// One thread per extended edge, looping over the H snapshots: the edge is resolved once and the H positions /
// padding bytes of both endpoints are contiguous in memory.  flags is zero-filled beforehand (memset).
__global__ void k_aa_flags(int N, int A, int E, int H, int TT, const int32_t* __restrict__ counts,
                           const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ext_rowptr,
                           const int32_t* __restrict__ orig, const uint8_t* __restrict__ pad,
                           const float* __restrict__ pos, float radius, uint8_t* __restrict__ flags) {
  const int E_ext = counts[0];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E_ext; p += gridDim.x * blockDim.x) {
    const EdgeRef r = resolve_ext_edge(p, N, A, E, csr_src, csr_dst, rowptr, ext_rowptr, orig);
    const uint8_t* ps = pad + int64_t(r.src) * TT;
    const uint8_t* pd = pad + int64_t(r.o) * TT;
    const float* xs = pos + int64_t(r.src) * TT * 2;
    const float* xd = pos + int64_t(r.o) * TT * 2;
    for (int t = 0; t < H; ++t) {
      uint8_t keep = 0;
      if (!ps[t] && !pd[t]) {                                                        // subgraph, ENC:108
        const float dx = xs[2 * t] - xd[2 * t], dy = xs[2 * t + 1] - xd[2 * t + 1];
        keep = sqrtf(dx * dx + dy * dy) < radius;                                     // UTIL:88
      }
      flags[int64_t(t) * E_ext + p] = keep;
    }
  }
}

__global__ void k_aa_compact(int N, int A, int E, int H, int TT, const int32_t* __restrict__ counts,
                             const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ext_rowptr,
                             const int32_t* __restrict__ orig, const float* __restrict__ x, const float* __restrict__ pos,
                             const float* __restrict__ rot, const uint8_t* __restrict__ flags,
                             const int32_t* __restrict__ cpos, int32_t* __restrict__ aa_dst, float* __restrict__ geom) {
  const int E_ext = counts[0];
  const int Nt = N + A;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E_ext; p += gridDim.x * blockDim.x) {
    const EdgeRef r = resolve_ext_edge(p, N, A, E, csr_src, csr_dst, rowptr, ext_rowptr, orig);
    const f4 R = *reinterpret_cast<const f4*>(rot + 4 * r.o);          // [[R0,R1],[R2,R3]]
    const float* xs = x + int64_t(r.src) * H * 2;                      // senders are real actors
    const float* ps = pos + int64_t(r.src) * TT * 2;
    const float* pd = pos + int64_t(r.o) * TT * 2;
    for (int t = 0; t < H; ++t) {
      const int64_t f = int64_t(t) * E_ext + p;
      if (!flags[f]) continue;
      const int q = cpos[f];
      const float x0 = xs[2 * t], x1 = xs[2 * t + 1];
      const float dx = ps[2 * t] - pd[2 * t], dy = ps[2 * t + 1] - pd[2 * t + 1];
      f4 g = {x0 * R[0] + x1 * R[2], x0 * R[1] + x1 * R[3], dx * R[0] + dy * R[2], dx * R[1] + dy * R[3]};   // v @ R_i (ENC:584-585)
      *reinterpret_cast<f4*>(geom + 4 * int64_t(q)) = g;
      aa_dst[q] = t * Nt + r.i;
    }
  }
}

__global__ void k_aa_segptr(int Nt, int H, const int32_t* __restrict__ counts, const int32_t* __restrict__ ext_rowptr,
                            const int32_t* __restrict__ cpos, int32_t* __restrict__ segptr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > H * Nt) return;
  const int E_ext = counts[0];
  const int t = r / Nt, i = r - t * Nt;
  segptr[r] = (r == H * Nt) ? cpos[int64_t(H) * E_ext] : cpos[int64_t(t) * E_ext + ext_rowptr[i]];
}

// global interactor edges: both endpoints valid at the reference step (AGG:41)
__global__ void k_g_flags(int E, int TT, int tref, const int32_t* __restrict__ csr_src, const int32_t* __restrict__ csr_dst,
                          const uint8_t* __restrict__ pad, uint8_t* __restrict__ flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > E) return;
  flags[p] = (p < E) && !pad[int64_t(csr_src[p]) * TT + tref] && !pad[int64_t(csr_dst[p]) * TT + tref];
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
__global__ void k_segptr_from_rowptr(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cpos,
                                     int32_t* __restrict__ segptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) segptr[i] = cpos[rowptr[i]];
}

// lane feature (ENC:68-71); torch's negative index wraps when a lane is fully padded
__global__ void k_lane_feat(int L, int P, const float* __restrict__ lp, const float* __restrict__ pad, float* __restrict__ feat) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  float len = 0.f;
  for (int j = 0; j < P; ++j) len += 1.0f - pad[int64_t(l) * P + j];
  int last = int(len - 1.0f);
  if (last < 0) last += P;
  feat[2 * l] = lp[(int64_t(l) * P + last) * 2] - lp[int64_t(l) * P * 2];
  feat[2 * l + 1] = lp[(int64_t(l) * P + last) * 2 + 1] - lp[int64_t(l) * P * 2 + 1];
}
__global__ void k_la_flags(int E_al, const int32_t* __restrict__ eid, const float* __restrict__ vec, float radius,
                           uint8_t* __restrict__ flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > E_al) return;
  uint8_t keep = 0;
  if (p < E_al) {
    const float vx = vec[2 * int64_t(eid[p])], vy = vec[2 * int64_t(eid[p]) + 1];
    keep = sqrtf(vx * vx + vy * vy) < radius;                                                 // ENC:198
  }
  flags[p] = keep;
}
__global__ void k_la_compact(int E_al, const int32_t* __restrict__ actor, const int32_t* __restrict__ eid,
                             const int64_t* __restrict__ la_index, const float* __restrict__ vec,
                             const float* __restrict__ lane_feat, const float* __restrict__ rot,
                             const uint8_t* __restrict__ flags, const int32_t* __restrict__ cpos, int32_t* __restrict__ la_dst,
                             float* __restrict__ geom) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E_al || !flags[p]) return;
  const int i = actor[p], e = eid[p], q = cpos[p];
  const int lane = int(la_index[e]);                                   // row 0 of lane_actor_index
  const f4 R = *reinterpret_cast<const f4*>(rot + 4 * i);
  const float fx = lane_feat[2 * lane], fy = lane_feat[2 * lane + 1], vx = vec[2 * int64_t(e)], vy = vec[2 * int64_t(e) + 1];
  f4 g = {fx * R[0] + fy * R[2], fx * R[1] + fy * R[3], vx * R[0] + vy * R[2], vx * R[1] + vy * R[3]};   // ENC:763-764
  *reinterpret_cast<f4*>(geom + 4 * int64_t(q)) = g;
  la_dst[q] = i;
}
__global__ void k_collect_counts(int H, int E, int E_al, const int32_t* __restrict__ cpos_aa, const int32_t* __restrict__ cpos_g,
                                 const int32_t* __restrict__ cpos_la, int32_t* __restrict__ counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    counts[1] = cpos_aa[int64_t(H) * counts[0]];
    counts[2] = cpos_g[E];
    counts[3] = cpos_la[E_al];
  }
}

// flags are bytes; scan them as int32 (an accumulator of the input type would wrap at 256)
struct ByteToInt {
  __host__ __device__ __forceinline__ int32_t operator()(const uint8_t& v) const { return int32_t(v); }
};
using FlagIter = hipcub::TransformInputIterator<int32_t, ByteToInt, const uint8_t*>;
static hipError_t scan_flags(void* tmp, size_t& tmp_bytes, const uint8_t* flags, int32_t* out, int n, hipStream_t st) {
  return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, FlagIter(flags, ByteToInt()), out, n, st);
}

static int bits_for(int n) {
  int b = 1;
  while ((int64_t(1) << b) <= n) ++b;
  return b;
}

// phase-1 workspace layout, derived from the batch sizes alone (so both phases agree on it)
struct PrepWs {
  int32_t *k_in, *v_in, *csr_dst, *csr_src, *rowptr, *ext_rowptr, *orig, *eos, *pick_slot, *counts;
  int32_t *la_k_in, *la_v_in, *la_actor, *la_eid, *la_rowptr;
  uint8_t *nus, *flags_aa, *flags_g, *flags_la;
  int32_t *cpos_aa, *cpos_g, *cpos_la;
  float *x_fake, *lane_feat;
  void* cub_tmp;
  int64_t cub_bytes, f_ub;
  int64_t total;
  bool ok;
  PrepWs(const trajsde_batch* b, void* ws, int64_t ws_bytes) {
    Carver c(ws, ws_bytes);
    const int64_t N = b->N, A = b->A, E = b->E, Nt = N + A, H = b->H, Ea = b->E_al;
    f_ub = H * 2 * E;                       // E_ext <= 2E
    k_in = c.take<int32_t>(E); v_in = c.take<int32_t>(E); csr_dst = c.take<int32_t>(E); csr_src = c.take<int32_t>(E);
    rowptr = c.take<int32_t>(N + 1); ext_rowptr = c.take<int32_t>(Nt + 1);
    orig = c.take<int32_t>(Nt); eos = c.take<int32_t>(Nt); pick_slot = c.take<int32_t>(Nt); counts = c.take<int32_t>(8);
    la_k_in = c.take<int32_t>(Ea); la_v_in = c.take<int32_t>(Ea); la_actor = c.take<int32_t>(Ea); la_eid = c.take<int32_t>(Ea);
    la_rowptr = c.take<int32_t>(N + 1);
    nus = c.take<uint8_t>(Nt); flags_aa = c.take<uint8_t>(f_ub + 1); flags_g = c.take<uint8_t>(E + 1); flags_la = c.take<uint8_t>(Ea + 1);
    cpos_aa = c.take<int32_t>(f_ub + 1); cpos_g = c.take<int32_t>(E + 1); cpos_la = c.take<int32_t>(Ea + 1);
    x_fake = c.take<float>(A * H * 2 + 4); lane_feat = c.take<float>(int64_t(b->L) * 2);
    size_t s1 = 0, s2 = 0, s3 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, s1, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                       int(E > Ea ? E : Ea), 0, 32, (hipStream_t)0);
    (void)scan_flags(nullptr, s2, nullptr, nullptr, int(f_ub + 1), (hipStream_t)0);
    (void)s3;
    cub_bytes = int64_t(s1 > s2 ? s1 : s2) + 256;
    cub_tmp = c.take<uint8_t>(cub_bytes);
    total = c.off + 256;
    ok = c.ok;
  }
};

struct EdgeWs {
  float *aa_geom, *g_geom, *la_geom;
  int32_t *aa_dst, *aa_segptr, *g_src, *g_dst, *g_segptr, *la_dst, *la_segptr;
  int64_t total;
  bool ok;
  EdgeWs(const trajsde_batch* b, const trajsde_graph* g, void* ws, int64_t ws_bytes) {
    Carver c(ws, ws_bytes);
    aa_geom = c.take<float>(4 * int64_t(g->E_aa) + 4); aa_dst = c.take<int32_t>(g->E_aa + 1);
    aa_segptr = c.take<int32_t>(int64_t(b->H) * g->Nt + 1);
    g_geom = c.take<float>(4 * int64_t(g->E_g) + 4); g_src = c.take<int32_t>(g->E_g + 1); g_dst = c.take<int32_t>(g->E_g + 1);
    g_segptr = c.take<int32_t>(b->N + 1);
    la_geom = c.take<float>(4 * int64_t(g->E_la) + 4); la_dst = c.take<int32_t>(g->E_la + 1); la_segptr = c.take<int32_t>(b->N + 1);
    total = c.off + 256;
    ok = c.ok;
  }
};

static int check_batch(const trajsde_batch* b) {
  TS_REQUIRE(b != nullptr, "batch: null");
  TS_REQUIRE(b->N > 0 && b->A >= 0 && b->H > 0 && b->TT >= b->H, "batch: bad sizes");
  TS_REQUIRE(b->E >= 0 && b->E_al >= 0 && b->L >= 0, "batch: negative sizes");
  TS_REQUIRE(int64_t(b->H) * 2 * b->E < (int64_t(1) << 31) - 2, "batch: too many (t, edge) candidates for int32 positions");
  TS_REQUIRE(b->x && b->positions && b->padding_mask && b->bos_mask && b->rotate_angles && b->batch && b->source,
             "batch: null actor tensor");
  TS_REQUIRE(b->A == 0 || b->agent_index, "batch: null agent_index");   // A == 0: no fake agents (forward_ood, ENC:204-370)
  TS_REQUIRE(b->E == 0 || b->edge_index, "batch: null edge_index");
  TS_REQUIRE(b->E_al == 0 || (b->lane_actor_index && b->lane_actor_vectors && b->lane_positions && b->lane_paddings),
             "batch: null lane tensor");
  return TRAJSDE_OK;
}

}  // namespace tsde

using namespace tsde;

extern "C" {

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

int trajsde_graph_prepare(const trajsde_batch* b, const float* rot, float radius, const trajsde_noise* fake_noise, void* ws,
                          int64_t ws_bytes, trajsde_graph* out, void* stream_) {
  if (int rc = check_batch(b)) return rc;
  TS_REQUIRE(rot && ws && out, "graph_prepare: null pointer");
  PrepWs w(b, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "graph_prepare: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int N = b->N, A = b->A, E = b->E, H = b->H, TT = b->TT, Ea = b->E_al, Nt = N + A;
  NoiseArg na{0, nullptr, nullptr};
  if (fake_noise) { na.seed = fake_noise->seed; na.z = fake_noise->z; na.row_ids = fake_noise->row_ids; }

  // CSR by target of edge_index
  if (E > 0) {
    k_split_edges<<<cdiv(E, 256), 256, 0, st>>>(b->edge_index, E, /*row_key=*/1, w.k_in, w.v_in, 0);
    size_t tmp = size_t(w.cub_bytes);
    TS_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub_tmp, tmp, w.k_in, w.csr_dst, w.v_in, w.csr_src, E, 0, bits_for(N), st));
  }
  k_rowptr<<<cdiv(N + 1, 256), 256, 0, st>>>(w.csr_dst, E, N, w.rowptr);
  k_ext_nodes<<<cdiv(Nt, 256), 256, 0, st>>>(N, A, H, b->agent_index, b->batch, b->source, b->bos_mask, w.orig, w.nus, w.eos, w.pick_slot);
  if (A > 0) {
    k_agent_slots<<<cdiv(A, 256), 256, 0, st>>>(A, b->agent_index, w.pick_slot);
    k_fake_x<<<cdiv(A * ((2 * H + 3) / 4), 256), 256, 0, st>>>(A, H, b->x, b->agent_index, na, w.x_fake);
  }
  k_ext_rowptr<<<cdiv(N + 1, 256), 256, 0, st>>>(N, A, E, w.rowptr, b->agent_index, w.ext_rowptr, w.counts);
  // 21 snapshots: flags + prefix sum
  TS_HIP(hipMemsetAsync(w.flags_aa, 0, size_t(w.f_ub + 1), st));
  k_aa_flags<<<cdiv(2 * int64_t(E) + 1, 256), 256, 0, st>>>(N, A, E, H, TT, w.counts, w.csr_src, w.csr_dst, w.rowptr, w.ext_rowptr,
                                                            w.orig, b->padding_mask, b->positions, radius, w.flags_aa);
  {
    size_t tmp = size_t(w.cub_bytes);
    TS_HIP(scan_flags(w.cub_tmp, tmp, w.flags_aa, w.cpos_aa, int(w.f_ub + 1), st));
  }
  // global interactor edges
  k_g_flags<<<cdiv(E + 1, 256), 256, 0, st>>>(E, TT, H - 1, w.csr_src, w.csr_dst, b->padding_mask, w.flags_g);
  {
    size_t tmp = size_t(w.cub_bytes);
    TS_HIP(scan_flags(w.cub_tmp, tmp, w.flags_g, w.cpos_g, E + 1, st));
  }
  // lane-actor edges sorted by actor
  if (b->L > 0) k_lane_feat<<<cdiv(b->L, 256), 256, 0, st>>>(b->L, b->lane_pts, b->lane_positions, b->lane_paddings, w.lane_feat);
  if (Ea > 0) {
    k_split_edges<<<cdiv(Ea, 256), 256, 0, st>>>(b->lane_actor_index, Ea, /*row_key=*/1, w.la_k_in, w.la_v_in, 1);
    size_t tmp = size_t(w.cub_bytes);
    TS_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub_tmp, tmp, w.la_k_in, w.la_actor, w.la_v_in, w.la_eid, Ea, 0, bits_for(N), st));
  }
  k_rowptr<<<cdiv(N + 1, 256), 256, 0, st>>>(w.la_actor, Ea, N, w.la_rowptr);
  k_la_flags<<<cdiv(Ea + 1, 256), 256, 0, st>>>(Ea, w.la_eid, b->lane_actor_vectors, radius, w.flags_la);
  {
    size_t tmp = size_t(w.cub_bytes);
    TS_HIP(scan_flags(w.cub_tmp, tmp, w.flags_la, w.cpos_la, Ea + 1, st));
  }
  k_collect_counts<<<1, 64, 0, st>>>(H, E, Ea, w.cpos_aa, w.cpos_g, w.cpos_la, w.counts);
  TS_LAUNCH_CHECK("graph_prepare kernels");
  int32_t h[4];
  TS_HIP(hipMemcpyAsync(h, w.counts, sizeof(h), hipMemcpyDeviceToHost, st));
  TS_HIP(hipStreamSynchronize(st));
  std::memset(out, 0, sizeof(*out));
  out->Nt = Nt; out->E_ext = h[0]; out->E_aa = h[1]; out->E_g = h[2]; out->E_la = h[3];
  out->orig = w.orig; out->nus_mask = w.nus; out->eos_idx = w.eos; out->pick_slot = w.pick_slot; out->x_fake = w.x_fake;
  return TRAJSDE_OK;
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
  k_aa_compact<<<cdiv(int64_t(out->E_ext) + 1, 256), 256, 0, st>>>(N, A, E, H, TT, w.counts, w.csr_src, w.csr_dst, w.rowptr, w.ext_rowptr, w.orig, b->x,
                                     b->positions, rot, w.flags_aa, w.cpos_aa, e.aa_dst, e.aa_geom);
  k_aa_segptr<<<cdiv(int64_t(H) * Nt + 1, 256), 256, 0, st>>>(Nt, H, w.counts, w.ext_rowptr, w.cpos_aa, e.aa_segptr);
  if (E > 0)
    k_g_compact<<<cdiv(E, 256), 256, 0, st>>>(E, TT, H - 1, w.csr_src, w.csr_dst, b->positions, rot, b->rotate_angles, w.flags_g,
                                              w.cpos_g, e.g_src, e.g_dst, e.g_geom);
  k_segptr_from_rowptr<<<cdiv(N + 1, 256), 256, 0, st>>>(N, w.rowptr, w.cpos_g, e.g_segptr);
  if (Ea > 0)
    k_la_compact<<<cdiv(Ea, 256), 256, 0, st>>>(Ea, w.la_actor, w.la_eid, b->lane_actor_index, b->lane_actor_vectors, w.lane_feat,
                                                rot, w.flags_la, w.cpos_la, e.la_dst, e.la_geom);
  k_segptr_from_rowptr<<<cdiv(N + 1, 256), 256, 0, st>>>(N, w.la_rowptr, w.cpos_la, e.la_segptr);
  TS_LAUNCH_CHECK("graph_compact kernels");
  out->aa_geom = e.aa_geom; out->aa_dst = e.aa_dst; out->aa_segptr = e.aa_segptr;
  out->g_geom = e.g_geom; out->g_src = e.g_src; out->g_dst = e.g_dst; out->g_segptr = e.g_segptr;
  out->la_geom = e.la_geom; out->la_dst = e.la_dst; out->la_segptr = e.la_segptr;
  return TRAJSDE_OK;
}

}  // extern "C"
