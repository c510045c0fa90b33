// kernels.hpp -- declarations of the __global__ entry points defined in attn.hip / recur.hip, launched from stages.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dropout.hpp"
#include "layouts.hpp"
#include "philox.hpp"

namespace tsde {

__global__ void k_aa_center(const float* img, const float* x, const float* x_fake, const float* rot, const uint8_t* bos,
                            const int32_t* orig, int N, int Nt, int H, float* center, float* cn, float* q);
template <bool X6>
__global__ void k_edge_kv(const float* img, const float* geom, const int32_t* dst, const float* q, int64_t E, float* logits, float* v,
                          int heads);
template <int THREADS>
__global__ void k_edge_kv2(const float* img, const float* geom, const int32_t* dst, const float* q, int64_t E, float* logits, float* v,
                           int heads);
// fused edge attention (attn.hip): edge streams of C consecutive edges, one per tile row; records of SEG_REC floats per
// (target, stream) pair at slot target + stream
constexpr int SEG_REC = 96;                       // acc[64] | m[16] | s[16]
constexpr int SEG_STREAMS_512 = 256 * 8 * 32;     // 256 CUs x 8 waves x 32 rows: every wave of a resident 512-thread grid owns 32 streams
struct AttnPlan {
  int C;                                          // edges per stream
  int64_t nstreams;                               // stream_count(E, C): whole blocks of 256
  int64_t rec_slots(int64_t targets) const { return targets + nstreams + 1; }
};
// the list length of a launch: known on the host (dev == null), or only an upper bound `E` with the true value on the device
// (the sync-free forward, trajsde_graph_prepare_async) -- the kernels then derive the same C from it that attn_plan would
struct EdgeCount {
  int64_t E;
  const int32_t* dev;
  int streams_target;
};
__device__ __forceinline__ int64_t edge_count(const EdgeCount& c) { return c.dev ? int64_t(*c.dev) : c.E; }
__device__ __forceinline__ int stream_len(const EdgeCount& c, int64_t E, int C_host) {
  if (!c.dev) return C_host;
  const int C = int((E + c.streams_target - 1) / c.streams_target);
  return C < 1 ? 1 : C;
}
// Streams are NOT equally long.  The list is cut into blocks of 256 streams and 256 C edges -- one workgroup of the fused edge attention
// (8 waves x 32 rows) --, and inside a block the first 128 streams (the four waves that reach their SIMDs first) are Co = 2 C - Cy long,
// the last 128 (the waves that share those SIMDs as the younger ones) Cy ~ 3/4 C.  A SIMD serves its older wave first (DESIGN section 5,
// finding 2): with equal streams the older waves of the one resident round finish at 2/3 of the kernel's time and the younger ones run the
// last third alone; giving the older waves 5/8 of the edges makes both finish together (-4.5 % on the kernel; 60 % / 88 % for the
// younger streams measured worse).  Everything that maps edges to streams goes through these three functions.
struct StreamMap {
  int C, Co, Cy;
};
__host__ __device__ __forceinline__ StreamMap stream_map(int C) {
#ifdef TSDE_EQUAL_STREAMS
  return StreamMap{C, C, C};
#else
#ifndef TSDE_STREAM_SHORT_PCT
#define TSDE_STREAM_SHORT_PCT 70
#endif
  int Cy = (TSDE_STREAM_SHORT_PCT * C + 50) / 100;
  Cy = Cy < 1 ? 1 : (Cy > C ? C : Cy);
  return StreamMap{C, 2 * C - Cy, Cy};
#endif
}
__host__ __device__ __forceinline__ int64_t stream_count(int64_t E, int C) {      // whole blocks: some streams of the last one are empty
  const int64_t per_block = int64_t(256) * C;
  return (E + per_block - 1) / per_block * 256;
}
__device__ __forceinline__ int64_t stream_base(const StreamMap& m, int64_t s) {   // first edge of stream s
  const int r = int(s & 255);
  return (s >> 8) * (int64_t(256) * m.C) + (r < 128 ? int64_t(r) * m.Co : int64_t(128) * m.Co + int64_t(r - 128) * m.Cy);
}
__device__ __forceinline__ int stream_length(const StreamMap& m, int64_t s) { return (s & 255) < 128 ? m.Co : m.Cy; }
__device__ __forceinline__ int64_t stream_of(const StreamMap& m, int64_t e) {     // the stream that holds edge e
  const int64_t per_block = int64_t(256) * m.C, blk = e / per_block, r = e - blk * per_block, half = int64_t(128) * m.Co;
  return blk * 256 + (r < half ? r / m.Co : 128 + (r - half) / m.Cy);
}
// The records of the fused edge attention (16-row-tile layout) merged by the kernel that consumes the aggregate (k_node_update)
// instead of by k_seg_merge: rec == nullptr means "read the agg rows".  cv: the per-target constant of the v rows (EdgeL6F::CV).
struct SegMerge {
  const float* rec;
  const int32_t* segptr;
  EdgeCount ec;
  int C_host;
  const float* cv;
};
inline SegMerge no_merge() { return SegMerge{nullptr, nullptr, EdgeCount{0, nullptr, 0}, 0, nullptr}; }
inline AttnPlan attn_plan(int64_t E, int streams_target = SEG_STREAMS_512) {
  AttnPlan p;
  p.C = int((E + streams_target - 1) / streams_target);
  if (p.C < 1) p.C = 1;
  p.nstreams = stream_count(E, p.C);
  return p;
}
template <int NT, bool DROP, bool SAVE, int LIST, bool H8>
__global__ void k_edge_attn2(const float* img, const float* geom, const int32_t* dst, const float* q, EdgeCount ec, int C, float* rec, int heads,
                             const int32_t* segptr, DropArg drop, float* emb_out);
template <int LIST>
__global__ void k_edge_attn2p(const float* img, const float* geom, const int32_t* dst, const float* q, EdgeCount ec, int C, float* rec, int heads);
// host side of the fused edge attention (stages.hip): k_edge_attn2 + k_seg_merge -> agg [R,64]; the training path also asks
// for the embedding rows (emb_out [E,64]) and the softmax statistics (stats [R,heads,2]).  img: the stage blob's EdgeL6F image
bool attn_fused_enabled();
int64_t fused_rec_floats(int64_t E, bool exact, int64_t targets);
int fused_edge_attention(const char* tag, bool dominant, const float* img, const float* geom, const int32_t* dst, const float* q,
                         const EdgeCount& ec, const int32_t* segptr, int64_t R, float* rec, float* agg, int heads, hipStream_t st,
                         const DropArg& drop, float* emb_out = nullptr, float* stats = nullptr, SegMerge* defer = nullptr);
// the same attention on the matrix cores (gattn.hip): inference, 8 heads, fp32 rows, no dropout; an alternative, TRAJSDE_GATTN_MM=1 selects it
bool gattn_mm_enabled();
// gattn_h3.hip: the same attention as fp16x3 products on rel rows stored split by their producer (inference, 8 heads, fp32 state)
bool rel_split_enabled();
int launch_global_attn_h3(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, hipStream_t st);
bool rel_split_scene_cache();
int launch_scene_ptr(const int64_t* scene_of, int N, int A, const int32_t* esrc, const int32_t* edst, const int32_t* segptr, int64_t E_bound,
                     int32_t* scene_ptr, hipStream_t st);
int launch_global_attn_sc(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, int A, const int64_t* scene_of, const int32_t* scene_ptr, float* agg, hipStream_t st);
// gattn_f32.hip: the same attention with its rel-row products on the fp32 matrix cores (8 heads, fp32 rows)
bool gattn_f32mm_enabled();
int launch_global_attn_mf(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, float* stats, const DropArg& drop, hipStream_t st);
int launch_global_attn_mm(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                          const float* vn, int64_t N, float* agg, hipStream_t st);
// launch the instantiation selected by (heads, bf16 state storage, dropout)
#define TS_GLOBAL_ATTN(heads, bf16, drop, ...)                                                                       \
  do {                                                                                                              \
    const bool _d = (drop).p > 0.f;                                                                                 \
    if ((heads) == 4) {                                                                                             \
      if (bf16) { if (_d) TS_LAUNCH_TAG("k_global_attn<4>", false, (k_global_attn<4, true, true, true>), __VA_ARGS__, drop); else TS_LAUNCH_TAG("k_global_attn<4>", false, (k_global_attn<4, true, false, true>), __VA_ARGS__, drop); } \
      else { if (_d) TS_LAUNCH_TAG("k_global_attn<4>", false, (k_global_attn<4, false, true, true>), __VA_ARGS__, drop); else TS_LAUNCH_TAG("k_global_attn<4>", false, (k_global_attn<4, false, false, true>), __VA_ARGS__, drop); } \
    } else {                                                                                                        \
      if (bf16) { if (_d) TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn<8, true, true, true>), __VA_ARGS__, drop); else TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn<8, true, false, true>), __VA_ARGS__, drop); } \
      else { if (_d) TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn<8, false, true, true>), __VA_ARGS__, drop); else TS_LAUNCH_TAG("k_global_attn<8>", false, (k_global_attn<8, false, false, true>), __VA_ARGS__, drop); } \
    }                                                                                                               \
  } while (0)
// the same kernel over edge rows alone (NODE = false: AA / AL attention on stored embedding rows, training path):
// TS_EDGE_ATTN(heads, drop, grid, block, lds, stream, img /*GAttnL*/, segptr, emb, q, R, agg, stats)
#define TS_EDGE_ATTN(heads, drop, grid, block, lds, st, img, segptr, emb, q, R, agg, stats)                          \
  do {                                                                                                              \
    const bool _d = (drop).p > 0.f;                                                                                 \
    const int32_t* _ns = nullptr;                                                                                   \
    const float* _nf = nullptr;                                                                                     \
    if ((heads) == 4) {                                                                                             \
      if (_d) TS_LAUNCH_TAG("k_edge_attn_rows<4>", false, (k_global_attn<4, false, true, false>), grid, block, lds, st, img, segptr, _ns, emb, q, _nf, _nf, R, agg, stats, drop); \
      else TS_LAUNCH_TAG("k_edge_attn_rows<4>", false, (k_global_attn<4, false, false, false>), grid, block, lds, st, img, segptr, _ns, emb, q, _nf, _nf, R, agg, stats, drop); \
    } else {                                                                                                        \
      if (_d) TS_LAUNCH_TAG("k_edge_attn_rows<8>", false, (k_global_attn<8, false, true, false>), grid, block, lds, st, img, segptr, _ns, emb, q, _nf, _nf, R, agg, stats, drop); \
      else TS_LAUNCH_TAG("k_edge_attn_rows<8>", false, (k_global_attn<8, false, false, false>), grid, block, lds, st, img, segptr, _ns, emb, q, _nf, _nf, R, agg, stats, drop); \
    }                                                                                                               \
  } while (0)
__global__ void k_seg_merge(const int32_t* segptr, const float* rec, EdgeCount ec, int C, int64_t R, float* agg, float* stats, int heads,
                            const float* img, const float* q, int add_cv, int rec_layout);
// the fused edge attention on 32x32x16 matrix tiles (edge32.hip): same arguments, records in layout 1
template <bool DROP, bool SAVE, bool PP>
__global__ void k_edge_attn3(const float* img, const float* geom, const int32_t* dst, const float* q, EdgeCount ec, int C, float* rec, int heads,
                             const int32_t* segptr, DropArg drop, float* emb_out);
template <bool X6>
__global__ void k_edge_embed(const float* img, const float* geom, EdgeCount ec, float* emb_out, int st_bf16);
bool rel_embed_fused();
constexpr int edge_embed2_lds(int threads) { return (EdgeL6G::SIZE + (threads / 64) * ROWSTAGE) * 4; }   // image + one store tile per wave
__global__ void k_edge_embed2(const float* img, const float* geom, EdgeCount ec, float* emb_out, int st_bf16);   // fp16x3 build: the fused form
template <bool X6>
__global__ void k_global_edge(const float* img, const float* rel, const int32_t* src, const int32_t* dst, const float* q,
                              const float* kn, const float* vn, int64_t E, float* logits, float* v);
template <int HEADS, bool ST_BF16, bool DROP, bool NODE>
__global__ void k_global_attn(const float* img, const int32_t* segptr, const int32_t* src, const float* rel, const float* q, const float* kn,
                              const float* vn, int64_t N, float* agg, float* stats, DropArg drop);
__global__ void k_seg_softmax_agg(const int32_t* segptr, const float* logits, const float* v, int64_t R, float* agg, int heads, DropArg drop);
template <bool X6>
__global__ void k_node_update(const float* img, const float* agg, const float* xn, const float* x, int64_t R, float* x1, float* xn2, DropArg drop,
                              SegMerge mg);
__global__ void k_ffn6(const float* img, const float* x1, const float* xn2, int64_t R, float* out, DropArg drop);
__global__ void k_ffn(const float* img, const float* x1, const float* xn2, int64_t R, float* out, DropArg drop, int out_bf16);
template <int NQ, bool SPLITKV = false>
__global__ void k_node_proj(const float* img, const float* x, int64_t R, float* xn_out, float* p0, float* p1, float* p2);
__global__ void k_mode_proj(const float* norm_g, const float* proj_g, const float* x, int64_t N, float* out);

__global__ void k_enc_sde_step(const float* img, const float* h_in, const float* hidden0, int Nt, float dt, float sq, float sn, float cs,
                               int idx, int noise_step0, NoiseArg na, const uint8_t* nus, const int32_t* eos, const int32_t* pick_slot, float* h_ode,
                               float* diff_pick);
__global__ void k_enc_gru_step(const float* img, const float* h_ode, const float* x_t, int Nt, int N, int t, int TT, int idx,
                               const uint8_t* pad, const int32_t* orig, const int32_t* eos, float* h_out, float* local_out,
                               float* latent_t);

constexpr int COOP_TMAX = 4;                                  // row tiles one workgroup can own (register / LDS budget)
constexpr int COOP_RS = 68;                                   // padded row stride of an LDS activation tile
constexpr int COOP_TILE = 16 * COOP_RS;
// LDS floats of the cooperative recurrence for `tiles` row tiles per workgroup: per tile the fp32 state + five operand tiles
// (1024 floats each as pre-split fp16 planes in the fp16x3 build, a padded fp32 tile otherwise: recur.hip) + 64 partial sums
constexpr int COOP_BIAS_VECS = 21;                            // + the bias / head vectors of the three nets and the GRU unit (recur.hip BV)
// + the three time-conditioned first-layer biases of every step (32 steps x 3 x 64: recur.hip BT)
constexpr int coop_lds_floats(int tiles) {
  return tiles * (COOP_TILE + 5 * (TSDE_SPLIT_H3 ? 1024 : COOP_TILE)) + tiles * 64 + COOP_BIAS_VECS * 64 + 32 * 3 * 64;
}
struct StepTab {                                              // per-iteration (dt, sqrt_h, sin t0, cos t0), H <= 32
  float dt[32], sq[32], sn[32], cs[32];
};
// activation slabs [H][Nt][64] (GS: [H][Nt]) the training forward keeps for the recurrence backward (encoder_bwd.hip EncBwdWs)
struct RecurTape {
  float *HIN, *H1, *H2, *G1, *G2, *GS, *HODE, *XS, *U1, *R1, *UU, *RR, *RH, *N1, *NW;
};
template <int TW, bool SAVE>
__global__ void k_enc_recur_coop(const float* sde_img, const float* gru_img, const float* coop6, const float* h0, const float* aa_out, int Nt, int N, int H,
                                 int TT, int tiles_per_wg, StepTab tab, int noise_step0, NoiseArg na, const uint8_t* nus, const uint8_t* pad,
                                 const int32_t* orig, const int32_t* eos, const int32_t* pick_slot, float* kept, float* diff_pick,
                                 float* latent_ys, int aa_bf16, RecurTape tp);
// the recurrence backward in the cooperative form (recur.hip k_enc_recur_bwd_coop)
struct RecurBwdCoopArgs {
  const float *gru_t, *sde_t;          // GruBwdL / EncSdeBwdL images: every matrix transposed, split-precision fragments
  int Nt, N, H, TT;
  NoiseArg na;
  const uint8_t *pad, *nus;
  const int32_t *orig, *eos;
  const float *dlat, *DLDG;
  RecurTape tp;                        // the forward's activation slabs
  float *DNW, *DN1P, *DUP, *DRP, *DU1, *DR1, *DAA, *DF, *DH2, *DH1, *DG2N, *DG1N, *DG2A, *DG1A, *DGPN, *DGPA, *dh_out;
  float dt[32], sq[32];
};
constexpr int COOP_BWD_TILES = 6;                                 // operand tiles per row tile (+ 64 partial dots, 64 row magnitudes)
constexpr int coop_bwd_lds_floats(int tiles) { return tiles * (COOP_BWD_TILES * (TSDE_SPLIT_H3 ? 1024 : COOP_TILE) + 128); }
template <int TW>
__global__ void k_enc_recur_bwd_coop(RecurBwdCoopArgs a);
// the decoder backward's forward replay in the cooperative form (recur.hip k_sde_replay_coop; fp16x3 build).  LDS: four operand tiles + 64 floats
constexpr int SDE_REPLAY_COOP_LDS_BYTES = (4 * 1024 + 64) * 4;
__global__ void k_sde_replay_coop(const float* img, const int32_t* best, int N, int K, int n_euler, const float* step_tab, NoiseArg na,
                                  float* states, float* H1, float* H2, float* G1, float* G2, float* GS);
// the decoder's reverse sweep in the cooperative form (recur.hip k_sde_bwd_coop; fp16x3 build).  LDS: five operand tiles + 128 floats;
// vpart: one row of SDE_SWEEP_V_FLOATS per workgroup (d diffusion.4.weight [64], its bias at [64])
constexpr int SDE_SWEEP_V_FLOATS = 68;
constexpr int SDE_BWD_COOP_LDS_BYTES = (5 * 1024 + 128) * 4;
struct SdeBwdCoopArgs {
  const float* img;                 // SweepL
  const int32_t* best;
  int N, K, T, n_euler;
  const float *step_tab, *out_tab;
  NoiseArg na;
  const float *H1, *H2, *G1, *G2, *GS, *DS;
  float *DH1, *DH2, *DF, *DG1, *DG2, *DY0, *vpart;     // vpart: one row of SweepV::SIZE floats per WORKGROUP
};
__global__ void k_sde_bwd_coop(SdeBwdCoopArgs a);
// vanilla HiVT variant (grid.hip)
__global__ void k_tr_prep(const float* aa_out, const uint8_t* pad, const float* tok, int N, int TT, float* X);
template <int HEADS, bool DROP>
__global__ void k_tr_attention(const float* q, const float* k, const float* v, int N, float* o, DropArg drop);
__global__ void k_tr_outproj(const float* img, const float* o, const float* x, int64_t R, float* x1, float* xn2, DropArg drop);
int launch_tr_attention(int heads, const float* q, const float* k, const float* v, int N, float* o, const DropArg& drop, hipStream_t st);
__global__ void k_tr_final(const float* norm, const float* x, int N, float* out);

__global__ void k_ood_stats(const float* samples, int S, int N, float* mean, float* stds);

}  // namespace tsde
