// stages.hip -- C-ABI entry points of the encoder and aggregator stages: workspace carving and the launch
// sequences over the kernels of attn.hip / recur.hip.  Everything is enqueued on the caller's stream; nothing
// is allocated and nothing synchronises here.
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"

#include <algorithm>
#include <cstdlib>

namespace tsde {

// workgroup sizes; overridable for sweeps: TRAJSDE_THREADS_EDGE / _NODE / _RECUR (multiples of 64); env_threads: common.hpp
static bool edge_x6() { static bool v = []() { const char* e = getenv("TRAJSDE_EDGE_FP32"); return !(e && atoi(e) != 0); }(); return v; }
static bool edge_pair() { static bool v = []() { const char* e = getenv("TRAJSDE_EDGE_PAIR"); return !(e && atoi(e) == 0); }(); return v; }   // two tiles per wave (default on)
static int pair_threads() { static int t = []() { const char* e = getenv("TRAJSDE_PAIR_THREADS"); const int v = e ? atoi(e) : 768; return v == 512 ? 512 : 768; }(); return t; }
// fused edge attention (k_edge_attn2 + k_seg_merge: no per-edge v / logits in HBM): default, inference and training forward
// alike; TRAJSDE_ATTN_FUSED=0 runs the older two-kernel form (k_edge_kv2 -> HBM -> k_seg_softmax_agg), kept as a cross-check
static bool attn_fused() { static bool v = []() { const char* e = getenv("TRAJSDE_ATTN_FUSED"); return !(e && atoi(e) == 0); }(); return v && edge_x6() && edge_pair() && TSDE_SPLIT_H3; }
static int fused_threads() {                 // TRAJSDE_FUSED_THREADS=256: one wave per SIMD (diagnostic runs of the phase stamps)
  static const int t = [] { const char* e = getenv("TRAJSDE_FUSED_THREADS"); const int v = e ? atoi(e) : 512; return v == 256 ? 256 : 512; }();
  return t;
}   // 2 waves per SIMD: ~220 VGPRs, weight image + 8 x 8 KB of parked query rows = 150 KB of LDS
// TRAJSDE_EDGE_TILE=32: the fused edge attention on 32x32x16 matrix tiles (edge32.hip) -- half the matrix instructions and 12 %
// fewer vector instructions, parity-tested, and 5 % slower un-profiled (0.839 against 0.800 ms on one box): the default stays
// the 16x16x32 form.  TRAJSDE_EDGE_PINGPONG=1 adds its phase barriers that keep the second wave of a SIMD one phase behind the
// first (edge32.hip phase_sync): slower still, kept as the record of the experiment.
// TRAJSDE_FUSED_TILES=1: the inference instantiation with one tile per wave and 16 waves per workgroup (4 waves per SIMD, 116
// VGPRs, weight fragments not shared between tiles): bit-identical, and within the noise of the default (0.85 vs 0.86-0.91 ms)
// TRAJSDE_MERGE_KERNEL=1: the records are merged by k_seg_merge into agg rows (always so in training) instead of inside k_node_update
static bool merge_in_update() { static const bool v = []() { const char* e = getenv("TRAJSDE_MERGE_KERNEL"); return !(e && atoi(e) != 0); }(); return v; }
// TRAJSDE_EDGE_PIPE=1: the software-pipelined form of the fused edge attention (k_edge_attn2p: the wave's two tiles one stage apart,
// vector work issued between the other tile's matrix instructions; bit-identical) for inference.  Measured slower than the lockstep
// form (0.87 against 0.81 ms per launch; one wave per SIMD 1.12 against 0.98) -- HISTORY.md section 5 -- so it is an alternative.
static bool edge_pipe() { static const bool v = []() { const char* e = getenv("TRAJSDE_EDGE_PIPE"); return e && atoi(e) == 1; }(); return v; }
static bool fused_one_tile() { static const bool v = []() { const char* e = getenv("TRAJSDE_FUSED_TILES"); return e && atoi(e) == 1; }(); return v; }
static bool edge_pingpong() { static const bool v = []() { const char* e = getenv("TRAJSDE_EDGE_PINGPONG"); return e && atoi(e) != 0; }(); return v; }
static bool edge_tile32() { static const bool v = []() { const char* e = getenv("TRAJSDE_EDGE_TILE"); return e && atoi(e) == 32; }(); return v && TSDE_SPLIT_H3; }
static bool global_fused_env() { static const bool v = []() { const char* e = getenv("TRAJSDE_GLOBAL_UNFUSED"); return !(e && atoi(e) != 0); }(); return v; }
static int fused_streams() { return 256 * 256; }             // a workgroup walks 256 streams (8 waves x 2 tiles or 4 waves x 4 tiles), one workgroup per CU
static AttnPlan fused_plan(int64_t E) { return attn_plan(E, fused_streams()); }
// record slots of a list whose length is only bounded by E: any E' <= E cuts into at most min(E, streams) streams
static int64_t fused_rec_slots(int64_t E, bool exact, int64_t targets) {
  return exact ? fused_plan(E).rec_slots(targets) : targets + std::min<int64_t>(E, fused_streams()) + 1;
}
// an inexact graph (trajsde_graph_prepare_async) carries its list lengths on the device
static EdgeCount count_of(const trajsde_graph* g, int which, int64_t E) {
  return EdgeCount{E, g->exact ? nullptr : g->counts + which, fused_streams()};
}
static int threads_edge() { static int t = env_threads("TRAJSDE_THREADS_EDGE", 1024); return t; }
static int threads_node() { static int t = env_threads("TRAJSDE_THREADS_NODE", 512); return t; }
static int threads_recur() { static int t = env_threads("TRAJSDE_THREADS_RECUR", 256); return t; }

struct EncWs {
  float *center, *cn, *q, *logits, *v, *agg, *x1, *xn2, *aa_out, *hA, *hB, *lat, *al_xn, *al_q, *al_logits, *al_v, *al_agg, *al_x1,
      *al_xn2;
  float *rec, *al_rec;          // fused form: (target, stream) records instead of per-edge logits / v
  int64_t total;
  bool ok;
  EncWs(const trajsde_batch* b, const trajsde_graph* g, void* ws, int64_t bytes) {
    Carver c(ws, bytes);
    const int64_t R = int64_t(b->H) * g->Nt, N = b->N;
    const bool fused = attn_fused();
    center = c.take<float>(R * 64); cn = c.take<float>(R * 64); q = c.take<float>(R * 64);
    rec = c.take<float>(fused ? fused_rec_slots(g->E_aa, g->exact != 0, R) * SEG_REC : 4);
    al_rec = c.take<float>(fused ? fused_rec_slots(g->E_la, g->exact != 0, N) * SEG_REC : 4);
    logits = c.take<float>(fused ? 8 : int64_t(g->E_aa) * 8 + 8); v = c.take<float>(fused ? 64 : int64_t(g->E_aa) * 64 + 64);
    agg = c.take<float>(R * 64); x1 = c.take<float>(R * 64); xn2 = c.take<float>(R * 64); aa_out = c.take<float>(R * 64);
    hA = c.take<float>(int64_t(g->Nt) * 64); hB = c.take<float>(int64_t(g->Nt) * 64); lat = c.take<float>(N * 64);
    al_xn = c.take<float>(N * 64); al_q = c.take<float>(N * 64);
    al_logits = c.take<float>(fused ? 8 : int64_t(g->E_la) * 8 + 8); al_v = c.take<float>(fused ? 64 : int64_t(g->E_la) * 64 + 64);
    al_agg = c.take<float>(N * 64); al_x1 = c.take<float>(N * 64); al_xn2 = c.take<float>(N * 64);
    total = c.off + 256;
    ok = c.ok;
  }
};

struct AggWs {
  float *rel, *xn, *q, *kn, *vn, *logits, *v, *agg, *x1, *xn2, *xa, *xb;
  int64_t total;
  bool ok;
  AggWs(const trajsde_batch* b, const trajsde_graph* g, void* ws, int64_t bytes) {
    Carver c(ws, bytes);
    const int64_t N = b->N, E = g->E_g;
    rel = c.take<float>(E * 64 + 64);
    xn = c.take<float>(N * 64); q = c.take<float>(N * 64); kn = c.take<float>(N * 64); vn = c.take<float>(N * 64);
    logits = c.take<float>(E * 8 + 8); v = c.take<float>(E * 64 + 64);
    agg = c.take<float>(N * 64); x1 = c.take<float>(N * 64); xn2 = c.take<float>(N * 64); xa = c.take<float>(N * 64); xb = c.take<float>(N * 64);
    total = c.off + 256;
    ok = c.ok;
  }
};

// one attention block over an edge list already reduced to (logits, v): softmax-aggregate, gated update, FFN
// fp16x3 build: the plain images are split-precision images of the same size, so the one-pass FFN (k_ffn, 128 KB image)
// replaces the two-half k_ffn6 that the 1.5x larger bf16x6 planes needed; TRAJSDE_NODE_FP32=0/1 forces either pair
static bool node_x6() {
  static bool v = []() {
    const char* e = getenv("TRAJSDE_NODE_FP32");
    return e ? atoi(e) == 0 : !TSDE_SPLIT_H3;
  }();
  return v;
}

// gated update + FFN of one attention block; `upd6/ffn6` are the split-precision images, `upd/ffn` the plain fp32 ones
struct NodeImgs {
  const float *upd, *ffn, *upd6, *ffn6;
};
static int update_ffn(const NodeImgs& im, const float* agg, const float* xn, const float* x, int64_t R, float* x1, float* xn2, float* out,
                      hipStream_t st, const DropArg& drop = no_drop(), bool out_bf16 = false, const SegMerge& mg = no_merge()) {
  const int64_t ntiles = (R + 15) / 16;
  TS_REQUIRE(!(out_bf16 && node_x6()), "bf16 state storage needs the one-pass FFN kernel (default fp16x3 build, TRAJSDE_NODE_FP32 unset)");
  if (node_x6()) {
    TS_LAUNCH(k_node_update<true>, tile_grid(ntiles, threads_node(), UpdL6::SIZE * 4), threads_node(), UpdL6::SIZE * 4, st, im.upd6, agg, xn, x,
              R, x1, xn2, drop, mg);
    TS_LAUNCH(k_ffn6, tile_grid(ntiles, threads_node(), FfnL6::HALF * 4), threads_node(), FfnL6::HALF * 4, st, im.ffn6, x1, xn2, R, out, drop);
  } else {
    TS_LAUNCH(k_node_update<false>, tile_grid(ntiles, threads_node(), UpdL::SIZE * 4), threads_node(), UpdL::SIZE * 4, st, im.upd, agg, xn, x, R,
              x1, xn2, drop, mg);
    TS_LAUNCH(k_ffn, tile_grid(ntiles, threads_node(), FfnL::SIZE * 4), threads_node(), FfnL::SIZE * 4, st, im.ffn, x1, xn2, R, out, drop, out_bf16 ? 1 : 0);
  }
  return TRAJSDE_OK;
}
static int attention_tail(const NodeImgs& im, const int32_t* segptr, const float* logits, const float* v, const float* xn, const float* x,
                          int64_t R, float* agg, float* x1, float* xn2, float* out, hipStream_t st, int heads = 8,
                          const DropArg& drop = no_drop()) {
  TS_LAUNCH(k_seg_softmax_agg, cdiv(R, 4), 256, 0, st, segptr, logits, v, R, agg, heads, drop);
  return update_ffn(im, agg, xn, x, R, x1, xn2, out, st, drop);
}

// embedding + lin_k|lin_v + softmax-aggregate of one edge list in the fused form: records, then one merged agg row per target
int fused_edge_attention(const char* tag, bool dominant, const float* img, const float* geom, const int32_t* dst, const float* q,
                         const EdgeCount& ec, const int32_t* segptr, int64_t R, float* rec, float* agg, int heads, hipStream_t st,
                         const DropArg& drop, float* emb_out, float* stats, SegMerge* defer) {
  const int64_t E = ec.E;
  const AttnPlan pl = fused_plan(E);
  // (k_edge_attn2 walks the list at 32-bit byte offsets through buffer descriptors: 16 B of geometry per edge)
  TS_REQUIRE(E < (int64_t(1) << 28), "fused edge attention: an edge list of 2^28 or more entries exceeds its 32-bit byte offsets; split the batch");
  if (E > 0) {
    const int threads = fused_threads();
    // a bounded list: enough waves for the most streams any E' <= E cuts into; the ones beyond the true count leave at once
    const int64_t streams = ec.dev ? std::min<int64_t>(E, fused_streams()) : pl.nstreams;
    const int per_wg = (fused_one_tile() ? 1024 : threads) / 64 * (fused_one_tile() ? 16 : 32);      // streams a workgroup walks (256 by default)
    const int grid = xcd_grid((streams + per_wg - 1) / per_wg);
    const int lds = (EdgeL6F::LDS_SIZE + 256 * 64) * 4;                      // weight image + the parked query rows of 256 streams (64 KB)
    const bool d = drop.p > 0.f, sv = emb_out != nullptr;
#define TS_EA2H(N_, D_, S_, L_, H_) TS_LAUNCH_TAG(tag, dominant, (k_edge_attn2<N_, D_, S_, L_, H_>), grid, threads, lds, st, img, geom, dst, q, ec, pl.C, rec, heads, segptr, drop, emb_out)
#define TS_EA2L(N_, D_, S_, L_) do { if (heads == 8) TS_EA2H(N_, D_, S_, L_, true); else TS_EA2H(N_, D_, S_, L_, false); } while (0)
#define TS_EA2(N_, D_, S_) do { if (dominant) TS_EA2L(N_, D_, S_, 0); else TS_EA2L(N_, D_, S_, 1); } while (0)
#define TS_EA3(D_, S_) if (edge_pingpong()) TS_LAUNCH_TAG(tag, dominant, (k_edge_attn3<D_, S_, true>), grid, threads, lds, st, img + EdgeL6F::SIZE, geom, dst, q, ec, pl.C, rec, heads, segptr, drop, emb_out); else TS_LAUNCH_TAG(tag, dominant, (k_edge_attn3<D_, S_, false>), grid, threads, lds, st, img + EdgeL6F::SIZE, geom, dst, q, ec, pl.C, rec, heads, segptr, drop, emb_out)
#if !TSDE_SPLIT_H3
    (void)grid; (void)lds; (void)threads;
    return fail(TRAJSDE_ERR_UNSUPPORTED, "the fused edge attention (and with it the training forward) exists in the fp16x3 build only");
#else
#ifdef TSDE_PRODUCT
    // the product library carries the default form only; a switch that asks for an alternative one is refused, not ignored
    TS_REQUIRE(!edge_tile32() && !edge_pipe() && !fused_one_tile(),
               "TRAJSDE_EDGE_TILE / TRAJSDE_EDGE_PIPE / TRAJSDE_FUSED_TILES select alternative kernel forms that live in "
               "trajsde_amd/variants/libtrajsde_alt.so: point TRAJSDE_LIB at it");
    {
#else
    if (edge_tile32()) {
#if TSDE_SPLIT_H3
      if (d && sv) TS_EA3(true, true);
      else if (d) TS_EA3(true, false);
      else if (sv) TS_EA3(false, true);
      else TS_EA3(false, false);
#endif
    } else if (edge_pipe() && !d && !sv && !fused_one_tile()) {
      // inference: the two tiles of a wave one stage apart, vector work issued between the other tile's matrix instructions
      if (dominant) TS_LAUNCH_TAG(tag, dominant, (k_edge_attn2p<0>), grid, threads, lds, st, img, geom, dst, q, ec, pl.C, rec, heads);
      else TS_LAUNCH_TAG(tag, dominant, (k_edge_attn2p<1>), grid, threads, lds, st, img, geom, dst, q, ec, pl.C, rec, heads);
    } else if (fused_one_tile() && !d && !sv) {
      if (dominant) TS_LAUNCH_TAG(tag, dominant, (k_edge_attn2<1, false, false, 0, false>), grid, 1024, lds, st, img, geom, dst, q, ec, pl.C, rec, heads, segptr, drop, emb_out);
      else TS_LAUNCH_TAG(tag, dominant, (k_edge_attn2<1, false, false, 1, false>), grid, 1024, lds, st, img, geom, dst, q, ec, pl.C, rec, heads, segptr, drop, emb_out);
    } else {
#endif
      if (d && sv) TS_EA2(2, true, true);
      else if (d) TS_EA2(2, true, false);
      else if (sv) TS_EA2(2, false, true);
      else TS_EA2(2, false, false);
    }
#endif   // TSDE_SPLIT_H3
#undef TS_EA2
#undef TS_EA2L
#undef TS_EA3
  }
  // inference with the default record layout: the consumer of the aggregate (k_node_update) merges the records itself
  if (defer != nullptr && stats == nullptr && emb_out == nullptr && drop.p == 0.f && !edge_tile32() && merge_in_update()) {
    *defer = SegMerge{rec, segptr, ec, pl.C, img + EdgeL6F::CV};
    return TRAJSDE_OK;
  }
  if (defer != nullptr) *defer = no_merge();
  TS_LAUNCH(k_seg_merge, cdiv(R, 4), 256, 0, st, segptr, rec, ec, pl.C, R, agg, stats, heads, img, q, drop.p > 0.f ? 0 : 1, edge_tile32() ? 1 : 0);
  return TRAJSDE_OK;
}
bool attn_fused_enabled() { return attn_fused(); }
bool rel_embed_fused() {            // TRAJSDE_REL_EMBED_FUSED=0: the one-tile kernel on the plain split image (A/B runs)
  static const bool v = []() { const char* e = getenv("TRAJSDE_REL_EMBED_FUSED"); return !(e && atoi(e) == 0); }();
  return v;
}
int64_t fused_rec_floats(int64_t E, bool exact, int64_t targets) { return fused_rec_slots(E, exact, targets) * SEG_REC; }

}  // namespace tsde

using namespace tsde;

// ---- trajsde_encoder_fork_stream (ABI 8): see include/trajsde_hip.h
static thread_local hipStream_t t_fork_stream = nullptr;
static thread_local hipEvent_t t_fork_event = nullptr;


extern "C" {

int trajsde_sync_free_supported(void) { return attn_fused() && global_fused_env() ? 1 : 0; }

int64_t trajsde_encoder_ws_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EncWs w(b, g, nullptr, 0);
  return w.total;
}

// AAEncoder on the H snapshots at once (ENC:112-121, 538-566) -> aa_out [H, Nt, 64]
static int run_aa_encoder(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob, EncWs& w, float* aa_out,
                          hipStream_t st, int heads = 8, const DropArg& drop = no_drop()) {
  const int N = b->N, Nt = g->Nt, H = b->H;
  const int64_t R = int64_t(H) * Nt;
  TS_LAUNCH(k_aa_center, tile_grid((R + 15) / 16, 512, AaCenterL::SIZE * 4), 512, AaCenterL::SIZE * 4, st, blob + EncBlob::AA_CENTER,
            b->x, g->x_fake, rot, b->bos_mask, g->orig, N, Nt, H, w.center, w.cn, w.q);
  const NodeImgs im{blob + EncBlob::AA_UPD, blob + EncBlob::AA_FFN, blob + EncBlob::AA_UPD6, blob + EncBlob::AA_FFN6};
  if (attn_fused()) {
    SegMerge mg = no_merge();
    if (int rc = fused_edge_attention("k_edge_kv[aa]", true, blob + EncBlob::AA_EDGE6F, g->aa_geom, g->aa_dst, w.q, count_of(g, 1, g->E_aa), g->aa_segptr, R,
                                      w.rec, w.agg, heads, st, drop, nullptr, nullptr, &mg))
      return rc;
    return update_ffn(im, w.agg, w.cn, w.center, R, w.x1, w.xn2, aa_out, st, drop, state_bf16(), mg);  // aa_out in the state storage type
  }
  if (g->E_aa > 0) {
    if (edge_x6() && edge_pair())
    {
      if (pair_threads() == 768)
        TS_LAUNCH_TAG("k_edge_kv[aa]", true, k_edge_kv2<768>, tile_grid((int64_t(g->E_aa) + 31) / 32, 768, EdgeL6::SIZE * 4), 768, EdgeL6::SIZE * 4,
                      st, blob + EncBlob::AA_EDGE6, g->aa_geom, g->aa_dst, w.q, int64_t(g->E_aa), w.logits, w.v, heads);
      else
        TS_LAUNCH_TAG("k_edge_kv[aa]", true, k_edge_kv2<512>, tile_grid((int64_t(g->E_aa) + 31) / 32, 512, EdgeL6::SIZE * 4), 512, EdgeL6::SIZE * 4,
                      st, blob + EncBlob::AA_EDGE6, g->aa_geom, g->aa_dst, w.q, int64_t(g->E_aa), w.logits, w.v, heads);
    }
    else if (edge_x6())
      TS_LAUNCH_TAG("k_edge_kv[aa]", true, k_edge_kv<true>, tile_grid((int64_t(g->E_aa) + 15) / 16, threads_edge(), EdgeL6::SIZE * 4), threads_edge(),
                    EdgeL6::SIZE * 4, st, blob + EncBlob::AA_EDGE6, g->aa_geom, g->aa_dst, w.q, int64_t(g->E_aa), w.logits, w.v, heads);
    else
      TS_LAUNCH_TAG("k_edge_kv[aa]", true, k_edge_kv<false>, tile_grid((int64_t(g->E_aa) + 15) / 16, threads_edge(), EdgeL::SIZE * 4), threads_edge(),
                    EdgeL::SIZE * 4, st, blob + EncBlob::AA_EDGE, g->aa_geom, g->aa_dst, w.q, int64_t(g->E_aa), w.logits, w.v, heads);
  }
  TS_REQUIRE(!state_bf16(), "bf16 state storage needs the fused edge attention (default)");
  return attention_tail(im, g->aa_segptr, w.logits, w.v, w.cn, w.center, R, w.agg, w.x1, w.xn2, aa_out, st, heads, drop);
}

// one pass of the latent SDE + GRU recurrence; iteration idx consumes history step t = H-1-idx (ENC:128-182)
static int run_recurrence(const trajsde_batch* b, const trajsde_graph* g, const float* blob, const float* step_tab, const float* h0,
                          int noise_step0, NoiseArg na, EncWs& w, const float* aa_out, float* kept, float* diff_pick,
                          float* latent_ys, hipStream_t st) {
  const int N = b->N, Nt = g->Nt, H = b->H;
  const int64_t rtiles = (int64_t(Nt) + 15) / 16;
  // cooperative persistent kernel: all H iterations in one launch, weights resident in the register file
  static const bool legacy = []() { const char* e = getenv("TRAJSDE_RECUR_LEGACY"); return e && atoi(e) != 0; }();
  static const int force_tw = []() { const char* e = getenv("TRAJSDE_RECUR_TW"); return e ? atoi(e) : 0; }();      // experiments
  const int tiles_per_wg = force_tw > 0 ? force_tw : int((rtiles + 255) / 256);
  TS_REQUIRE(!state_bf16() || (!legacy && tiles_per_wg <= COOP_TMAX && H <= 32),
             "bf16 state storage needs the cooperative recurrence kernel (at most 16384 extended rows)");
  if (!legacy && tiles_per_wg <= COOP_TMAX && H <= 32) {
    StepTab tab;
    for (int i = 0; i < H; ++i) {
      tab.dt[i] = step_tab[8 * i + 1]; tab.sq[i] = step_tab[8 * i + 2]; tab.sn[i] = step_tab[8 * i + 3]; tab.cs[i] = step_tab[8 * i + 4];
    }
    const int grid = int((rtiles + tiles_per_wg - 1) / tiles_per_wg);
    const int lds = coop_lds_floats(tiles_per_wg) * 4;
#define TS_COOP(TW) TS_LAUNCH_TAG("k_enc_recur_coop", false, (k_enc_recur_coop<TW, false>), grid, 256, lds, st, blob + EncBlob::SDE, blob + EncBlob::GRU, blob + EncBlob::COOP6, h0, aa_out, Nt, N, H, b->TT, tiles_per_wg, tab, \
              noise_step0, na, g->nus_mask, b->padding_mask, g->orig, g->eos_idx, g->pick_slot, kept, diff_pick, latent_ys, state_bf16() ? 1 : 0, RecurTape{})
    switch (tiles_per_wg) {
      case 1: TS_COOP(1); break;
      case 2: TS_COOP(2); break;
      case 3: TS_COOP(3); break;
      default: TS_COOP(4); break;
    }
#undef TS_COOP
    return TRAJSDE_OK;
  }
  for (int idx = 0; idx < H; ++idx) {
    const int t = H - 1 - idx;
    const float* e = step_tab + 8 * idx;
    TS_LAUNCH(k_enc_sde_step, tile_grid(rtiles, threads_recur(), EncSdeL::SIZE * 4), threads_recur(), EncSdeL::SIZE * 4, st, blob + EncBlob::SDE,
              idx == 0 ? nullptr : w.hA, h0, Nt, e[1], e[2], e[3], e[4], idx, noise_step0, na, g->nus_mask, g->eos_idx,
              g->pick_slot, w.hB, diff_pick);
    TS_LAUNCH(k_enc_gru_step, tile_grid(rtiles, threads_recur(), EncGruL::SIZE * 4), threads_recur(), EncGruL::SIZE * 4, st, blob + EncBlob::GRU, w.hB,
              aa_out + int64_t(t) * Nt * 64, Nt, N, t, b->TT, idx, b->padding_mask, g->orig, g->eos_idx, w.hA, kept,
              latent_ys ? latent_ys + int64_t(idx) * N * 64 : nullptr);
  }
  return TRAJSDE_OK;
}

// ALEncoder (ENC:198-200, 732-797): lat [N,64] -> local_embed [N,64]
static int run_al_encoder(const trajsde_batch* b, const trajsde_graph* g, const float* blob, EncWs& w, const float* lat,
                          float* local_embed, hipStream_t st, int heads = 8, const DropArg& drop = no_drop()) {
  const int N = b->N;
  TS_LAUNCH(k_node_proj<1>, tile_grid((int64_t(N) + 15) / 16, 512, NodeProjL<1>::SIZE * 4), 512, NodeProjL<1>::SIZE * 4, st,
            blob + EncBlob::AL_Q, lat, int64_t(N), w.al_xn, w.al_q, nullptr, nullptr);
  const NodeImgs im{blob + EncBlob::AL_UPD, blob + EncBlob::AL_FFN, blob + EncBlob::AL_UPD6, blob + EncBlob::AL_FFN6};
  if (attn_fused()) {
    SegMerge mg = no_merge();
    if (int rc = fused_edge_attention("k_edge_kv[al]", false, blob + EncBlob::AL_EDGE6F, g->la_geom, g->la_dst, w.al_q, count_of(g, 3, g->E_la), g->la_segptr,
                                      int64_t(N), w.al_rec, w.al_agg, heads, st, drop, nullptr, nullptr, &mg))
      return rc;
    return update_ffn(im, w.al_agg, w.al_xn, lat, N, w.al_x1, w.al_xn2, local_embed, st, drop, false, mg);
  }
  if (g->E_la > 0) {
    if (edge_x6() && edge_pair())
      TS_LAUNCH_TAG("k_edge_kv[al]", false, k_edge_kv2<512>, tile_grid((int64_t(g->E_la) + 31) / 32, 512, EdgeL6::SIZE * 4), 512, EdgeL6::SIZE * 4, st,
                    blob + EncBlob::AL_EDGE6, g->la_geom, g->la_dst, w.al_q, int64_t(g->E_la), w.al_logits, w.al_v, heads);
    else if (edge_x6())
      TS_LAUNCH_TAG("k_edge_kv[al]", false, k_edge_kv<true>, tile_grid((int64_t(g->E_la) + 15) / 16, threads_edge(), EdgeL6::SIZE * 4), threads_edge(),
                    EdgeL6::SIZE * 4, st, blob + EncBlob::AL_EDGE6, g->la_geom, g->la_dst, w.al_q, int64_t(g->E_la), w.al_logits, w.al_v, heads);
    else
      TS_LAUNCH_TAG("k_edge_kv[al]", false, k_edge_kv<false>, tile_grid((int64_t(g->E_la) + 15) / 16, threads_edge(), EdgeL::SIZE * 4), threads_edge(),
                    EdgeL::SIZE * 4, st, blob + EncBlob::AL_EDGE, g->la_geom, g->la_dst, w.al_q, int64_t(g->E_la), w.al_logits, w.al_v, heads);
  }
  return attention_tail(im, g->la_segptr, w.al_logits, w.al_v, w.al_xn, lat, N, w.al_agg, w.al_x1, w.al_xn2, local_embed, st, heads, drop);
}

int trajsde_encoder_forward(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob,
                            const float* step_tab /*HOST [H,8]*/, const trajsde_noise* noise, void* ws, int64_t ws_bytes,
                            float* local_embed, float* diff_pick, float* aa_out_user, float* latent_ys, const trajsde_dropout* dropout,
                            void* stream_) {
  TS_REQUIRE(b && g && rot && blob && step_tab && ws && local_embed && diff_pick, "encoder_forward: null pointer");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "encoder_forward: dropout p must be in [0, 1)");
  const DropArg drop_aa = dropout ? make_drop(dropout->p, dropout->seed, 0) : no_drop();     // block ids of dropout.hpp
  const DropArg drop_al = dropout ? make_drop(dropout->p, dropout->seed, 1) : no_drop();
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_forward: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(g->exact || (attn_fused() && g->counts), "encoder_forward: a graph from trajsde_graph_prepare_async needs the fused edge attention");
  TS_REQUIRE(b->A > 0 && g->Nt == b->N + b->A, "encoder_forward: graph was prepared without the fake-agent rows");
  EncWs w(b, g, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_forward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  float* aa_out = aa_out_user ? aa_out_user : w.aa_out;
  NoiseArg na{0, nullptr, nullptr};
  if (noise) { na.seed = noise->seed; na.z = noise->z; na.row_ids = noise->row_ids; na.seed_dev = noise->seed_dev; }
  if (int rc = run_aa_encoder(b, g, rot, blob, w, aa_out, st, 8, drop_aa)) return rc;
  TS_HIP(hipMemsetAsync(diff_pick, 0, size_t(2) * b->A * 64 * sizeof(float), st));
  if (t_fork_stream != nullptr) {
    // trajsde_encoder_fork_stream: whatever the caller enqueues on that stream from now on starts when the recurrence does
    // (one-shot; an event record + a stream wait, both capturable)
    if (t_fork_event == nullptr) TS_HIP(hipEventCreateWithFlags(&t_fork_event, hipEventDisableTiming));
    TS_HIP(hipEventRecord(t_fork_event, st));
    TS_HIP(hipStreamWaitEvent(t_fork_stream, t_fork_event, 0));
    t_fork_stream = nullptr;
  }
  if (int rc = run_recurrence(b, g, blob, step_tab, blob + EncBlob::HIDDEN, 0, na, w, aa_out, w.lat, diff_pick, latent_ys, st)) return rc;
  return run_al_encoder(b, g, blob, w, w.lat, local_embed, st, 8, drop_al);
}

// LocalEncoderSDESepPara2.forward_ood (ENC:204-370): the batch/graph carry no fake agents (A = 0); `n_samples`
// stochastic recurrences from a zero state, per-actor std of the kept latent, AL encoder on the mean.
int64_t trajsde_encoder_ood_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int n_samples) {
  if (!b || !g || n_samples < 1) return -1;
  EncWs w(b, g, nullptr, 0);
  return w.total + align_up(int64_t(n_samples) * b->N * 64 * 4, 256) + align_up(int64_t(b->N) * 64 * 4, 256) + 1024;
}

int trajsde_encoder_forward_ood(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob,
                                const float* step_tab /*HOST [H,8]*/, const trajsde_noise* noise, int n_samples, void* ws,
                                int64_t ws_bytes, float* local_embed, float* stds, void* stream_) {
  TS_REQUIRE(b && g && rot && blob && step_tab && ws && local_embed && stds, "encoder_forward_ood: null pointer");
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_forward_ood: graph not compacted");
  TS_REQUIRE(g->exact, "encoder_forward_ood: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(b->A == 0 && g->Nt == b->N, "encoder_forward_ood: prepare the graph with A = 0 (no fake agents)");
  TS_REQUIRE(n_samples >= 1, "encoder_forward_ood: n_samples < 1");
  if (ws_bytes < trajsde_encoder_ood_ws_bytes(b, g, n_samples)) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_forward_ood: workspace too small");
  EncWs w(b, g, ws, ws_bytes);
  Carver extra(static_cast<char*>(ws) + align_up(w.total, 256), ws_bytes - align_up(w.total, 256));
  float* samples = extra.take<float>(int64_t(n_samples) * b->N * 64);
  float* mean = extra.take<float>(int64_t(b->N) * 64);
  float* zero = extra.take<float>(64);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  NoiseArg na{0, nullptr, nullptr};
  if (noise) { na.seed = noise->seed; na.z = noise->z; na.row_ids = noise->row_ids; na.seed_dev = noise->seed_dev; }
  if (int rc = run_aa_encoder(b, g, rot, blob, w, w.aa_out, st)) return rc;
  TS_HIP(hipMemsetAsync(zero, 0, 64 * sizeof(float), st));                         // ENC:257 prev_hidden = zeros
  for (int j = 0; j < n_samples; ++j)
    if (int rc = run_recurrence(b, g, blob, step_tab, zero, j * b->H, na, w, w.aa_out, samples + int64_t(j) * b->N * 64, nullptr,
                                nullptr, st))
      return rc;
  TS_LAUNCH(k_ood_stats, cdiv(b->N, 4), 256, 0, st, samples, n_samples, b->N, mean, stds);
  return run_al_encoder(b, g, blob, w, mean, local_embed, st);
}

// vanilla LocalEncoder.forward (GENC:52-93): AAEncoder on the 21 snapshots (no fake agents: graph prepared with A = 0),
// TemporalEncoder per actor, ALEncoder.
int64_t trajsde_encoder_grid_ws_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EncWs w(b, g, nullptr, 0);
  return w.total + 9 * align_up(int64_t(b->N) * 22 * 64 * 4, 256) + align_up(int64_t(b->N) * 64 * 4, 256) + 1024;
}

int trajsde_encoder_grid_forward(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob, int num_heads,
                                 int num_temporal_layers, void* ws, int64_t ws_bytes, float* local_embed, void* stream_) {
  return trajsde_encoder_grid_forward_train(b, g, rot, blob, num_heads, num_temporal_layers, ws, ws_bytes, local_embed, nullptr, stream_);
}

// the same forward with the train-mode dropout of the reference's modules (GENC:52-93 under model.train()): attention weights,
// proj_drop and the two FFN sites of the AA / AL blocks (blocks 0 / 1 of dropout.hpp) and, per TemporalEncoder layer l (block 16 + l),
// nn.MultiheadAttention's dropout on the softmax output, dropout1, the FFN's dropout and dropout2 (GENC:262-283)
int trajsde_encoder_grid_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob, int num_heads,
                                       int num_temporal_layers, void* ws, int64_t ws_bytes, float* local_embed,
                                       const trajsde_dropout* dropout, void* stream_) {
  TS_REQUIRE(b && g && rot && blob && ws && local_embed, "encoder_grid_forward: null pointer");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "encoder_grid_forward: dropout p must be in [0, 1)");
  const bool dropping = dropout && dropout->p > 0.f;
  auto drop_of = [&](int block) { return dropping ? make_drop(dropout->p, dropout->seed, block) : no_drop(); };
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_grid_forward: graph not compacted");
  TS_REQUIRE(g->exact, "encoder_grid_forward: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(b->A == 0 && g->Nt == b->N, "encoder_grid_forward: prepare the graph with A = 0 (no fake agents)");
  TS_REQUIRE(b->H == 21, "encoder_grid_forward: the temporal kernels are built for historical_steps = 21");
  TS_REQUIRE(num_heads == 8 || num_heads == 4, "encoder_grid_forward: num_heads must be 8 or 4");
  TS_REQUIRE(num_temporal_layers >= 1, "encoder_grid_forward: no temporal layers");
  if (ws_bytes < trajsde_encoder_grid_ws_bytes(b, g)) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_grid_forward: workspace too small");
  EncWs w(b, g, ws, ws_bytes);
  Carver extra(static_cast<char*>(ws) + align_up(w.total, 256), ws_bytes - align_up(w.total, 256));
  const int N = b->N;
  const int64_t R = int64_t(N) * 22, rtiles = (R + 15) / 16;
  float *xa = extra.take<float>(R * 64), *xb = extra.take<float>(R * 64), *xn = extra.take<float>(R * 64), *q = extra.take<float>(R * 64),
        *k = extra.take<float>(R * 64), *v = extra.take<float>(R * 64), *o = extra.take<float>(R * 64), *x1 = extra.take<float>(R * 64),
        *xn2 = extra.take<float>(R * 64), *tout = extra.take<float>(int64_t(N) * 64);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (int rc = run_aa_encoder(b, g, rot, blob, w, w.aa_out, st, num_heads, drop_of(0))) return rc;
  TS_LAUNCH(k_tr_prep, cdiv(R * 64, 256), 256, 0, st, w.aa_out, b->padding_mask, blob + EncGridBlob::TOK, N, b->TT, xa);
  float* x = xa;
  float* nx = xb;
  for (int l = 0; l < num_temporal_layers; ++l) {
    const float* lb = blob + EncGridBlob::layer(l);
    TS_LAUNCH(k_node_proj<3>, tile_grid(rtiles, 512, NodeProjL<3>::SIZE * 4), 512, NodeProjL<3>::SIZE * 4, st, lb + TrLayerL::QKV, x, R, xn, q, k,
              v);
    const DropArg dl = drop_of(DROP_TEMPORAL_BLOCK0 + l);
    if (int rc = launch_tr_attention(num_heads, q, k, v, N, o, dl, st)) return rc;
    TS_LAUNCH(k_tr_outproj, tile_grid(rtiles, 512, TrOutL::SIZE * 4), 512, TrOutL::SIZE * 4, st, lb + TrLayerL::OUT, o, x, R, x1, xn2, dl);
    TS_LAUNCH(k_ffn, tile_grid(rtiles, 512, FfnL::SIZE * 4), 512, FfnL::SIZE * 4, st, lb + TrLayerL::FFN, x1, xn2, R, nx, dl, 0);
    float* t = x; x = nx; nx = t;
  }
  TS_LAUNCH(k_tr_final, tile_grid((int64_t(N) + 15) / 16, 256, 0), 256, 0, st, blob + EncGridBlob::norm(num_temporal_layers), x, N, tout);
  return run_al_encoder(b, g, blob, w, tout, local_embed, st, num_heads, drop_of(1));
}

int64_t trajsde_aggregator_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_modes) {
  if (!b || !g) return -1;
  (void)num_modes;
  AggWs w(b, g, nullptr, 0);
  return w.total;
}

int trajsde_aggregator_forward(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers, int num_modes,
                               const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed, void* stream_) {
  return trajsde_aggregator_forward_heads(b, g, blob, num_layers, num_modes, 8, local_embed, ws, ws_bytes, global_embed, nullptr, stream_);
}

// relative-pose embedding of every global edge (AGG:42-51), once for the layers: depends on the graph stage alone
// The rel rows of an INFERENCE forward are stored as split-precision operand pieces (fp16 hi | lo, the same 256 bytes) when their
// readers are the split-image attention (gattn_h3.hip): 8 heads, fp32 state, no dropout, the default kernel forms.
static bool rel_split_possible() {
#if TSDE_SPLIT_H3
  return rel_split_enabled() && !state_bf16() && edge_x6() && rel_embed_fused() && global_fused_env() && !gattn_mm_enabled();
#else
  return false;
#endif
}
static int aggregator_rel_embed(const trajsde_batch* b, const trajsde_graph* g, const float* blob, AggWs& w, hipStream_t st, bool split) {
  const int64_t E = g->E_g, etiles = (E + 15) / 16;
  if (E > 0) {
#if TSDE_SPLIT_H3
    if (edge_x6() && rel_embed_fused())                     // two tiles per wave on the centred image (attn.hip k_edge_embed2)
      TS_LAUNCH_TAG("k_edge_embed<true>", false, k_edge_embed2, tile_grid((E + 31) / 32, 1024, edge_embed2_lds(1024)), 1024, edge_embed2_lds(1024), st,
                    blob + AggBlob::REL6G, g->g_geom, count_of(g, 2, E), w.rel, split ? 2 : (state_bf16() ? 1 : 0));
    else
#endif
    if (edge_x6())
      TS_LAUNCH(k_edge_embed<true>, tile_grid(etiles, threads_edge(), EdgeL6::EMB_SIZE * 4), threads_edge(), EdgeL6::EMB_SIZE * 4, st,
                blob + AggBlob::REL6, g->g_geom, count_of(g, 2, E), w.rel, state_bf16() ? 1 : 0);
    else
      TS_LAUNCH(k_edge_embed<false>, tile_grid(etiles, threads_edge(), EdgeL::EMB_SIZE * 4), threads_edge(), EdgeL::EMB_SIZE * 4, st,
                blob + AggBlob::REL, g->g_geom, count_of(g, 2, E), w.rel, state_bf16() ? 1 : 0);
  }
  return TRAJSDE_OK;
}

int trajsde_encoder_fork_stream(void* side_stream) {
  t_fork_stream = static_cast<hipStream_t>(side_stream);
  return TRAJSDE_OK;
}

int trajsde_aggregator_prepare(const trajsde_batch* b, const trajsde_graph* g, const float* blob, void* ws, int64_t ws_bytes, void* stream_) {
  TS_REQUIRE(b && g && blob && ws, "aggregator_prepare: null pointer");
  TS_REQUIRE(g->g_src && g->g_segptr, "aggregator_prepare: graph not compacted (call trajsde_graph_compact)");
  AggWs w(b, g, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "aggregator_prepare: workspace too small");
  // (the layers' caller may turn out to need fp32 rows -- 4 heads, dropout --: aggregator_forward_impl then forms them again)
  return aggregator_rel_embed(b, g, blob, w, static_cast<hipStream_t>(stream_), rel_split_possible());
}

static int aggregator_forward_impl(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers, int num_modes,
                                   int num_heads, const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed,
                                   const trajsde_dropout* dropout, void* stream_, bool rel_ready);

int trajsde_aggregator_forward_heads(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers, int num_modes,
                                     int num_heads, const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed,
                                     const trajsde_dropout* dropout, void* stream_) {
  return aggregator_forward_impl(b, g, blob, num_layers, num_modes, num_heads, local_embed, ws, ws_bytes, global_embed, dropout, stream_, false);
}

int trajsde_aggregator_forward_prepared(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers, int num_modes,
                                        int num_heads, const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed,
                                        const trajsde_dropout* dropout, void* stream_) {
  return aggregator_forward_impl(b, g, blob, num_layers, num_modes, num_heads, local_embed, ws, ws_bytes, global_embed, dropout, stream_, true);
}

static int aggregator_forward_impl(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers, int num_modes,
                                   int num_heads, const float* local_embed, void* ws, int64_t ws_bytes, float* global_embed,
                                   const trajsde_dropout* dropout, void* stream_, bool rel_ready) {
  TS_REQUIRE(b && g && blob && local_embed && ws && global_embed, "aggregator_forward: null pointer");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "aggregator_forward: dropout p must be in [0, 1)");
  TS_REQUIRE(num_heads == 8 || num_heads == 4, "aggregator_forward: num_heads must be 8 or 4");
  TS_REQUIRE(g->g_src && g->g_segptr, "aggregator_forward: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(num_layers >= 0 && num_modes > 0, "aggregator_forward: bad layer/mode count");
  AggWs w(b, g, ws, ws_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "aggregator_forward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int64_t N = b->N, E = g->E_g, ntiles = (N + 15) / 16, etiles = (E + 15) / 16;
  const bool split = rel_split_possible() && num_heads == 8 && (!dropout || dropout->p == 0.f) && E > 0;
  if (!rel_ready || split != rel_split_possible())
    if (int rc = aggregator_rel_embed(b, g, blob, w, st, split)) return rc;
  const float* x = local_embed;
  float* bufs[2] = {w.xa, w.xb};
  for (int i = 0; i < num_layers; ++i) {
    const float* lb = blob + AggBlob::layer(i);
    if (split)                                               // k_node / v_node rows leave split like the rel rows (gattn_h3.hip)
      TS_LAUNCH((k_node_proj<3, true>), tile_grid(ntiles, 512, NodeProjL<3>::SIZE * 4), 512, NodeProjL<3>::SIZE * 4, st, lb + AggLayerL::QKV, x, N,
                w.xn, w.q, w.kn, w.vn);
    else
      TS_LAUNCH(k_node_proj<3>, tile_grid(ntiles, 512, NodeProjL<3>::SIZE * 4), 512, NodeProjL<3>::SIZE * 4, st, lb + AggLayerL::QKV, x, N,
                w.xn, w.q, w.kn, w.vn);
    float* out = bufs[i & 1];
    const NodeImgs im{lb + AggLayerL::UPD, lb + AggLayerL::FFN, lb + AggLayerL::UPD6, lb + AggLayerL::FFN6};
    const bool fused = global_fused_env() || num_heads != 8;           // the unfused edge kernel exists for 8 heads only
    const DropArg drop = dropout ? make_drop(dropout->p, dropout->seed, 2 + i) : no_drop();   // block ids of dropout.hpp
    TS_REQUIRE(fused || drop.p == 0.f, "aggregator_forward: dropout needs the fused global attention (unset TRAJSDE_GLOBAL_UNFUSED)");
    TS_REQUIRE(fused || !state_bf16(), "aggregator_forward: bf16 state storage needs the fused global attention");
    TS_REQUIRE(fused || g->exact, "aggregator_forward: a graph from trajsde_graph_prepare_async needs the fused global attention");
    if (fused) {
      // one wave per target: logits, softmax and aggregation in one pass over the rel rows (no per-edge GEMM)
      TS_REQUIRE(N < (1 << 23), "aggregator_forward: node rows are addressed with 32-bit byte offsets (N < 2^23)");
      // (the scene-cached form needs the batch vector and room for the scene pointers; without them the gathering form serves alone)
      if (split && rel_split_scene_cache() && b->batch != nullptr && b->A > 0 && b->A + 2 <= E * 8 + 8) {      // (A = scenes: 0 without fake agents)
        int32_t* scene_ptr = reinterpret_cast<int32_t*>(w.logits);        // (the unfused path's per-edge logits: idle in this form)
        if (i == 0)
          if (int rc = launch_scene_ptr(b->batch, b->N, b->A, g->g_src, g->g_dst, g->g_segptr, E, scene_ptr, st)) return rc;
        if (int rc = launch_global_attn_sc(lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q, w.kn, w.vn, N, b->A, b->batch, scene_ptr, w.agg, st))
          return rc;
      } else if (split) {
        if (int rc = launch_global_attn_h3(lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q, w.kn, w.vn, N, w.agg, st)) return rc;
      } else if (num_heads == 8 && !state_bf16() && drop.p == 0.f && gattn_mm_enabled()) {
        if (int rc = launch_global_attn_mm(lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q, w.kn, w.vn, N, w.agg, st)) return rc;
      } else if (num_heads == 8 && !state_bf16() && gattn_f32mm_enabled() && E > 0) {
        if (int rc = launch_global_attn_mf(lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q, w.kn, w.vn, N, w.agg, nullptr, drop, st)) return rc;
      } else {
        TS_GLOBAL_ATTN(num_heads, state_bf16(), drop, xcd_grid(cdiv(N, 4)), 256, 0, st, lb + AggLayerL::ATTN, g->g_segptr, g->g_src, w.rel, w.q, w.kn, w.vn, N, w.agg, static_cast<float*>(nullptr));
      }
      if (int rc = update_ffn(im, w.agg, w.xn, x, N, w.x1, w.xn2, out, st, drop)) return rc;
      x = out;
      continue;
    }
    if (E > 0) {
      if (edge_x6())
        TS_LAUNCH(k_global_edge<true>, tile_grid(etiles, 512, GEdgeL6::SIZE * 4), 512, GEdgeL6::SIZE * 4, st,
                  lb + AggLayerL::EDGE6, w.rel, g->g_src, g->g_dst, w.q, w.kn, w.vn, E, w.logits, w.v);
      else
        TS_LAUNCH(k_global_edge<false>, tile_grid(etiles, 512, GEdgeL::SIZE * 4), 512, GEdgeL::SIZE * 4, st,
                  lb + AggLayerL::EDGE, w.rel, g->g_src, g->g_dst, w.q, w.kn, w.vn, E, w.logits, w.v);
    }
    if (int rc = attention_tail(im, g->g_segptr, w.logits, w.v, w.xn, x, N, w.agg, w.x1, w.xn2, out, st)) return rc;
    x = out;
  }
  const int lds = (128 + MAT64 + 64) * 4;
  dim3 grid(tile_grid(ntiles, 512, lds), num_modes);
  TS_LAUNCH(k_mode_proj, grid, 512, lds, st, blob + AggBlob::norm(num_layers), blob + AggBlob::proj(num_layers, 0), x, N, global_embed);
  return TRAJSDE_OK;
}

}  // extern "C"
