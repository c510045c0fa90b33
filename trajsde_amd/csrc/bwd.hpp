// bwd.hpp -- pieces of the backward pass shared across translation units: the weight-gradient reduction kernels
// (defined in decoder_bwd.hip) and the node-level backward kernels (node_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "dropout.hpp"

namespace tsde {

constexpr int WGRAD_CHUNK = 512;           // rows one k_wgrad workgroup reduces
constexpr int WGRAD_MAX_JOBS = 16;         // weight-gradient problems over the same rows that share one launch
// partial slots of a workspace: the largest single problem (+32: run_headwise_outer slices) plus room for batching small ones
inline int64_t wgrad_max_parts(int64_t rows, int64_t groups) { return (rows + WGRAD_CHUNK - 1) / WGRAD_CHUNK + groups + 33 + 2048; }

struct WgradJob {              // W[o*ldw + col0 + i] = sum_r delta[r*ldd + o] * a[r*lda + i];  bias[o] = sum_r delta[r*ldd + o] (or null)
  const float *delta, *a;
  float *W, *bias;
  int ldd, lda, ldw, col0, time_cols;
  // computed operand: when in2 is set, row r of `a` is ReLU(LN(Linear(2,64)(geom[r][pair], geom[r][pair+1]))) evaluated
  // on the fly from the 16-byte geometry record `a + 4r` with the closed-form block `in2` (layouts.hpp In2L) and `beta`
  // -- the activation rows of the two embedding branches are never written to or read from HBM
  const float *in2, *beta;
  int pair;
};
struct WgradJobs {
  WgradJob j[WGRAD_MAX_JOBS];
  int n;
};
__global__ void k_wgrad(WgradJobs jobs, int64_t R, int64_t rows_per_group, int chunk, int chunks_per_group, int P, float* part, float* cs);
__global__ void k_reduce_partials(WgradJobs jobs, const float* part, const float* cs, int P, int chunks_per_group, const float* step_tab);
constexpr int COLSUM_MAX_JOBS = 8;         // vectors cut from the same slab of per-wave partials that share one launch
struct ColsumJob {
  const float* src;            // first column of the vector inside the slab
  float* dst;
  int n, dst_stride;
};
struct ColsumJobs {
  ColsumJob j[COLSUM_MAX_JOBS];
  int n;
};
__global__ void k_colsum(ColsumJobs jobs, int64_t rows, int stride);

// ---- pieces of the SDE decoder backward reused by the MLP decoder backward (decoder_bwd.hip)
struct InitV { enum : int { DGAM = 0, DBET = 64, SIZE = 128 }; };               // per-wave vector slots of k_dec_init_bwd
__global__ void k_l2_wta(const float* loc, const float* y, const uint8_t* mask, int N, int K, int T, int32_t* best, float* minsum,
                         int32_t* cnt, int KP);
__global__ void k_l2_finalize(const float* minsum, const int32_t* cnt, int N, float* scal);
__global__ void k_init_sel(const float* img, const float* local, const float* global, const int32_t* best, int N, float* y0, float* gsel);
__global__ void k_dec_init_bwd(const float* img, const float* local, const float* gsel, const float* DY0, const int32_t* best, int N,
                               float* DA, float* d_local, float* d_global, float* vpart);

struct WgradCtx {
  hipStream_t st;
  float *part, *cs;            // scratch for `cap` partials of 4096 / 64 floats
  const float* step_tab;       // only read when time_cols is set
  int64_t cap;                 // = wgrad_max_parts(...) the workspace was carved with
};
// several weight-gradient problems over the SAME rows (R, rows_per_group) in one pair of launches (grid.y = problem)
struct WgradBatch {
  const WgradCtx& c;
  int64_t R, rows_per_group;
  WgradJobs jobs;
  const char* tag;             // name of the launch in the profile table (the edge-embedding batch carries its own: its roofline)
  WgradBatch(const WgradCtx& ctx, int64_t R_, int64_t rpg, const char* tag_ = "k_wgrad") : c(ctx), R(R_), rows_per_group(rpg), tag(tag_) { jobs.n = 0; }
  int add(const float* delta, int ldd, const float* a, int lda, float* W, int ldw, int col0, float* bias, int time_cols);
  int add_in2(const float* delta, int ldd, const float* geom, int pair, const float* in2, const float* beta, float* W, int ldw, float* bias);
  int flush();
  int flush_edge();           // the edge embedding's three problems (decoder_bwd.hip k_wgrad6_edge); falls back to flush()
};
// W[o*ldw + col0 + i] = sum_r delta[r*ldd + o] * a[r*lda + i]  (o, i < 64);  bias[o] = sum_r delta[r*ldd + o] (or null)
int run_wgrad(const WgradCtx& c, const float* delta, int ldd, const float* a, int lda, int64_t R, int64_t rows_per_group, float* W,
              int ldw, int col0, float* bias, int time_cols);
// ---- deferred sums.  A backward entry point of the SDE path produces ~35 weight-gradient batches and ~33 slabs of per-wave vector
// partials; summing each with its own launch (two dozen microseconds apiece, a handful of workgroups) cost 0.7 ms of a 12.4 ms step.
// With a DeferredSums object alive, WgradBatch::flush / run_headwise_outer leave their partials where they are (a bump allocator over
// the context's partial buffer) and ColsumBatch::flush leaves the slab where it is (vpart_slab hands every producer its own), and the
// sums of the whole entry point run in a few wide launches at finish() -- or earlier, whenever one of the two areas is full.  Same
// fixed summation order as the immediate kernels; nothing reads a gradient buffer before the entry point returns.
struct ReduceJob {             // one 64 x 64 (+ bias / time columns) block: its partials start at slot `base`
  float *W, *bias;
  int64_t base;
  int P, cpg, ldw, col0, time_cols;
};
constexpr int REDUCE_MAX_JOBS = 64;
struct ReduceJobs {
  ReduceJob j[REDUCE_MAX_JOBS];
  int n;
};
struct ColsumQJob {
  const float* src;
  float* dst;
  int64_t rows;
  int n, stride, dst_stride;
};
constexpr int COLSUMQ_MAX_JOBS = 96;
struct ColsumQJobs {
  ColsumQJob j[COLSUMQ_MAX_JOBS];
  int n;
};
struct ReduceQueue {
  hipStream_t st;
  float *part, *cs;
  const float* step_tab;
  int64_t cap, used;
  std::vector<ReduceJob> jobs;
  int64_t take(int64_t slots, int* rc);      // first slot of a fresh run of `slots` partials (drains first when they do not fit)
  int drain();
};
struct ColsumQueue {
  hipStream_t st;
  float* arena;
  int64_t cap, used;
  std::vector<ColsumQJob> jobs;
  float* take(int64_t floats);               // null: does not fit even after a drain (the caller falls back to its shared slab)
  int drain();
};
ReduceQueue*& active_reduce_queue();
ColsumQueue*& active_colsum_queue();
struct DeferredSums {
  ReduceQueue rq;
  ColsumQueue cq;
  DeferredSums(hipStream_t st, float* part, float* cs, int64_t cap, const float* step_tab, float* arena, int64_t arena_floats);
  ~DeferredSums();                           // deactivates the queues (error paths); launches nothing
  int finish();                              // the remaining sums; the queues stay active and empty
  DeferredSums(const DeferredSums&) = delete;
};
// the slab a producer of per-wave vector partials writes ([rows][stride] floats) and the ColsumBatch built on it reads: the
// caller's shared slab, or -- with deferred sums active -- a slab of its own that stays intact until the sums have run
float* vpart_slab(float* shared_slab, int64_t rows, int stride);
constexpr int64_t VPART_ARENA_SLABS = 6;     // the arena of a workspace, in units of VPART_FLOATS

int run_colsum(hipStream_t st, const float* src, int64_t rows, int stride, int n, float* dst, int dst_stride = 1);
int run_colsum_tall(hipStream_t st, const float* src, int64_t rows, int stride, int n, float* dst, float* scratch /* 256 x n floats */);
// several vectors of the same slab (rows x stride floats of per-wave partials) in one launch
struct ColsumBatch {
  hipStream_t st;
  int64_t rows;
  int stride;
  ColsumJobs jobs;
  ColsumBatch(hipStream_t st_, int64_t rows_, int stride_) : st(st_), rows(rows_), stride(stride_) { jobs.n = 0; }
  int add(const float* src, int n, float* dst, int dst_stride = 1);
  int flush();
};
// W[d][c] = sum_i X[i][d] * Y[i][head(d)][c]   (X [N,64], Y [N,heads,64]) through wc.part
int run_headwise_outer(const WgradCtx& wc, const float* X, const float* Y, int64_t N, float* W, int heads = 8);

// ---- node-level backward blocks (node_bwd.hip)
__global__ void k_ffn_bwd_a(const float* img, const float* dout, const float* xn2, int64_t R, float* H, float* DH, float* DOUT2, DropArg drop);
__global__ void k_ffn_bwd_b(const float* img, const float* DH, const float* dout, const float* x1, int64_t R, float* dx1, float* vpart);
__global__ void k_upd_bwd(const float* img, const float* dx1, const float* agg, const float* xn, int64_t R, float* UPD, float* DGP,
                          float* DS, float* DAGG, float* DXN, float* DX1M, DropArg drop);
template <int NQ>
__global__ void k_node_proj_bwd(const float* img, const float* x, const float* dres, const float* dxn_part, const float* dp0,
                                const float* dp1, const float* dp2, int64_t R, float* dx_out, float* xn_out, float* vpart);
template <int BR>
__global__ void k_edge_embed_bwd_branch(const float* img, const float* geom, const float* DSP, int64_t E, float* vpart);
__global__ void k_lin_t_acc(const float* wt, const float* d, int64_t R, float* out, int accumulate);
__global__ void k_lin_t_sum(const float* wt, int64_t wt_stride, const float* d, int64_t d_stride, int K, int64_t R, float* out);

constexpr int64_t VPART_FLOATS = int64_t(2048) * 4 * 320;     // per-wave vector partials of the widest kernel at the largest grid

struct NodeBlockTape { const float *agg, *xn, *x1, *xn2; };                      // forward activations [R,64]
struct NodeBlockScratch { float *H, *DH, *dx1, *UPD, *DGP, *DS, *vpart; };        // [R,256] x2, [R,64] x4, VPART_FLOATS
struct NodeBlockGrads {                                                           // parameter-shaped gradient buffers
  float *w_ih, *b_ih, *w_hh, *b_hh, *w_self, *b_self, *w_out, *b_out, *n2g, *n2b, *w1, *b1, *w2, *b2;
};
// backward of  x1 = x + out_proj(gated update(agg, xn)),  out = x1 + mlp(norm2(x1))  given dout [R,64]:
// writes dagg, dxn (the block's contribution to d xn) and sc.dx1 (= d x1, also the residual gradient of x)
int node_block_backward(const float* img /*NodeBlockBwdL*/, const NodeBlockTape& tp, const float* dout, int64_t R,
                        const NodeBlockScratch& sc, const WgradCtx& wc, const NodeBlockGrads& gr, float* dagg, float* dxn,
                        hipStream_t st, const DropArg& drop, WgradBatch* defer = nullptr);

// the FFN half alone (TemporalEncoderLayer: linear1 / linear2 / norm2); uses gr.w1, b1, w2, b2, n2g, n2b only
int ffn_block_backward(const float* img_a /*FfnBwdAL*/, const float* img_b /*FfnBwdBL*/, const float* xn2, const float* x1,
                       const float* dout, int64_t R, const NodeBlockScratch& sc, const WgradCtx& wc, const NodeBlockGrads& gr,
                       hipStream_t st, const DropArg& drop);

struct EdgeEmbedScratch { float *S, *DEP, *DSP, *vpart; };                       // [E,64] x3
struct EdgeEmbedGrads {
  float *a_w0, *a_b0, *a_g, *a_e, *b_w0, *b_b0, *b_g, *b_e, *wa3, *ba3, *wb3, *bb3, *ag0, *ae0, *w2, *b2, *ag3, *ae3;
};
// what the attention backward over stored embedding rows leaves per edge (run_edge_attn_bwd), for the embedding backward to
// build d emb from: the edges' targets, the targets' q / dagg rows, the (alpha d, d logit) scalars and lin_k^T | lin_v^T
struct EdgeAttnGrad {
  const int32_t* dst;
  const float *q, *dagg, *EA, *ED;
  const float* wkvt;           // [2][64][64]: lin_k.weight^T | lin_v.weight^T (EdgeKvBwdL::WKT, WVT)
  int heads;
};
// `demb`: d emb rows [E,64], or (ag != null) built inside the kernel from the attention scalars
int edge_embed_backward(const float* img /*EdgeBwdL*/, const float* geom, const float* demb, int64_t E, const EdgeEmbedScratch& sc,
                        const WgradCtx& wc, const EdgeEmbedGrads& gr, hipStream_t st, const EdgeAttnGrad* ag = nullptr);

// ---- wave-per-target attention backward (aggregator_bwd.hip), shared by the global interactor and the AA / AL encoders
struct DrelArgs {
  const float* EA[4];          // per layer: [E][HEADS] alpha d_e (the weight the values were summed with)
  const float* ED[4];          // per layer: [E][HEADS] d logit / sqrt(dh)
  const float* UZ[4];          // per layer: [N][HEADS][2][64] (U_h, Z_h) of every target, or (from_rows) unused
  const float *img, *q, *dagg; // from_rows: GAttnL image and the targets' q / dagg rows
};
// DREL[e] (+)= sum over the `nl` <= 4 layers and heads of ED U_h + EA Z_h
int run_gattn_drel(hipStream_t st, int heads, int nl, bool from_rows, const DrelArgs& da, const int32_t* segptr, int64_t N, float* DREL,
                   int accumulate);
// attention over stored edge rows `emb` [E,64] (k = lin_k(emb), v = lin_v(emb); img = GAttnL): given dagg [R,64] and the forward's
// (max, 1/sum) statistics, writes DQ [R,64], RL / SS [R,heads,64] (lin_k / lin_v weight gradients as headwise outer products
// with q / dagg), DAGGM [R,64] (column sum = lin_v bias gradient) and the per-edge scalars EA / ED [E,heads]
int run_edge_attn_bwd(hipStream_t st, int heads, const float* img, const int32_t* segptr, const float* emb, const float* q, const float* agg,
                      const float* dagg, const float* stats, int64_t R, float* DQ, float* RL, float* SS, float* DAGGM, float* EA, float* ED,
                      const DropArg& drop, const WgradCtx* wc = nullptr, float* wk = nullptr, float* wv = nullptr, float* bv = nullptr,
                      bool* weights_done = nullptr);
// (wc / wk / wv / weights_done: the kernel may accumulate the lin_k / lin_v weight gradients itself -- *weights_done then tells the caller to
//  skip its two run_headwise_outer calls and the column sum of DAGGM (lin_v.bias); RL / SS / DAGGM are not written in that case)

// ---- TemporalEncoder backward kernels (grid_bwd.hip)
__global__ void k_tr_final_bwd(const float* norm, const float* x, const float* dtout, int N, float* DX, float* vpart);
template <int HEADS, bool DROP>
__global__ void k_tr_attention_bwd(const float* q, const float* k, const float* v, const float* dO, int N, float* dq, float* dk, float* dv,
                                   DropArg drop);
__global__ void k_drop_rows(const float* src, int64_t R, float* dst, DropArg drop, int kind);        // dst = src * factors of (row, feature)
__global__ void k_tr_prep_bwd(const float* DX0, const uint8_t* pad, int N, int TT, float* DAA);
__global__ void k_tr_tok_grad(const float* DX0, const uint8_t* pad, int N, int TT, float* dpad, float* dcls, float* dpos);

// ordered parameter names of a stage (the dry run of its pack recipe, pack.hip)
std::vector<std::string> stage_param_names(int stage, int num_layers, int num_modes);

}  // namespace tsde
