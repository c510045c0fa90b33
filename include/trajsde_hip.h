/* trajsde_hip.h -- C-ABI of libtrajsde_hip.so: the MI355X (gfx950) kernels of TrajSDE's forward hot path.
 *
 * Boundary: the reference is pure Python; its stage modules
 *   models/encoders/enc_hivt_nusargo_sde_sep2.py   LocalEncoderSDESepPara2.forward   :66-202
 *   models/aggregators/agg_hivt.py                 GlobalInteractor.forward          :38-58
 *   models/decoders/dec_hivt_nusargo_sde.py        SDEDecoder.forward                :77-105
 *   models/model_base_mix_sde.py                   rotate_mat / y rotation           :75-85
 * are what a maintainer re-points (YAML file_path) at trajsde_amd/models/...; those Python classes bind the
 * entry points below with ctypes.  Every entry point
 *   - takes plain device pointers + sizes + a hipStream_t (passed as void*), no torch types;
 *   - BORROWS its pointers for the duration of the enqueued work, allocates nothing persistent and writes
 *     only into caller-provided buffers (workspace sizes come from the *_ws_bytes queries);
 *   - returns 0 on success or a negative trajsde_status; trajsde_last_error() gives the message
 *     (thread-local).  No exception crosses this boundary.
 * All floating point is fp32; bool tensors are 1 byte per element; index tensors are int64 as in the
 * reference's batch and are narrowed to int32 on device.
 */
#ifndef TRAJSDE_HIP_H
#define TRAJSDE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TRAJSDE_OK = 0,
  TRAJSDE_ERR_INVALID = -1,   /* bad argument (null pointer, unsupported size)           */
  TRAJSDE_ERR_HIP = -2,       /* a HIP runtime call or kernel launch failed              */
  TRAJSDE_ERR_WORKSPACE = -3, /* caller workspace too small                              */
  TRAJSDE_ERR_UNSUPPORTED = -4
} trajsde_status;

const char* trajsde_last_error(void);
int trajsde_abi_version(void);
/* matrix products per fp32 product of the split-precision kernels this library was built with:
 * 3 = fp16x3 (default), 6 = bf16x6 (build with TRAJSDE_SPLIT=bf16x6); see csrc/tile.hpp */
int trajsde_split_products(void);
/* Storage of the [rows][64] activations that stay INSIDE a stage between its kernels -- the relative-pose embeddings of the
 * global interactor [E_g,64], aa_out [H,Nt,64] entering the recurrence, the decoder's initial states y0 [K*N,64], and the
 * state of trajsde_sde_step: mode 0 = fp32 (default), 1 = bf16 ("bf16 hidden state" of the stress configuration: rows are
 * rounded to nearest-even when stored and widened exactly when loaded; all arithmetic stays fp32 in registers; stage
 * boundaries -- local_embed, global_embed, loc, pi -- stay fp32).  Process-wide; returns the previous mode.  Inference only:
 * the backward entry points keep their tape in fp32 and refuse to run in mode 1.  In mode 1 the buffers trajsde_sde_step
 * reads and writes hold bf16 elements. */
int trajsde_state_storage(int mode);
/* fp16 range guard of the split-precision products (csrc/range.hpp).  Kernels that feed a data-dependent, unnormalised
 * tensor (SDE states, rows entering the recurrence / the decoder, attention aggregates, FFN hidden units) or a weight to an
 * fp16x3 product set a sticky per-device bit when a magnitude reaches 65504, where the fp16 pieces would saturate.  This call
 * synchronises `stream`, reads (and with reset != 0 clears) the bits and returns TRAJSDE_OK or TRAJSDE_ERR_UNSUPPORTED with
 * the affected sites in trajsde_last_error(); *sites_out (optional) receives the bit mask.  Call it where the host
 * synchronises anyway (end of an evaluation epoch, when a loss value is read); always 0 in a bf16x6 build. */
int trajsde_range_status(int reset, uint32_t* sites_out, void* stream);

/* ---- weights ----------------------------------------------------------------------------------
 * Parameters stay owned by the Python modules (nn.Parameter).  Each stage hands the library an array
 * of device pointers in the order given by trajsde_param_name(stage, i) (state_dict key relative to
 * the stage) and gets back one packed fp32 blob laid out as the kernels' LDS images
 * (MFMA-fragment order for matrices).  Re-pack whenever a parameter changes. */
typedef enum {
  TRAJSDE_STAGE_ENCODER = 0,
  TRAJSDE_STAGE_AGGREGATOR = 1,
  TRAJSDE_STAGE_DECODER = 2,
  /* backward images: the parameter list of a *_BWD stage names the parameters that receive a gradient, in the
   * order of the gradient buffers the stage's backward entry point takes */
  TRAJSDE_STAGE_DECODER_BWD = 3,    /* trajsde_decoder_l2_backward (pi / scale heads get no gradient from that loss) */
  TRAJSDE_STAGE_AGGREGATOR_BWD = 4, /* trajsde_aggregator_backward */
  TRAJSDE_STAGE_ENCODER_BWD = 5,    /* trajsde_encoder_backward */
  /* vanilla HiVT variant (configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml): `num_layers` carries the number of
   * temporal layers for ENCODER_GRID and the number of future steps for DECODER_MLP */
  TRAJSDE_STAGE_ENCODER_GRID = 6,
  TRAJSDE_STAGE_DECODER_MLP = 7,
  TRAJSDE_STAGE_DECODER_MLP_BWD = 8, /* trajsde_mlp_decoder_l2_backward; num_layers = future steps */
  TRAJSDE_STAGE_ENCODER_GRID_BWD = 9, /* trajsde_encoder_grid_backward; num_layers = temporal layers */
  TRAJSDE_STAGE_DECODER_NLL_BWD = 10  /* trajsde_decoder_nll_backward: the DECODER_BWD table followed by the scale head (ABI 8) */
} trajsde_stage;

int trajsde_param_count(int stage, int num_layers /*aggregator*/, int num_modes);
const char* trajsde_param_name(int stage, int index, int num_layers, int num_modes);
int64_t trajsde_blob_floats(int stage, int num_layers, int num_modes);
int trajsde_pack_weights(int stage, int num_layers, int num_modes, const float* const* params, int n_params,
                         float* blob, int64_t blob_floats, void* stream);

/* The weight images of SEVERAL stages in three launches over one job table in device memory (ABI 10).  A training step re-packs
 * every image after the optimizer step (the reference's modules read their nn.Parameters directly, MODEL:104-115: packing is this
 * library's cost, so it is kept to one call): zero the blobs | first-pass jobs | second-pass jobs.  The merged table is built when
 * the arguments are first seen, written to `table_host` (PINNED host memory) and copied to `table_dev` on `stream`; while the item
 * list, the parameter addresses, the blob addresses and `table_dev` stay the same, later calls only launch.  `fresh` != 0 forces
 * the upload (the caller re-allocated or wrote `table_dev`).  Both tables hold trajsde_pack_many_table_bytes(items, n) bytes, are
 * owned by the caller and must not be written by it while calls with these arguments continue.  Blobs: distinct, 16-byte aligned.
 * The images are bit-identical to those of n calls of trajsde_pack_weights. */
typedef struct {
  int32_t stage, num_layers, num_modes, n_params;
  const float* const* params; /* n_params device pointers in the order of trajsde_param_name(stage, ...) */
  float* blob;                /* >= trajsde_blob_floats(stage, ...) floats */
  int64_t blob_floats;
} trajsde_pack_item;
int64_t trajsde_pack_many_table_bytes(const trajsde_pack_item* items, int n);
int trajsde_pack_weights_many(const trajsde_pack_item* items, int n, void* table_host, void* table_dev, int64_t table_bytes,
                              int fresh, void* stream);

/* ---- the end of a training step (ABI 10): two element-wise launches in place of ~14 ------------------------------------------
 * trajsde_grad_gather_add: dst[i] += src[index[i]] * (scale[0] * mult) for up to 8 (dst, src, index) items in ONE launch -- the
 * stage backward calls hand back their gradients as one buffer per stage in the order of trajsde_param_name; the training loop keeps
 * one flat gradient tensor in the order of model.parameters() (what `loss.backward()` fills through autograd in the reference,
 * MODEL:104-115).  `scale`: a device scalar (the incoming gradient of the loss) or NULL for 1.
 * trajsde_adamw_step: torch.optim.AdamW's update (MODEL:204-207; amsgrad and maximize off) over n elements in one launch, operation by
 * operation as torch/optim/adam.py performs it; the caller forms the scalars as torch does: decay = 1 - lr * weight_decay,
 * w1 = 1 - beta1, w2 = 1 - beta2, neg_step = -(lr / (1 - beta1^step)), step counted from 1, and
 *   divide = 1, bias2 = sqrt(1 - beta2^step):      the multi-tensor form (`foreach=True`: what AdamW(model.parameters()) runs on a GPU),
 *   divide = 0, bias2 = 1 / sqrt(1 - beta2^step):  the single-tensor form (`foreach=False`; reciprocal formed in double, then rounded)
 * -- the two differ in that one operation (a true division against a product with the reciprocal) and each is reproduced bit for bit. */
typedef struct {
  float* dst;
  const float* src;
  const int64_t* index;
  int64_t n;
  float mult;
} trajsde_gather_item;
int trajsde_grad_gather_add(const trajsde_gather_item* items, int n_items, const float* scale, void* stream);
int trajsde_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float decay, float w1,
                       float beta2, float w2, float bias2, int divide, float eps, float neg_step, void* stream);

/* ---- the batch ------------------------------------------------------------ */
typedef struct {
  int32_t N;            /* actors                                   */
  int32_t A;            /* target agents (= scenes)                 */
  int32_t E;            /* columns of edge_index                    */
  int32_t L;            /* lane segments                            */
  int32_t E_al;         /* columns of lane_actor_index              */
  int32_t H;            /* historical steps (21)                    */
  int32_t TT;           /* time slots in positions/padding_mask (21+F_data) */
  int32_t lane_pts;     /* points per lane segment (10)             */
  const float* x;                  /* [N,H,2]            */
  const float* positions;          /* [N,TT,2]           */
  const uint8_t* padding_mask;     /* [N,TT]  1 = missing */
  const uint8_t* bos_mask;         /* [N,H]              */
  const float* rotate_angles;      /* [N]                */
  const int64_t* edge_index;       /* [2,E]  row0 src, row1 dst */
  const int64_t* agent_index;      /* [A]    ascending   */
  const int64_t* batch;            /* [N]    scene id    */
  const int64_t* source;           /* [A]    0 = nuScenes, 1 = Argoverse */
  const float* lane_positions;     /* [L,lane_pts,2]     */
  const float* lane_paddings;      /* [L,lane_pts] 1 = pad */
  const int64_t* lane_actor_index; /* [2,E_al] row0 lane, row1 actor */
  const float* lane_actor_vectors; /* [E_al,2]           */
} trajsde_batch;

/* ---- noise: injected standard normals or in-kernel Philox4x32-7 (curand-free) ------------------
 * Philox counter = (row id, step, stream, column/4), key = seed; row ids are GLOBAL ids when row_ids
 * is given, else the local row index (host twin: trajsde_amd/philox.py). */
typedef struct {
  uint64_t seed;
  const float* z;            /* if non-null: injected N(0,1) values, layout documented per call */
  const int32_t* row_ids;    /* optional global row ids for the Philox counter                  */
  const uint64_t* seed_dev;  /* ABI 7, optional: the key is read from this DEVICE word when the kernel runs (`seed` is then
                              * ignored), so a captured hipGraph of the forward draws fresh noise on every replay           */
} trajsde_noise;

/* ---- train-mode dropout of the attention blocks (AAEncoder ENC:521-533,592,611; ALEncoder ENC:711-723,771,794;
 *      GlobalInteractorLayer AGG:78-90,116,132): nn.Dropout(p) on the attention weights, on out_proj's output and on the two
 *      activations of the FFN.  Masks are cut from the same counter-based Philox stream as the SDE noise (csrc/dropout.hpp;
 *      host twin trajsde_amd/philox.py), keyed by (seed, block, site, element) -- so the forward, the forward recomputation
 *      inside the backward entry points and the backward kernels all see the same mask when they are handed the same struct.
 *      Pass null (or p = 0) for eval mode. */
typedef struct {
  float p;          /* drop probability in [0, 1); kept elements are scaled by 1 / (1 - p) */
  uint64_t seed;
} trajsde_dropout;

/* ---- MODEL:75-85  rotate_mat[n] = [[cos,-sin],[sin,cos]],  y_rot = y @ R_n ---------------------- */
int trajsde_rotate(const float* rotate_angles, int32_t N, const float* y /*[N,F,2] or null*/, int32_t F,
                   float* rotate_mat /*[N,2,2]*/, float* y_rot /*[N,F,2] or null*/, void* stream);

/* ---- graph preparation (ENC:88-118 fake agents + per-step subgraph + radius drop; AGG:41; ENC:198;
 *      UTIL:83-92).  Builds, on device: CSR-by-target of edge_index (counting sort, rows in canonical
 *      ascending-source order), the virtual rows of the fake agents, the compacted (t, edge) list of the
 *      21 agent-agent snapshots with pre-rotated geometry (count pass, prefix sum, fill pass), the
 *      compacted global and lane-actor edge lists, and their segment pointers.
 *      Synchronises the stream once to return the three edge counts. */
typedef struct {
  int32_t Nt;        /* N + A                                   */
  int32_t E_ext;     /* = E (the fake agents' in-edges are virtual) */
  int32_t E_aa;      /* surviving (t,edge) pairs over 21 steps  */
  int32_t E_g;       /* global-interactor edges                 */
  int32_t E_la;      /* lane-actor edges within the radius      */
  /* device pointers into the caller's graph workspace (valid while it lives) */
  const int32_t* orig;        /* [Nt] original actor of each extended row         */
  const uint8_t* nus_mask;    /* [Nt]                                              */
  const int32_t* eos_idx;     /* [Nt] recurrence iteration whose state is kept    */
  const int32_t* pick_slot;   /* [Nt] row of diff_pick [2A,64] or -1              */
  const float* x_fake;        /* [A,H,2]                                           */
  const float* aa_geom;       /* [E_aa,4] (x_j R_i, edge_attr R_i)                 */
  const int32_t* aa_dst;      /* [E_aa] snapshot node id t*Nt+i                    */
  const int32_t* aa_segptr;   /* [H*Nt+1]                                          */
  const float* g_geom;        /* [E_g,4] (rel_pos R_i, cos dtheta, sin dtheta)     */
  const int32_t* g_src;       /* [E_g]                                             */
  const int32_t* g_dst;       /* [E_g]                                             */
  const int32_t* g_segptr;    /* [N+1]                                             */
  const float* la_geom;       /* [E_la,4] (lane_feat R_i, lane_actor_vector R_i)   */
  const int32_t* la_dst;      /* [E_la]                                            */
  const int32_t* la_segptr;   /* [N+1]                                             */
  /* ABI 2: the sender of every compacted record (the kernels do not read these: the geometry above already carries what a
   * sender contributes; they make the index work of ENC:107-118 / ENC:198 checkable edge for edge) */
  const int32_t* aa_src;      /* [E_aa] sending actor (a real actor, < N); null unless trajsde_export_senders(1) */
  const int32_t* la_lane;     /* [E_la] lane segment; null unless trajsde_export_senders(1)                      */
  /* ABI 5: the three list lengths on the DEVICE (counts[1] = E_aa, [2] = E_g, [3] = E_la).  With exact = 0
   * (trajsde_graph_prepare_async) the E_* fields above are UPPER BOUNDS that size buffers and grids, the kernels of the
   * inference forward read the true lengths here, and no host synchronisation happens anywhere in the forward. */
  const int32_t* counts;
  int32_t exact;              /* 1: E_aa / E_g / E_la are the true lengths (trajsde_graph_prepare) */
} trajsde_graph;

/* Process-wide switch: have trajsde_graph_compact also write the sender of every compacted record (aa_src / la_lane above).
 * Off by default -- the kernels never read them, they exist so that tests can compare the index work edge for edge.
 * Returns the previous setting. */
int trajsde_export_senders(int on);

/* Two phases, because the sizes of the compacted lists are data dependent:
 *   prepare : sort/CSR, fake-agent rows, validity+radius flags, prefix sums; fills the counts and the
 *             per-node arrays of `out`; synchronises the stream to return the counts.
 *   compact : writes the compacted edge lists / geometry / segment pointers into `edges_ws`
 *             (trajsde_graph_edges_ws_bytes(out) bytes) and fills the remaining pointers of `out`.
 * Both workspaces must stay alive while `out` is in use. */
int64_t trajsde_graph_ws_bytes(const trajsde_batch* b);
int trajsde_graph_prepare(const trajsde_batch* b, const float* rotate_mat, float local_radius,
                          const trajsde_noise* fake_noise /* z: [A,H,2] */, void* ws, int64_t ws_bytes,
                          trajsde_graph* out, void* stream);
/* trajsde_graph_prepare without the stream synchronisation: the counts stay on the device (out->counts), the E_* fields
 * are upper bounds (E_aa <= 2 * H * E, E_g <= E, E_la <= E_al) and out->exact = 0.  Accepted by
 * trajsde_graph_compact and by the inference entry points trajsde_encoder_forward / trajsde_aggregator_forward_heads in
 * their default (fused) kernel forms; every other entry point (backward, training, OOD, vanilla variant) needs an exact
 * graph and says so. */
int trajsde_graph_prepare_async(const trajsde_batch* b, const float* rotate_mat, float radius, const trajsde_noise* fake_noise,
                                void* ws, int64_t ws_bytes, trajsde_graph* out, void* stream);
/* The radius test of the snapshot lists is  dx*dx + dy*dy < T  with T = the smallest float whose correctly rounded square root
 * reaches `radius`: exactly the survivors of the reference's  sqrt(dx*dx + dy*dy) < radius  (utils/util.py:88).  Exported so
 * that the equivalence can be checked on the host (tests/test_host_logic.py). */
float trajsde_radius2_threshold(float radius);
/* 1 when the kernel forms this process selected (environment switches of csrc/stages.hip) accept a graph from
 * trajsde_graph_prepare_async, i.e. the default build and environment */
int trajsde_sync_free_supported(void);
int64_t trajsde_graph_edges_ws_bytes(const trajsde_batch* b, const trajsde_graph* g);
int trajsde_graph_compact(const trajsde_batch* b, const float* rotate_mat, void* ws, int64_t ws_bytes,
                          void* edges_ws, int64_t edges_ws_bytes, trajsde_graph* out, void* stream);

/* ---- encoder stage: AAEncoder (ENC:538-614) -> SDE+GRU recurrence (ENC:128-182, SDEINT:477-485,
 *      ODEU:136-152) -> ALEncoder (ENC:732-797). */
int64_t trajsde_encoder_ws_bytes(const trajsde_batch* b, const trajsde_graph* g);
int trajsde_encoder_forward(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat,
                            const float* blob, const float* enc_step_table /*HOST memory, [H,8] (t0,dt,sqrt_h,sin,cos,..)*/,
                            const trajsde_noise* noise /* z: [H,Nt,64] */, void* ws, int64_t ws_bytes,
                            float* local_embed /*[N,64]*/, float* diff_pick /*[2A,64]*/,
                            float* aa_out /*[H,Nt,64] or null*/, float* latent_ys /*[H,N,64] or null*/,
                            const trajsde_dropout* dropout /* null = eval mode */, void* stream);

/* ---- LocalEncoderSDESepPara2.forward_ood (ENC:204-370): graph prepared with b->A = 0 (no fake agents);
 *      n_samples (reference: 10) stochastic recurrences from a zero state; stds[n] = std over samples of
 *      the kept latent state (unbiased), averaged over the 64 channels; local_embed from the sample mean.
 *      Noise layout when injected: z [n_samples*H, N, 64]; Philox step index = sample*H + iteration. */
int64_t trajsde_encoder_ood_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int n_samples);
int trajsde_encoder_forward_ood(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat,
                                const float* blob, const float* enc_step_table /*HOST memory, [H,8]*/,
                                const trajsde_noise* noise, int n_samples, void* ws, int64_t ws_bytes,
                                float* local_embed /*[N,64]*/, float* stds /*[N]*/, void* stream);

/* ---- aggregator stage: GlobalInteractor (AGG:38-58, 92-135) ------------------------------------ */
int64_t trajsde_aggregator_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_modes);
int trajsde_aggregator_forward(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers,
                               int num_modes, const float* local_embed /*[N,64]*/, void* ws, int64_t ws_bytes,
                               float* global_embed /*[K,N,64]*/, void* stream);

/* the same with the head count of the attention spelled out (8: the SDE config, 4: the vanilla HiVT config) */
int trajsde_aggregator_forward_heads(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers,
                                     int num_modes, int num_heads, const float* local_embed, void* ws, int64_t ws_bytes,
                                     float* global_embed, const trajsde_dropout* dropout /* null = eval mode */, void* stream);

/* ABI 8.  The relative-pose embedding of the global edges (AGG:42-51: `rel_embed` of edge_attr, once for all layers) reads
 * the graph stage's output only, not the encoder's: trajsde_aggregator_prepare runs it alone into the stage's workspace
 * (same `ws` / `ws_bytes` as the forward call that follows), trajsde_aggregator_forward_prepared is
 * trajsde_aggregator_forward_heads without it.  A host that enqueues `prepare` on a second stream lets that chip-filling
 * kernel share the GPU with the encoder's serial recurrence (AGG:38-58 split at AGG:51 | AGG:52; trajsde_encoder_fork_stream
 * below says where the second stream starts); the caller orders the two streams (an event after `prepare`, waited for before
 * `forward_prepared`). */
int trajsde_aggregator_prepare(const trajsde_batch* b, const trajsde_graph* g, const float* blob, void* ws, int64_t ws_bytes,
                               void* stream);
/* Where on the encoder's timeline such side work should start: the NEXT trajsde_encoder_forward call of this host thread records
 * an event on its stream right before it launches the recurrence (ENC:128-200: the 21 SDE + GRU iterations, one persistent kernel
 * on 2/3 of the CUs at one wave per SIMD) and makes `side_stream` wait for it -- so work enqueued on `side_stream` after that call
 * returns runs beside the recurrence, not beside the chip-filling attention kernels before it.  One-shot; null disarms. */
int trajsde_encoder_fork_stream(void* side_stream);
int trajsde_aggregator_forward_prepared(const trajsde_batch* b, const trajsde_graph* g, const float* blob, int num_layers,
                                        int num_modes, int num_heads, const float* local_embed, void* ws, int64_t ws_bytes,
                                        float* global_embed, const trajsde_dropout* dropout /* null = eval mode */, void* stream);

/* ---- decoder stage: SDEDecoder (DEC:77-105) with the stock Euler-Maruyama solve over the float32
 *      schedule tables of SURVEY.md App. D (trajsde_amd/schedule.py). */
int64_t trajsde_decoder_ws_bytes(int32_t N, int num_modes);
int trajsde_decoder_forward(int32_t N, int num_modes, int future_steps, const float* blob,
                            const float* local_embed /*[N,64]*/, const float* global_embed /*[K,N,64]*/,
                            const float* step_table /*[n_euler,8]*/, int n_euler,
                            const float* out_table /*[T,4] (steps_done,w0,w1,0)*/, float min_scale,
                            const trajsde_noise* noise /* z: [n_euler,K*N,64] */, void* ws, int64_t ws_bytes,
                            float* loc /*[K,N,T,4]*/, float* pi /*[N,K]*/, void* stream);

/* ---- vanilla HiVT variant (configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml): LocalEncoder.forward
 *      (enc_hivt_nusargo_grid.py:52-93: AAEncoder, TemporalEncoder :225-292, ALEncoder; graph prepared with A = 0) and
 *      MLPDecoder.forward (dec_hivt_nusargo_grid.py:47-63).  Blobs: TRAJSDE_STAGE_ENCODER_GRID / _DECODER_MLP. */
int64_t trajsde_encoder_grid_ws_bytes(const trajsde_batch* b, const trajsde_graph* g);
int trajsde_encoder_grid_forward(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob,
                                 int num_heads, int num_temporal_layers, void* ws, int64_t ws_bytes,
                                 float* local_embed /*[N,64]*/, void* stream);
/* The same forward under model.train(): the reference's dropout modules are active (configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml
 * ships dropout 0.1) -- the four sites of the AA and AL attention blocks and, per TemporalEncoder layer, nn.MultiheadAttention's dropout on
 * the softmax output, dropout1, the feed-forward dropout and dropout2 (enc_hivt_nusargo_grid.py:256-283).  Masks come from the Philox
 * stream keyed by `dropout` (null or p = 0: identical to trajsde_encoder_grid_forward). */
int trajsde_encoder_grid_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob,
                                       int num_heads, int num_temporal_layers, void* ws, int64_t ws_bytes,
                                       float* local_embed /*[N,64]*/, const trajsde_dropout* dropout, void* stream);
int64_t trajsde_mlp_decoder_ws_bytes(int32_t N, int num_modes);
int trajsde_mlp_decoder_forward(int32_t N, int num_modes, int future_steps, const float* blob,
                                const float* local_embed /*[N,64]*/, const float* global_embed /*[K,N,64]*/, float min_scale,
                                void* ws, int64_t ws_bytes, float* loc /*[K,N,T,4]*/, float* pi /*[N,K]*/, void* stream);

/* backward of MLPDecoder under the winner-takes-all L2 loss (the vanilla configuration's only loss): like
 * trajsde_decoder_l2_backward without noise; grads follow trajsde_param_name(TRAJSDE_STAGE_DECODER_MLP_BWD, i). */
int64_t trajsde_mlp_decoder_backward_ws_bytes(int32_t N);
int trajsde_mlp_decoder_l2_backward(int32_t N, int num_modes, int future_steps, const float* blob_bwd, const float* local_embed,
                                    const float* global_embed, const float* loc /*[K,N,T,4]*/, const float* y /*[N,T,2]*/,
                                    const uint8_t* reg_mask /*[N,T]*/, void* ws, int64_t ws_bytes, float* loss,
                                    int32_t* best_mode, float* const* grads, int n_grads, float* d_local, float* d_global,
                                    void* stream);

/* backward of the vanilla LocalEncoder: dL/d local_embed -> one gradient per parameter of
 * trajsde_param_name(TRAJSDE_STAGE_ENCODER_GRID_BWD, i) (buffers pre-zeroed by the caller); forward recomputed inside. */
int64_t trajsde_encoder_grid_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_temporal_layers);
int trajsde_encoder_grid_backward(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob_fwd,
                                  const float* blob_bwd, int num_heads, int num_temporal_layers, const float* d_local /*[N,64]*/,
                                  void* ws, int64_t ws_bytes, float* const* grads, int n_grads, void* stream);

/* ... of a forward that ran with dropout (trajsde_encoder_grid_forward_train with the same `dropout`) */
int trajsde_encoder_grid_backward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob_fwd,
                                        const float* blob_bwd, int num_heads, int num_temporal_layers, const float* d_local /*[N,64]*/,
                                        void* ws, int64_t ws_bytes, float* const* grads, int n_grads,
                                        const trajsde_dropout* dropout, void* stream);

/* ---- winner-takes-all L2 regression loss (losses/L2.py:10-27) + backward of the decoder stage: gradients of
 *      loss = mean over valid (actor, step) of |y - loc[best mode]| w.r.t. the decoder parameters and the stage
 *      inputs.  `loc` is the forward output, `noise` the same noise (seed or z) the forward used; the winning
 *      paths are replayed and differentiated through every Euler-Maruyama step.  `grads[i]` is a buffer shaped
 *      like parameter trajsde_param_name(TRAJSDE_STAGE_DECODER_BWD, i) and is overwritten; `blob_bwd` is that
 *      stage's packed image, `blob_fwd` the TRAJSDE_STAGE_DECODER one.  The pi and scale heads get no gradient
 *      from this loss.  best_mode may be null. */
int64_t trajsde_decoder_backward_ws_bytes(int32_t N, int num_modes, int future_steps, int n_euler);
int trajsde_decoder_l2_backward(int32_t N, int num_modes, int future_steps, const float* blob_fwd, const float* blob_bwd,
                                const float* local_embed /*[N,64]*/, const float* global_embed /*[K,N,64]*/,
                                const float* step_table /*[n_euler,8]*/, int n_euler, const float* out_table /*[T,4]*/,
                                const trajsde_noise* noise, const float* loc /*[K,N,T,4]*/, const float* y /*[N,T,2]*/,
                                const uint8_t* reg_mask /*[N,T]*/, void* ws, int64_t ws_bytes, float* loss /*[1] device*/,
                                int32_t* best_mode /*[N] device or null*/, float* const* grads, int n_grads,
                                float* d_local /*[N,64]*/, float* d_global /*[K,N,64]*/, void* stream);

/* The same stage under the Laplace negative log-likelihood (losses/laplace_nll_loss.py:18-47; ABI 8): the winner is still the
 * mode with the smallest masked L2, the loss is mean over valid steps and both coordinates of log(2 s) + |y - l| / s with
 * s = max(scale, eps) of that mode -- so the scale head (ELU + 1 + min_scale, DEC:97-98) receives a gradient too.  `grads`
 * follow trajsde_param_name(TRAJSDE_STAGE_DECODER_NLL_BWD, i): the DECODER_BWD table, then scale.0 / .1 / .3 weight and bias;
 * `blob_bwd` is that stage's image. */
int64_t trajsde_decoder_nll_backward_ws_bytes(int32_t N, int num_modes, int future_steps, int n_euler);
int trajsde_decoder_nll_backward(int32_t N, int num_modes, int future_steps, const float* blob_fwd, const float* blob_bwd,
                                 const float* local_embed, const float* global_embed, const float* step_table, int n_euler,
                                 const float* out_table, const trajsde_noise* noise, const float* loc, const float* y,
                                 const uint8_t* reg_mask, float eps, float min_scale, void* ws, int64_t ws_bytes, float* loss,
                                 int32_t* best_mode, float* const* grads, int n_grads, float* d_local, float* d_global, void* stream);

/* ---- backward of the aggregator stage (AGG:38-58, 92-135): dL/d global_embed -> dL/d local_embed (overwritten;
 *      the caller adds the decoder's own d local_embed) and one gradient per aggregator parameter, grads[i] shaped
 *      like parameter trajsde_param_name(TRAJSDE_STAGE_AGGREGATOR_BWD, i) and overwritten (pre-zero them: with no
 *      global edges the rel_embed entries are left untouched).  The forward is recomputed internally in fp32. */
int64_t trajsde_aggregator_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_layers, int num_modes);
int trajsde_aggregator_backward(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, const float* blob_bwd,
                                int num_layers, int num_modes, const float* local_embed /*[N,64]*/,
                                const float* d_global /*[K,N,64]*/, void* ws, int64_t ws_bytes, float* const* grads, int n_grads,
                                float* d_local /*[N,64]*/, void* stream);

/* ---- backward of the encoder stage (ENC:66-202) + the DiffBCE loss on its diffusion outputs (losses/diff_BCE.py):
 *      d_local = dL/d local_embed (decoder + aggregator contributions); diff_weight = weight of the DiffBCE term in
 *      the total loss (0 skips it); *diff_loss receives diff_weight * DiffBCE.  `noise` must be the forward's.
 *      grads[i] is shaped like parameter trajsde_param_name(TRAJSDE_STAGE_ENCODER_BWD, i) (pre-zeroed by the caller,
 *      overwritten where the batch reaches the parameter).  d_latent [N,64] / d_aa_out [H,Nt,64] are optional
 *      outputs (null to skip) of the gradients at the two internal stage boundaries.  The step table is passed
 *      twice: host copy (scalars for the launches) and device copy (time-feature column reductions). */
/*      Workspaces (ABI 6): the stage's memory is the forward TAPE (trajsde_encoder_tape_bytes: what the training forward writes
 *      and the backward reads) plus the backward's own SCRATCH (trajsde_encoder_backward_scratch_bytes).  Either hand both in one
 *      buffer `ws` of trajsde_encoder_backward_ws_bytes (tape first; scratch = null), or hand the tape as `ws` and a separate
 *      `scratch` buffer: a training step then holds only the tape between its forward and this call, and can allocate the scratch
 *      after the decoder's and aggregator's workspaces are released (peak memory of a step: max, not sum). */
int64_t trajsde_encoder_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g);
int64_t trajsde_encoder_tape_bytes(const trajsde_batch* b, const trajsde_graph* g);
int64_t trajsde_encoder_backward_scratch_bytes(const trajsde_batch* b, const trajsde_graph* g);
int trajsde_encoder_backward(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob_fwd,
                             const float* blob_bwd, const float* enc_step_table /*HOST [H,8]*/,
                             const float* enc_step_table_dev /*device [H,8]*/, const trajsde_noise* noise,
                             const float* d_local /*[N,64]*/, float diff_weight, void* ws, int64_t ws_bytes,
                             float* diff_loss /*[1] device*/, float* const* grads, int n_grads, float* d_latent, float* d_aa_out,
                             const trajsde_dropout* dropout /* the forward's, or null */,
                             int tape_valid /* 1: `ws` still holds the tape trajsde_encoder_forward_train left in it */,
                             void* scratch /* or null: scratch follows the tape inside `ws` */, int64_t scratch_bytes, void* stream);

/* Training forward of the encoder stage: the same function as trajsde_encoder_forward (same weights, noise, dropout), run
 * through the kernels that KEEP every activation the backward needs, in `ws` (at least trajsde_encoder_tape_bytes).
 * Hand the SAME `ws` to trajsde_encoder_backward with tape_valid = 1 and the backward skips its own recomputation of the
 * forward (one forward per training step instead of two).  Outputs as trajsde_encoder_forward. */
int trajsde_encoder_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rotate_mat, const float* blob_fwd,
                                  const float* enc_step_table /*HOST [H,8]*/, const trajsde_noise* noise, void* ws,
                                  int64_t ws_bytes, float* local_embed /*[N,64]*/, float* diff_pick /*[2A,64]*/,
                                  const trajsde_dropout* dropout, void* stream);

int trajsde_aggregator_backward_heads(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, const float* blob_bwd,
                                      int num_layers, int num_modes, int num_heads, const float* local_embed,
                                      const float* d_global, void* ws, int64_t ws_bytes, float* const* grads, int n_grads,
                                      float* d_local, const trajsde_dropout* dropout /* the forward's, or null */,
                                      int tape_valid /* 1: `ws` holds the tape of trajsde_aggregator_forward_train */, void* stream);

/* Training forward of the aggregator stage (see trajsde_encoder_forward_train): keeps its per-layer activations in `ws`
 * (trajsde_aggregator_backward_ws_bytes) for trajsde_aggregator_backward_heads(..., tape_valid = 1). */
int trajsde_aggregator_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* blob_fwd, int num_layers,
                                     int num_modes, int num_heads, const float* local_embed, void* ws, int64_t ws_bytes,
                                     float* global_embed /*[K,N,64]*/, const trajsde_dropout* dropout, void* stream);

/* ---- step-granular decoder SDE step (state round-trips HBM every step): the 512 B/path-step variant
 *      of SURVEY.md 8(d), kept for the HBM-roofline measurement the north star asks for. */
int trajsde_sde_step(int32_t rows, const float* blob, const float* y_in, float* y_out,
                     const float* step_entry /*[8] host values*/, int step, const trajsde_noise* noise, void* stream);

/* ---- measurement hooks: HIP events recorded on the launch stream around the library's own kernel
 *      launches (bench.py's roofline leg).  mode 0 = off, 1 = only the dominant kernel (k_edge_kv on the
 *      agent-agent snapshots), 2 = every launch.  The report is text, one line per kernel tag:
 *      "<tag> <launches> <total_ms> <dominant>"; it waits for the events and resets the recorder. */
int trajsde_profile_mode(int mode);
int64_t trajsde_profile_report(char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TRAJSDE_HIP_H */
