// grid.hip -- kernels of the vanilla HiVT variant of the path (reference
// models/encoders/enc_hivt_nusargo_grid.py "GENC", models/decoders/dec_hivt_nusargo_grid.py "GDEC"):
//
//   TemporalEncoder (GENC:225-292): per actor a causal transformer over the 21 history tokens + a cls token
//     k_tr_prep        tokens = padded ? padding_token[t] : aa_out[t][n], cls appended, + pos_embed        (GENC:244-247)
//     k_node_proj<3>   (attn.hip) norm1 + in_proj -> q | k | v rows
//     k_tr_attention   one wave per actor: K and V of all 22 tokens live in registers (lane = feature), scores are
//                      head-wise lane sums, causal softmax per query -- no matrix cores, no LDS: 22x22/2 tiny dots
//     k_tr_outproj     x1 = x + out_proj(o); xn2 = norm2(x1)                                              (GENC:276)
//     k_ffn            (attn.hip) x = x1 + linear2(relu(linear1(xn2)))                                      (GENC:277)
//     k_tr_final       transformer_encoder.norm on the cls row                                             (GENC:249)
//   MLPDecoder (GDEC:47-63)
//     k_mlp_init       out = relu(LN(aggr_embed(cat(global, local)))) and the three-layer pi head
//     k_mlp_heads      loc / scale = Linear(relu(LN(Linear(out)))) with 2T outputs, ELU + 1 + min_scale on scale
#include "attn_common.hpp"
#include "common.hpp"
#include "dropout.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "tile.hpp"

namespace tsde {

constexpr int TR_S = 22;      // 21 history tokens + cls (the kernels are specialised for historical_steps = 21)

__global__ void k_tr_prep(const float* __restrict__ aa_out, const uint8_t* __restrict__ pad, const float* __restrict__ tok, int N,
                          int TT, float* __restrict__ X) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(N) * TR_S * 64) return;
  const int c = int(i & 63), s = int((i >> 6) % TR_S), n = int((i >> 6) / TR_S);
  float v;
  if (s == TR_S - 1) v = tok[EncGridBlob::TOK_CLS + c];
  else v = pad[int64_t(n) * TT + s] ? tok[EncGridBlob::TOK_PAD + s * 64 + c] : aa_out[(int64_t(s) * N + n) * 64 + c];
  X[i] = v + tok[EncGridBlob::TOK_POS + s * 64 + c];
}

// rows of q, k, v, o are [n][s][64]; nn.MultiheadAttention: q scaled by dh^-0.5, keys j <= i (causal mask GENC:251-255).
// DROP: the module's dropout on the softmax output (train mode), factors from dropout.hpp drop_tr_attn8
template <int HEADS, bool DROP>
__global__ __launch_bounds__(256) void k_tr_attention(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int N, float* __restrict__ o, DropArg drop) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  constexpr float SCALE = HEADS == 4 ? 0.25f : INV_SQRT_DH;
  const int head = lane / (64 / HEADS);
  const int64_t base = int64_t(n) * TR_S * 64 + lane;
  float kr[TR_S], vr[TR_S];
#pragma unroll
  for (int j = 0; j < TR_S; ++j) {
    kr[j] = k[base + j * 64];
    vr[j] = v[base + j * 64];
  }
#pragma unroll
  for (int i = 0; i < TR_S; ++i) {
    const float qd = q[base + i * 64] * SCALE;
    float p[TR_S];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      p[j] = head_sum_n<HEADS>(qd * kr[j]);
      m = fmaxf(m, p[j]);
    }
    float mk[24];
    if (DROP) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (8 * c <= i) drop_tr_attn8<HEADS>(mk + 8 * c, drop, uint32_t(n), head, i, c);
    }
    float s = 0.f, acc = 0.f;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const float e = fast_exp(p[j] - m);
      s += e;
      acc = fmaf(DROP ? e * mk[j] : e, vr[j], acc);
    }
    o[base + i * 64] = acc / s;
  }
}
template __global__ void k_tr_attention<4, false>(const float*, const float*, const float*, int, float*, DropArg);
template __global__ void k_tr_attention<8, false>(const float*, const float*, const float*, int, float*, DropArg);
template __global__ void k_tr_attention<4, true>(const float*, const float*, const float*, int, float*, DropArg);
template __global__ void k_tr_attention<8, true>(const float*, const float*, const float*, int, float*, DropArg);

int launch_tr_attention(int heads, const float* q, const float* k, const float* v, int N, float* o, const DropArg& drop, hipStream_t st) {
  const bool d = drop.p > 0.f;
  if (heads == 4) {
    if (d) TS_LAUNCH_TAG("k_tr_attention<drop>", false, (k_tr_attention<4, true>), cdiv(N, 4), 256, 0, st, q, k, v, N, o, drop);
    else TS_LAUNCH_TAG("k_tr_attention", false, (k_tr_attention<4, false>), cdiv(N, 4), 256, 0, st, q, k, v, N, o, drop);
  } else {
    if (d) TS_LAUNCH_TAG("k_tr_attention<drop>", false, (k_tr_attention<8, true>), cdiv(N, 4), 256, 0, st, q, k, v, N, o, drop);
    else TS_LAUNCH_TAG("k_tr_attention", false, (k_tr_attention<8, false>), cdiv(N, 4), 256, 0, st, q, k, v, N, o, drop);
  }
  return TRAJSDE_OK;
}

// x1 = x + dropout1(out_proj(o)) (GENC:270-276; the dropout in train mode only: DK_PROJ factors of the token row)
__global__ __launch_bounds__(512) void k_tr_outproj(const float* __restrict__ img, const float* __restrict__ o,
                                                    const float* __restrict__ x, int64_t R, float* __restrict__ x1,
                                                    float* __restrict__ xn2, DropArg drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, TrOutL::SIZE);
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (R + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < R ? row : R - 1;
    f4 a[4], t[4];
    load_row(a, o, r, L.g);
    linear<4, 4>(t, a, lds + TrOutL::WOUT, lds + TrOutL::BOUT, L);
    if (drop.p > 0.f) {
      f4 mk[4];
      drop_feat16(mk, drop, DK_PROJ, uint32_t(r), 0, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) t[jt] *= mk[jt];
    }
    load_row(a, x, r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) t[jt] += a[jt];
    if (row < R) store_row(t, x1, row, L.g);
    layer_norm<4>(t, lds + TrOutL::N2G, lds + TrOutL::N2B, L.g);
    if (row < R) store_row(t, xn2, row, L.g);
  }
}

// out[n] = LayerNorm(x[n][cls])
__global__ __launch_bounds__(256) void k_tr_final(const float* __restrict__ norm, const float* __restrict__ x, int N,
                                                  float* __restrict__ out) {
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int ntiles = (N + 15) / 16;
  for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
    const int row = tile * 16 + L.n, r = row < N ? row : N - 1;
    f4 a[4];
    load_row(a, x, int64_t(r) * TR_S + (TR_S - 1), L.g);
    layer_norm<4>(a, norm, norm + 64, L.g);
    if (row < N) store_row(a, out, row, L.g);
  }
}

// ------------------------------------------------------------------ MLPDecoder
__global__ __launch_bounds__(512) void k_mlp_init(const float* __restrict__ img, const float* __restrict__ local,
                                                  const float* __restrict__ global, int N, int K, float* __restrict__ out,
                                                  float* __restrict__ pi) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, MlpInitL::SIZE);
  using I = MlpInitL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t rows = int64_t(N) * K, ntiles = (rows + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < rows ? row : rows - 1;
    f4 gl[4], lo[4], a[4], b[4];
    load_row(gl, global, r, L.g);
    load_row(lo, local, r % N, L.g);
    load_vec<4>(a, lds + I::BA, L.g);
    linear_acc<4, 4>(a, gl, lds + I::WA_G, L.lane);
    linear_acc<4, 4>(a, lo, lds + I::WA_L, L.lane);
    layer_norm<4>(a, lds + I::AG, lds + I::AE, L.g);
    relu<4>(a);
    if (row < rows) store_row(a, out, row, L.g);
    load_vec<4>(a, lds + I::BP, L.g);
    linear_acc<4, 4>(a, lo, lds + I::WP_L, L.lane);
    linear_acc<4, 4>(a, gl, lds + I::WP_G, L.lane);
    layer_norm<4>(a, lds + I::PG, lds + I::PE, L.g);
    relu<4>(a);
    linear<4, 4>(b, a, lds + I::WP3, lds + I::BP3, L);
    layer_norm<4>(b, lds + I::PG4, lds + I::PE4, L.g);
    relu<4>(b);
    const float p = row_dot(b, lds + I::WP6, L.g) + lds[I::BP6];
    if (row < rows && L.g == 0) pi[(r % N) * K + (r / N)] = p;        // [N, K]  (.t() at GDEC:50)
  }
}

// loc [K*N][T][4] = (x, y, scale_x, scale_y); lane (n, g) ends up with outputs 16jt+4g .. +3 = steps 8jt+2g, 8jt+2g+1
__global__ __launch_bounds__(512) void k_mlp_heads(const float* __restrict__ img, const float* __restrict__ out, int64_t rows,
                                                   int T, float min_scale, float* __restrict__ loc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, MlpHeadsL::SIZE);
  using H = MlpHeadsL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < rows ? row : rows - 1;
    f4 a[4], h[4], lo[8], sc[8];
    load_row(a, out, r, L.g);
    linear<4, 4>(h, a, lds + H::L_W0, lds + H::L_B0, L);
    layer_norm<4>(h, lds + H::L_G, lds + H::L_E, L.g);
    relu<4>(h);
    linear<8, 4>(lo, h, lds + H::L_W3, lds + H::L_B3, L);
    linear<4, 4>(h, a, lds + H::S_W0, lds + H::S_B0, L);
    layer_norm<4>(h, lds + H::S_G, lds + H::S_E, L.g);
    relu<4>(h);
    linear<8, 4>(sc, h, lds + H::S_W3, lds + H::S_B3, L);
    if (row >= rows) continue;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      const int t0 = 8 * jt + 2 * L.g;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (t0 + u >= T) continue;
        float sx = sc[jt][2 * u], sy = sc[jt][2 * u + 1];
        sx = (sx > 0.f ? sx : fast_exp(sx) - 1.0f) + 1.0f + min_scale;          // ELU(alpha=1) + 1 + min_scale (GDEC:55-56)
        sy = (sy > 0.f ? sy : fast_exp(sy) - 1.0f) + 1.0f + min_scale;
        *reinterpret_cast<f4*>(loc + (row * T + t0 + u) * 4) = f4{lo[jt][2 * u], lo[jt][2 * u + 1], sx, sy};
      }
    }
  }
}

}  // namespace tsde

using namespace tsde;

extern "C" {

int64_t trajsde_mlp_decoder_ws_bytes(int32_t N, int num_modes) { return align_up(int64_t(N) * num_modes * 64 * 4, 256) + 256; }

int trajsde_mlp_decoder_forward(int32_t N, int num_modes, int future_steps, const float* blob, const float* local_embed,
                                const float* global_embed, float min_scale, void* ws, int64_t ws_bytes, float* loc, float* pi,
                                void* stream_) {
  TS_REQUIRE(blob && local_embed && global_embed && ws && loc && pi, "mlp_decoder_forward: null pointer");
  TS_REQUIRE(N > 0 && num_modes > 0 && future_steps > 0 && future_steps <= 64, "mlp_decoder_forward: need 0 < future_steps <= 64");
  if (ws_bytes < trajsde_mlp_decoder_ws_bytes(N, num_modes)) return fail(TRAJSDE_ERR_WORKSPACE, "mlp_decoder_forward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  Carver cv(ws, ws_bytes);
  const int64_t rows = int64_t(N) * num_modes, ntiles = (rows + 15) / 16;
  float* out = cv.take<float>(rows * 64);
  TS_LAUNCH(k_mlp_init, tile_grid(ntiles, 512, MlpInitL::SIZE * 4), 512, MlpInitL::SIZE * 4, st, blob + MlpDecBlob::INIT, local_embed,
            global_embed, N, num_modes, out, pi);
  TS_LAUNCH(k_mlp_heads, tile_grid(ntiles, 512, MlpHeadsL::SIZE * 4), 512, MlpHeadsL::SIZE * 4, st, blob + MlpDecBlob::HEADS, out, rows,
            future_steps, min_scale, loc);
  return TRAJSDE_OK;
}

}  // extern "C"
