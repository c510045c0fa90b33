// layouts.hpp -- offsets (in floats) of every kernel's LDS weight image, and of the images inside the three
// stage blobs.  Matrices are stored in MFMA fragment order [jo][q][lane][4] (tile.hpp), vectors plainly.
// The pack recipes in pack.hip fill exactly these offsets; the kernels read exactly these offsets.
#pragma once
#ifndef TSDE_SPLIT_H3
#define TSDE_SPLIT_H3 1   // 1: fp16x3 split precision, 0: bf16x6 (three exact bf16 pieces, six products)
#endif

namespace tsde {

#define TS_FIELD(name, size, prev) name = prev##_END, name##_END = name + (size)

constexpr int MAT64 = 64 * 64;

// k_aa_center: SingleInputEmbedding (EMB:22-40) + bos token + norm1 + lin_q  (ENC:546-556, 563, 586)
struct AaCenterL {
  enum : int {
    S_END = 0,
    TS_FIELD(W0, 128, S), TS_FIELD(B0, 64, W0), TS_FIELD(G1, 64, B0), TS_FIELD(E1, 64, G1),
    TS_FIELD(W3, MAT64, E1), TS_FIELD(B3, 64, W3), TS_FIELD(G4, 64, B3), TS_FIELD(E4, 64, G4),
    TS_FIELD(W6, MAT64, E4), TS_FIELD(B6, 64, W6), TS_FIELD(G7, 64, B6), TS_FIELD(E7, 64, G7),
    TS_FIELD(BOS, 21 * 64, E7), TS_FIELD(N1G, 64, BOS), TS_FIELD(N1B, 64, N1G),
    TS_FIELD(WQ, MAT64, N1B), TS_FIELD(BQ, 64, WQ),
    SIZE = BQ_END
  };
};

// edge kernels: MultipleInputEmbedding (EMB:43-70) [+ lin_k | lin_v] (ENC:577-588, 759-767; AGG:51)
struct EdgeL {
  enum : int {
    S_END = 0,
    TS_FIELD(A_W0, 128, S), TS_FIELD(A_B0, 64, A_W0), TS_FIELD(A_G, 64, A_B0), TS_FIELD(A_E, 64, A_G),
    TS_FIELD(B_W0, 128, A_E), TS_FIELD(B_B0, 64, B_W0), TS_FIELD(B_G, 64, B_B0), TS_FIELD(B_E, 64, B_G),
    TS_FIELD(WA3, MAT64, B_E), TS_FIELD(WB3, MAT64, WA3), TS_FIELD(B3, 64, WB3),
    TS_FIELD(AG0, 64, B3), TS_FIELD(AE0, 64, AG0), TS_FIELD(W2, MAT64, AE0), TS_FIELD(B2, 64, W2),
    TS_FIELD(AG3, 64, B2), TS_FIELD(AE3, 64, AG3),
    EMB_SIZE = AE3_END,
    TS_FIELD(WKV, 2 * MAT64, AE3), TS_FIELD(BKV, 128, WKV),
    SIZE = BKV_END
  };
};

// the same two images with the matrices as three bf16 planes (bf16x6 split precision, tile.hpp): a 64x64
// matrix takes 3 * 4096 * 2 B = 6144 floats
#if TSDE_SPLIT_H3
constexpr int MAT64X6 = 4096;   // two fp16 planes
#else
constexpr int MAT64X6 = 6144;
#endif
// Linear(2,64) -> LayerNorm with the statistics in closed form (tile.hpp in2_ln): the pre-LN activation is affine in the
// two inputs, h - mean(h) = w0c x0 + w1c x1 + bc with the feature-centred columns, so
//   var(h) = |L^T (x0, x1, 1)|^2   for the Cholesky factor L of the 3x3 Gram matrix of (w0c, w1c, bc) / 64
// -- a sum of three squares, no cancellation -- and LN(h) = rstd (GW0 x0 + GW1 x1 + GB) + beta with gamma folded in.
struct In2L {
  enum : int { GW0 = 0, GW1 = 64, GB = 128, CH = 192, SIZE = 200 };   // CH: l00 l10 l20 l11 l21 l22 (+2 pad)
};
// The same block as ONE matrix-core product per 16 output features (fp16x3 build; tile.hpp in2_mfma_relu_n): the pre-ReLU
// activation  rstd (GW0 x0 + GW1 x1 + GB) + beta  is an inner product over 14 of the 32 k-slots of a 16x16x32 step,
//   row operand (every lane of a row alike):  x0r_h x1r_h x0r_l x1r_l r_h r_l 1 1      (x0r = x0 rstd, ..; h / l fp16 pieces)
//   fragment, lane group 0:                   GW0_h GW1_h GW0_h GW1_h GB_h GB_h be_h be_l
//   fragment, lane group 1:                   GW0_l GW1_l 0     0     GB_l 0    0    0        groups 2, 3: zeros
// i.e. the three split-precision products h h + h l + l h of every term, exact in the fp32 accumulator.  4 fragments
// [jo][lane][8] fp16 = 1024 floats replace 48 fma per 16 x 64 tile by 4 matrix instructions.
constexpr int IN2F = 1024;
struct EdgeL6 {
  enum : int {
    S_END = 0,
    TS_FIELD(A_W0, 128, S), TS_FIELD(A_B0, 64, A_W0), TS_FIELD(A_G, 64, A_B0), TS_FIELD(A_E, 64, A_G),
    TS_FIELD(B_W0, 128, A_E), TS_FIELD(B_B0, 64, B_W0), TS_FIELD(B_G, 64, B_B0), TS_FIELD(B_E, 64, B_G),
    TS_FIELD(WA3, MAT64X6, B_E), TS_FIELD(WB3, MAT64X6, WA3), TS_FIELD(B3, 64, WB3),
    TS_FIELD(AG0, 64, B3), TS_FIELD(AE0, 64, AG0), TS_FIELD(W2, MAT64X6, AE0), TS_FIELD(B2, 64, W2),
    TS_FIELD(AG3, 64, B2), TS_FIELD(AE3, 64, AG3),
    TS_FIELD(A_C, In2L::SIZE, AE3), TS_FIELD(B_C, In2L::SIZE, A_C),     // closed-form first layers of the two branches
    TS_FIELD(A_F, IN2F, B_C), TS_FIELD(B_F, IN2F, A_F),                 // ... and their matrix-core fragments (IN2F)
    EMB_SIZE = B_F_END,
    TS_FIELD(WKV, 2 * MAT64X6, B_F), TS_FIELD(BKV, 128, WKV),
    SIZE = BKV_END
  };
};
// image of the fused edge-attention kernel (attn.hip k_edge_attn2).  The embedding of EdgeL6 with the LayerNorm algebra
// moved into the weights (pack.hip recipe_edge_fused):
//   WA3 | WB3 | B3 and W2 | B2 are FEATURE-CENTRED (the mean over the 64 outputs removed), so the rows that leave the matrix
//     cores are already y - mean(y): the LayerNorms behind them reduce the variance only;
//   WKV = [lin_k | lin_v] diag(gamma3): the last LayerNorm's gamma rides in the consumer, the kernel multiplies by rstd only;
//   CK = lin_k(beta3), CV = lin_v(beta3) (biases included) are the constant parts of k and v.  q . CK is the same for every
//     edge of a (target, head): the softmax does not see it (it is added back to the saved maximum of the training tape),
//     and sum_e alpha_e CV = CV is added once per target by k_seg_merge.  Both are read from the blob in global memory.
struct EdgeL6F {
  enum : int {
    S_END = 0,
    TS_FIELD(A_E, 64, S), TS_FIELD(B_E, 64, A_E),
    TS_FIELD(WA3, MAT64X6, B_E), TS_FIELD(WB3, MAT64X6, WA3), TS_FIELD(B3, 64, WB3),
    TS_FIELD(AG0, 64, B3), TS_FIELD(AE0, 64, AG0), TS_FIELD(W2, MAT64X6, AE0), TS_FIELD(B2, 64, W2),
    TS_FIELD(AG3, 64, B2), TS_FIELD(AE3, 64, AG3),                      // the embedding rows of the training tape only
    TS_FIELD(A_C, In2L::SIZE, AE3), TS_FIELD(B_C, In2L::SIZE, A_C),
    TS_FIELD(WKV, 2 * MAT64X6, B_C),
    TS_FIELD(A_F, IN2F, WKV), TS_FIELD(B_F, IN2F, A_F),                 // the two first layers as matrix-core fragments (below)
    LDS_SIZE = B_F_END,
    TS_FIELD(CK, 64, B_F), TS_FIELD(CV, 64, CK),
    SIZE = CV_END
  };
};
// the embedding part of EdgeL6F alone (the global interactor's relative-pose embedding, attn.hip k_edge_embed2): the same centred
// matrices and matrix-core first layers, no key / value image; AG3 | AE3 are applied by the kernel (its output IS the LayerNorm's)
struct EdgeL6G {
  enum : int {
    S_END = 0,
    TS_FIELD(A_E, 64, S), TS_FIELD(B_E, 64, A_E),
    TS_FIELD(WA3, MAT64X6, B_E), TS_FIELD(WB3, MAT64X6, WA3), TS_FIELD(B3, 64, WB3),
    TS_FIELD(AG0, 64, B3), TS_FIELD(AE0, 64, AG0), TS_FIELD(W2, MAT64X6, AE0), TS_FIELD(B2, 64, W2),
    TS_FIELD(AG3, 64, B2), TS_FIELD(AE3, 64, AG3),
    TS_FIELD(A_C, In2L::SIZE, AE3), TS_FIELD(B_C, In2L::SIZE, A_C),
    TS_FIELD(A_F, IN2F, B_C), TS_FIELD(B_F, IN2F, A_F),
    SIZE = B_F_END
  };
};
struct GEdgeL6 {
  enum : int { S_END = 0, TS_FIELD(WKV, 2 * MAT64X6, S), TS_FIELD(BKV, 128, WKV), SIZE = BKV_END };
};
static_assert(EdgeL6::SIZE * 4 <= 160 * 1024, "split-precision edge image must fit LDS");

// k_node_update: gate / self / out_proj / norm2 (ENC:595-600, 609; AGG:119-124, 131)
struct UpdL {
  enum : int {
    S_END = 0,
    TS_FIELD(WIH, MAT64, S), TS_FIELD(BIH, 64, WIH), TS_FIELD(WHH, MAT64, BIH), TS_FIELD(BHH, 64, WHH),
    TS_FIELD(WSELF, MAT64, BHH), TS_FIELD(BSELF, 64, WSELF), TS_FIELD(WOUT, MAT64, BSELF), TS_FIELD(BOUT, 64, WOUT),
    TS_FIELD(N2G, 64, BOUT), TS_FIELD(N2B, 64, N2G),
    SIZE = N2B_END
  };
};

struct UpdL6 {   // split-precision twin of UpdL
  enum : int {
    S_END = 0,
    TS_FIELD(WIH, MAT64X6, S), TS_FIELD(BIH, 64, WIH), TS_FIELD(WHH, MAT64X6, BIH), TS_FIELD(BHH, 64, WHH),
    TS_FIELD(WSELF, MAT64X6, BHH), TS_FIELD(BSELF, 64, WSELF), TS_FIELD(WOUT, MAT64X6, BSELF), TS_FIELD(BOUT, 64, WOUT),
    TS_FIELD(N2G, 64, BOUT), TS_FIELD(N2B, 64, N2G),
    SIZE = N2B_END
  };
};
// split-precision FFN: the 256 hidden units are processed in two halves of 128 so that each half's image
// (W1 half 128x64 + W2 half 64x128 as bf16x6 planes, 98 KB) fits LDS; HALF = size of one half image
struct FfnL6 {
  enum : int {
    S_END = 0,
    TS_FIELD(W1, 2 * MAT64X6, S), TS_FIELD(B1, 128, W1), TS_FIELD(W2, 2 * MAT64X6, B1), TS_FIELD(B2, 64, W2),
    HALF = B2_END, SIZE = 2 * B2_END
  };
};
static_assert(UpdL6::SIZE * 4 <= 160 * 1024 && FfnL6::HALF * 4 <= 160 * 1024, "split-precision node images must fit LDS");

// k_ffn: mlp.0 (64->256) ReLU mlp.3 (256->64)
struct FfnL {
  enum : int {
    S_END = 0,
    TS_FIELD(W1, 4 * MAT64, S), TS_FIELD(B1, 256, W1), TS_FIELD(W2, 4 * MAT64, B1), TS_FIELD(B2, 64, W2),
    SIZE = B2_END
  };
};

// k_node_proj<NQ>: norm1 + NQ stacked 64x64 projections (AL: q; global layers: q | k_node | v_node)
template <int NQ>
struct NodeProjL {
  enum : int {
    S_END = 0,
    TS_FIELD(N1G, 64, S), TS_FIELD(N1B, 64, N1G), TS_FIELD(W, NQ * MAT64, N1B), TS_FIELD(B, NQ * 64, W),
    SIZE = B_END
  };
};

// global edge kernel: lin_k_edge | lin_v_edge (AGG:110-112)
struct GEdgeL {
  enum : int { S_END = 0, TS_FIELD(WKV, 2 * MAT64, S), TS_FIELD(BKV, 128, WKV), SIZE = BKV_END };
};

// one drift/diffusion pair (FFunc / GFunc: ENC:372-440, DEC:107-158); WS/WC = columns 64/65 (sin t, cos t)
struct DriftL {
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64, S), TS_FIELD(WS, 64, W0), TS_FIELD(WC, 64, WS), TS_FIELD(B0, 64, WC),
    TS_FIELD(W2, MAT64, B0), TS_FIELD(B2, 64, W2), TS_FIELD(W4, MAT64, B2), TS_FIELD(B4, 64, W4),
    SIZE = B4_END
  };
};
struct DiffL {
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64, S), TS_FIELD(WS, 64, W0), TS_FIELD(WC, 64, WS), TS_FIELD(B0, 64, WC),
    TS_FIELD(W2, MAT64, B0), TS_FIELD(B2, 64, W2), TS_FIELD(W4, 64, B2), TS_FIELD(B4, 4, W4),
    SIZE = B4_END
  };
};

struct DriftL6 {   // the three 64x64 matrices as bf16x6 planes
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64X6, S), TS_FIELD(WS, 64, W0), TS_FIELD(WC, 64, WS), TS_FIELD(B0, 64, WC),
    TS_FIELD(W2, MAT64X6, B0), TS_FIELD(B2, 64, W2), TS_FIELD(W4, MAT64X6, B2), TS_FIELD(B4, 64, W4),
    SIZE = B4_END
  };
};
struct DiffL6 {
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64X6, S), TS_FIELD(WS, 64, W0), TS_FIELD(WC, 64, WS), TS_FIELD(B0, 64, WC),
    TS_FIELD(W2, MAT64X6, B0), TS_FIELD(B2, 64, W2), TS_FIELD(W4, 64, B2), TS_FIELD(B4, 4, W4),
    SIZE = B4_END
  };
};

struct EncSdeL {
  enum : int { F = 0, GN = F + DriftL::SIZE, GA = GN + DiffL::SIZE, SIZE = GA + DiffL::SIZE };
};

// GRU_Unit (ODEU:111-152).  y_concat = [h, x] (cols 0..63 = h); combined = [x, r*h] (cols 0..63 = x)
struct EncGruL {
  enum : int {
    S_END = 0,
    TS_FIELD(WUR_H, 2 * MAT64, S), TS_FIELD(WUR_X, 2 * MAT64, WUR_H), TS_FIELD(BUR, 128, WUR_X),
    TS_FIELD(WU2, MAT64, BUR), TS_FIELD(BU2, 64, WU2), TS_FIELD(WR2, MAT64, BU2), TS_FIELD(BR2, 64, WR2),
    TS_FIELD(WN_X, MAT64, BR2), TS_FIELD(WN_H, MAT64, WN_X), TS_FIELD(BN0, 64, WN_H),
    TS_FIELD(WN2, MAT64, BN0), TS_FIELD(BN2, 64, WN2),
    SIZE = BN2_END
  };
};

// the 16 matrices of one recurrence iteration as split-precision images, for the register-resident cooperative kernel
// (recur.hip): index m -> offset m * MAT64X6.  Vectors (biases, time columns, diffusion head) stay in EncSdeL / EncGruL.
struct EncCoopL6 {
  enum : int { F0 = 0, F2, F4, N0, N2, A0, A2, UH, RH, UX, RX, U2, R2, NX, NH, N2G, COUNT, SIZE = COUNT * MAT64X6 };
};

// encoder stage blob
// wave-per-target attention (k_global_attn and its backward): the key / value maps of the EDGE rows as plain row-major
// matrices -- lin_k_edge / lin_v_edge of a global layer, lin_k / lin_v of the AA and AL encoders (training path)
struct GAttnL {
  enum : int { S_END = 0, TS_FIELD(WKE, MAT64, S), TS_FIELD(BKE, 64, WKE), TS_FIELD(WVE, MAT64, BKE), TS_FIELD(BVE, 64, WVE), SIZE = BVE_END };
};
struct EncBlob {
  enum : int {
    AA_CENTER = 0,
    AA_EDGE = AA_CENTER + AaCenterL::SIZE,
    AA_UPD = AA_EDGE + EdgeL::SIZE,
    AA_FFN = AA_UPD + UpdL::SIZE,
    SDE = AA_FFN + FfnL::SIZE,
    GRU = SDE + EncSdeL::SIZE,
    HIDDEN = GRU + EncGruL::SIZE,
    AL_Q = HIDDEN + 64,
    AL_EDGE = AL_Q + NodeProjL<1>::SIZE,
    AL_UPD = AL_EDGE + EdgeL::SIZE,
    AL_FFN = AL_UPD + UpdL::SIZE,
    AA_EDGE6 = AL_FFN + FfnL::SIZE,
    AL_EDGE6 = AA_EDGE6 + EdgeL6::SIZE,
    AA_UPD6 = AL_EDGE6 + EdgeL6::SIZE,
    AA_FFN6 = AA_UPD6 + UpdL6::SIZE,
    AL_UPD6 = AA_FFN6 + FfnL6::SIZE,
    AL_FFN6 = AL_UPD6 + UpdL6::SIZE,
    COOP6 = AL_FFN6 + FfnL6::SIZE,
    AA_ATTN = COOP6 + EncCoopL6::SIZE,
    AL_ATTN = AA_ATTN + GAttnL::SIZE,
    AA_EDGE6F = AL_ATTN + GAttnL::SIZE,                    // each followed by its twin in the 32x32x16 fragment order (edge32.hip)
    AL_EDGE6F = AA_EDGE6F + 2 * EdgeL6F::SIZE,
    SIZE = AL_EDGE6F + 2 * EdgeL6F::SIZE
  };
};

// aggregator stage blob: rel_embed, then per layer {qkv, edge, upd, ffn}, then norm + per-mode projection
struct AggLayerL {
  enum : int {
    QKV = 0, EDGE = QKV + NodeProjL<3>::SIZE, UPD = EDGE + GEdgeL::SIZE, FFN = UPD + UpdL::SIZE, EDGE6 = FFN + FfnL::SIZE,
    ATTN = EDGE6 + GEdgeL6::SIZE, UPD6 = ATTN + GAttnL::SIZE, FFN6 = UPD6 + UpdL6::SIZE, SIZE = FFN6 + FfnL6::SIZE
  };
};
struct AggBlob {
  static constexpr int REL = 0;
  static constexpr int REL6 = EdgeL::EMB_SIZE;
  static constexpr int REL6G = REL6 + EdgeL6::EMB_SIZE;                         // the fused form's image (EdgeL6G)
  static constexpr int layer(int i) { return REL6G + EdgeL6G::SIZE + i * AggLayerL::SIZE; }
  static constexpr int norm(int nl) { return layer(nl); }                       // gamma | beta
  static constexpr int proj(int nl, int k) { return norm(nl) + 128 + k * (MAT64 + 64); }  // W_k frag | b_k
  static constexpr int size(int nl, int K) { return proj(nl, K); }
};

// decoder stage blob
struct DecInitL {
  enum : int {
    S_END = 0,
    TS_FIELD(WA_G, MAT64, S), TS_FIELD(WA_L, MAT64, WA_G), TS_FIELD(BA, 64, WA_L), TS_FIELD(AG, 64, BA), TS_FIELD(AE, 64, AG),
    TS_FIELD(WP_L, MAT64, AE), TS_FIELD(WP_G, MAT64, WP_L), TS_FIELD(BP, 64, WP_G), TS_FIELD(PG, 64, BP), TS_FIELD(PE, 64, PG),
    TS_FIELD(WP3, 64, PE), TS_FIELD(BP3, 4, WP3),
    SIZE = BP3_END
  };
};
struct HeadL {   // Linear(64,64) LN ReLU Linear(64,2)   (DEC:50-61)
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64, S), TS_FIELD(B0, 64, W0), TS_FIELD(G, 64, B0), TS_FIELD(E, 64, G), TS_FIELD(W3, 128, E), TS_FIELD(B3, 4, W3),
    SIZE = B3_END
  };
};
struct DecSdeL {
  enum : int { F = 0, G = F + DriftL::SIZE, LOC = G + DiffL::SIZE, SCALE = LOC + HeadL::SIZE, SIZE = SCALE + HeadL::SIZE };
};
// the loc and scale heads of the fused decode kernel as ONE pair: their first layers stacked into a 128x64 split-
// precision matrix (one split of the state feeds both), then each head's LayerNorm and Linear(64,2).  Only with fp16x3,
// where the pair is as large as two fp32 heads; the bf16x6 planes would not fit the 160 KB LDS next to drift+diffusion.
struct HeadPairL6 {
  enum : int {
    S_END = 0,
    TS_FIELD(W0, 2 * MAT64X6, S), TS_FIELD(B0, 128, W0),
    TS_FIELD(G_LOC, 64, B0), TS_FIELD(E_LOC, 64, G_LOC), TS_FIELD(W3_LOC, 128, E_LOC), TS_FIELD(B3_LOC, 4, W3_LOC),
    TS_FIELD(G_SC, 64, B3_LOC), TS_FIELD(E_SC, 64, G_SC), TS_FIELD(W3_SC, 128, E_SC), TS_FIELD(B3_SC, 4, W3_SC),
    SIZE = B3_SC_END
  };
};
// fused decode kernel image: drift + diffusion in split precision; the heads as HeadPairL6 at LOC (fp16x3) or as two
// plain fp32 heads (bf16x6)
struct DecSdeL6 {
#if TSDE_SPLIT_H3
  // fp16x3: the first layers of drift and diffusion stacked into ONE 128x64 matrix (rows 0..63 drift, 64..127 diffusion; one split
  // of the state feeds both) with their bias / sin / cos columns stacked alike, then the rest of the two nets, then the head pair
  enum : int {
    S_END = 0,
    TS_FIELD(W0FG, 2 * MAT64X6, S), TS_FIELD(B0FG, 128, W0FG), TS_FIELD(WSFG, 128, B0FG), TS_FIELD(WCFG, 128, WSFG),
    TS_FIELD(F_W2, MAT64X6, WCFG), TS_FIELD(F_B2, 64, F_W2), TS_FIELD(F_W4, MAT64X6, F_B2), TS_FIELD(F_B4, 64, F_W4),
    TS_FIELD(G_W2, MAT64X6, F_B4), TS_FIELD(G_B2, 64, G_W2), TS_FIELD(G_W4, 64, G_B2), TS_FIELD(G_B4, 4, G_W4),
    LOC = G_B4_END, SIZE = LOC + HeadPairL6::SIZE
  };
#else
  enum : int { F = 0, G = F + DriftL6::SIZE, LOC = G + DiffL6::SIZE, SCALE = LOC + HeadL::SIZE, SIZE = SCALE + HeadL::SIZE };
#endif
};
static_assert(DecSdeL6::SIZE * 4 <= 160 * 1024, "split-precision decoder image must fit LDS");
struct DecBlob {
  enum : int { INIT = 0, SDE = INIT + DecInitL::SIZE, SDE6 = SDE + DecSdeL::SIZE, SIZE = SDE6 + DecSdeL6::SIZE };
};

// ---- vanilla HiVT variant (grid.hip): TemporalEncoder (GENC:241-292) and MLPDecoder (GDEC:11-63)
struct TrOutL {       // self_attn.out_proj + norm2
  enum : int { S_END = 0, TS_FIELD(WOUT, MAT64, S), TS_FIELD(BOUT, 64, WOUT), TS_FIELD(N2G, 64, BOUT), TS_FIELD(N2B, 64, N2G), SIZE = N2B_END };
};
struct TrLayerL {     // norm1 + in_proj (q | k | v), out_proj + norm2, linear1 / linear2
  enum : int { QKV = 0, OUT = QKV + NodeProjL<3>::SIZE, FFN = OUT + TrOutL::SIZE, SIZE = FFN + FfnL::SIZE };
};
struct EncGridBlob {  // the AA / AL images at their EncBlob offsets (SDE / GRU regions unused), then the temporal encoder
  static constexpr int TOK = EncBlob::SIZE;                      // padding_token [21][64] | cls_token [64] | pos_embed [22][64]
  static constexpr int TOK_PAD = 0, TOK_CLS = 21 * 64, TOK_POS = 22 * 64, TOK_SIZE = 44 * 64;
  static constexpr int layer(int i) { return TOK + TOK_SIZE + i * TrLayerL::SIZE; }
  static constexpr int norm(int nl) { return layer(nl); }        // gamma | beta of transformer_encoder.norm
  static constexpr int size(int nl) { return norm(nl) + 128; }
};
struct MlpInitL {     // aggr_embed (-> out) and the three-layer pi head
  enum : int {
    S_END = 0,
    TS_FIELD(WA_G, MAT64, S), TS_FIELD(WA_L, MAT64, WA_G), TS_FIELD(BA, 64, WA_L), TS_FIELD(AG, 64, BA), TS_FIELD(AE, 64, AG),
    TS_FIELD(WP_L, MAT64, AE), TS_FIELD(WP_G, MAT64, WP_L), TS_FIELD(BP, 64, WP_G), TS_FIELD(PG, 64, BP), TS_FIELD(PE, 64, PG),
    TS_FIELD(WP3, MAT64, PE), TS_FIELD(BP3, 64, WP3), TS_FIELD(PG4, 64, BP3), TS_FIELD(PE4, 64, PG4),
    TS_FIELD(WP6, 64, PE4), TS_FIELD(BP6, 4, WP6),
    SIZE = BP6_END
  };
};
struct MlpHeadsL {    // loc and scale: Linear LN ReLU Linear(64, 2T), the last matrix zero-padded to 128 rows
  enum : int {
    S_END = 0,
    TS_FIELD(L_W0, MAT64, S), TS_FIELD(L_B0, 64, L_W0), TS_FIELD(L_G, 64, L_B0), TS_FIELD(L_E, 64, L_G),
    TS_FIELD(L_W3, 2 * MAT64, L_E), TS_FIELD(L_B3, 128, L_W3),
    TS_FIELD(S_W0, MAT64, L_B3), TS_FIELD(S_B0, 64, S_W0), TS_FIELD(S_G, 64, S_B0), TS_FIELD(S_E, 64, S_G),
    TS_FIELD(S_W3, 2 * MAT64, S_E), TS_FIELD(S_B3, 128, S_W3),
    SIZE = S_B3_END
  };
};
struct MlpDecBlob {
  enum : int { INIT = 0, HEADS = MlpInitL::SIZE, SIZE = HEADS + MlpHeadsL::SIZE };
};
static_assert(MlpInitL::SIZE * 4 <= 160 * 1024 && MlpHeadsL::SIZE * 4 <= 160 * 1024, "MLP decoder images must fit LDS");

struct MlpHeadBwdL {   // loc head of the MLP decoder: forward fields + W3^T (zero-padded to 128 columns) + W0^T
  enum : int {
    S_END = 0,
    TS_FIELD(W0, MAT64, S), TS_FIELD(B0, 64, W0), TS_FIELD(G, 64, B0), TS_FIELD(E, 64, G), TS_FIELD(W3, 2 * MAT64, E),
    TS_FIELD(B3, 128, W3), TS_FIELD(W3T, 2 * MAT64, B3), TS_FIELD(W0T, MAT64, W3T),
    SIZE = W0T_END
  };
};

// ---- backward images of the decoder stage (decoder_bwd.hip): `*T` fields hold the TRANSPOSED matrix in
// fragment order, so dX^T = W^T dY^T runs through the same linear_acc as the forward pass
struct SweepL {       // reverse Euler-Maruyama sweep: drift and diffusion nets
  enum : int {
    S_END = 0,
    TS_FIELD(F_W0T, MAT64, S), TS_FIELD(F_W2T, MAT64, F_W0T), TS_FIELD(F_W4T, MAT64, F_W2T),
    TS_FIELD(G_W0T, MAT64, F_W4T), TS_FIELD(G_W2T, MAT64, G_W0T), TS_FIELD(G_W4, 64, G_W2T),
    SIZE = G_W4_END
  };
};
struct HeadBwdL {     // loc head: forward image + W0 transposed
  enum : int { FWD = 0, W0T = HeadL::SIZE, SIZE = W0T + MAT64 };
};
struct InitBwdL {     // aggr_embed: forward fields + the two halves transposed
  enum : int {
    S_END = 0,
    TS_FIELD(WA_G, MAT64, S), TS_FIELD(WA_L, MAT64, WA_G), TS_FIELD(BA, 64, WA_L), TS_FIELD(AG, 64, BA), TS_FIELD(AE, 64, AG),
    TS_FIELD(WA_GT, MAT64, AE), TS_FIELD(WA_LT, MAT64, WA_GT),
    SIZE = WA_LT_END
  };
};
struct DecBwdBlob {
  enum : int { SWEEP = 0, HEAD = SWEEP + SweepL::SIZE, INIT = HEAD + HeadBwdL::SIZE, SIZE = INIT + InitBwdL::SIZE };
};
struct DecNllBwdBlob {   // Laplace NLL: the L2 blob followed by the scale head's images (forward + W0 transposed)
  enum : int { HEAD_SC = DecBwdBlob::SIZE, SIZE = HEAD_SC + HeadBwdL::SIZE };
};
// ---- backward images of the node-level blocks shared by the three attention families (node_bwd.hip)
struct FfnBwdAL {     // recompute h = relu(W1 xn2 + b1), dh = (W2^T dout) * (h > 0)
  enum : int { S_END = 0, TS_FIELD(W1, 4 * MAT64, S), TS_FIELD(B1, 256, W1), TS_FIELD(W2T, 4 * MAT64, B1), SIZE = W2T_END };
};
struct FfnBwdBL {     // dxn2 = W1^T dh, then norm2 backward
  enum : int { S_END = 0, TS_FIELD(W1T, 4 * MAT64, S), TS_FIELD(N2G, 64, W1T), SIZE = N2G_END };
};
struct UpdBwdL {      // gated update: forward matrices for the recompute + the four transposes
  enum : int {
    S_END = 0,
    TS_FIELD(WIH, MAT64, S), TS_FIELD(BIH, 64, WIH), TS_FIELD(WHH, MAT64, BIH), TS_FIELD(BHH, 64, WHH),
    TS_FIELD(WSELF, MAT64, BHH), TS_FIELD(BSELF, 64, WSELF),
    TS_FIELD(WOUT_T, MAT64, BSELF), TS_FIELD(WIH_T, MAT64, WOUT_T), TS_FIELD(WHH_T, MAT64, WIH_T), TS_FIELD(WSELF_T, MAT64, WHH_T),
    SIZE = WSELF_T_END
  };
};
template <int NQ>
struct ProjBwdL {     // norm1 + NQ transposed projections (NQ = 0: a bare LayerNorm backward)
  enum : int { S_END = 0, TS_FIELD(N1G, 64, S), TS_FIELD(N1B, 64, N1G), TS_FIELD(WT, NQ * MAT64, N1B), SIZE = WT_END };
};
struct EdgeBwdL {     // MultipleInputEmbedding: split-precision forward image (EdgeL6 up to EMB_SIZE, for the recompute) + the three
                      // fp32 transposes; the small per-feature vectors sit at the same offsets as in EdgeL
  enum : int { FWD = 0, W2T = EdgeL6::EMB_SIZE, WA3T = W2T + MAT64, WB3T = WA3T + MAT64, SIZE = WB3T + MAT64 };
};
static_assert(int(EdgeL6::WA3) == int(EdgeL::WA3), "the input-layer vectors of both edge images share offsets");
struct NodeBlockBwdL {   // one attention block's node-level backward images
  enum : int { FFN_A = 0, FFN_B = FFN_A + FfnBwdAL::SIZE, UPD = FFN_B + FfnBwdBL::SIZE, SIZE = UPD + UpdBwdL::SIZE };
};
// aggregator backward blob: rel_embed, per layer {node block, qkv projections, lin_k_edge | lin_v_edge plain}, final norm,
// per-mode transposed projections
struct AggLayerBwdL {
  enum : int { NODE = 0, PROJ = NODE + NodeBlockBwdL::SIZE, ATTN = PROJ + ProjBwdL<3>::SIZE, SIZE = ATTN + GAttnL::SIZE };
};
struct AggBwdBlob {
  static constexpr int REL = 0;
  static constexpr int layer(int i) { return EdgeBwdL::SIZE + i * AggLayerBwdL::SIZE; }
  static constexpr int norm(int nl) { return layer(nl); }                        // ProjBwdL<0>: gamma | beta
  static constexpr int proj(int nl, int k) { return norm(nl) + 128 + k * MAT64; }  // W_k^T fragment image
  static constexpr int size(int nl, int K) { return proj(nl, K); }
};
// ---- encoder backward images (encoder_bwd.hip)
struct EdgeKvBwdL {   // AA / AL edge kernel: split-precision embedding image + lin_k (for the recompute of emb and k), lin_k^T, lin_v^T
  enum : int { FWD = 0, WK6 = EdgeL6::EMB_SIZE, BK = WK6 + MAT64X6, WKT = BK + 64, WVT = WKT + MAT64, SIZE = WVT + MAT64 };
};
struct CenterTailL {  // center_embed (SingleInputEmbedding): forward fields in AaCenterL order up to the last LayerNorm, + W6^T
  enum : int { FWD = 0, W6T = AaCenterL::E7_END, SIZE = W6T + MAT64 };
};
struct GruBwdL {      // GRU_Unit: every matrix transposed, 64x64 blocks (h / x halves of the 128-wide inputs apart)
  enum : int {
    S_END = 0,
    TS_FIELD(WN2T, MAT64, S), TS_FIELD(WNXT, MAT64, WN2T), TS_FIELD(WNHT, MAT64, WNXT), TS_FIELD(WU2T, MAT64, WNHT),
    TS_FIELD(WR2T, MAT64, WU2T), TS_FIELD(UHT, MAT64, WR2T), TS_FIELD(RHT, MAT64, UHT), TS_FIELD(UXT, MAT64, RHT),
    TS_FIELD(RXT, MAT64, UXT),
    SIZE = RXT_END
  };
};
struct EncSdeBwdL {   // drift + the two diffusion nets, transposed
  enum : int {
    S_END = 0,
    TS_FIELD(F_W0T, MAT64, S), TS_FIELD(F_W2T, MAT64, F_W0T), TS_FIELD(F_W4T, MAT64, F_W2T),
    TS_FIELD(GN_W0T, MAT64, F_W4T), TS_FIELD(GN_W2T, MAT64, GN_W0T), TS_FIELD(GA_W0T, MAT64, GN_W2T), TS_FIELD(GA_W2T, MAT64, GA_W0T),
    TS_FIELD(GN_W4, 64, GA_W2T), TS_FIELD(GA_W4, 64, GN_W4),
    SIZE = GA_W4_END
  };
};
struct EncBwdBlob {
  enum : int {
    AA_EDGEKV = 0,
    AA_EDGEEMB = AA_EDGEKV + EdgeKvBwdL::SIZE,
    AA_NODE = AA_EDGEEMB + EdgeBwdL::SIZE,
    AA_PROJ = AA_NODE + NodeBlockBwdL::SIZE,
    AA_CTAIL = AA_PROJ + ProjBwdL<1>::SIZE,
    AA_CHEAD = AA_CTAIL + CenterTailL::SIZE,           // EdgeBwdL-shaped: first embedding layer in the A_* slots, W3^T in WA3T
    SDE = AA_CHEAD + EdgeBwdL::SIZE,
    GRU = SDE + EncSdeBwdL::SIZE,
    AL_PROJ = GRU + GruBwdL::SIZE,
    AL_EDGEKV = AL_PROJ + ProjBwdL<1>::SIZE,
    AL_EDGEEMB = AL_EDGEKV + EdgeKvBwdL::SIZE,
    AL_NODE = AL_EDGEEMB + EdgeBwdL::SIZE,
    SIZE = AL_NODE + NodeBlockBwdL::SIZE
  };
};
static_assert(GruBwdL::SIZE * 4 <= 160 * 1024 && EncSdeBwdL::SIZE * 4 <= 160 * 1024 && EdgeKvBwdL::SIZE * 4 <= 160 * 1024,
              "encoder backward images must fit LDS");
static_assert(FfnBwdAL::SIZE * 4 <= 160 * 1024 && UpdBwdL::SIZE * 4 <= 160 * 1024 && EdgeBwdL::SIZE * 4 <= 160 * 1024,
              "node backward images must fit LDS");
struct TrLayerBwdL {   // TemporalEncoderLayer backward: FFN (a, b), out_proj^T, norm1 + in_proj^T (q | k | v)
  enum : int { FFN_A = 0, FFN_B = FFN_A + FfnBwdAL::SIZE, WOUT_T = FFN_B + FfnBwdBL::SIZE, PROJ = WOUT_T + MAT64, SIZE = PROJ + ProjBwdL<3>::SIZE };
};
struct EncGridBwdBlob {   // EncBwdBlob's AA / AL images at their offsets (recurrence regions unused), then the temporal encoder
  static constexpr int TR = EncBwdBlob::SIZE;
  static constexpr int layer(int i) { return TR + i * TrLayerBwdL::SIZE; }
  static constexpr int norm(int nl) { return layer(nl); }          // ProjBwdL<0>: gamma | beta of transformer_encoder.norm
  static constexpr int size(int nl) { return norm(nl) + 128; }
};
struct MlpDecBwdBlob {
  enum : int { HEAD = 0, INIT = MlpHeadBwdL::SIZE, SIZE = INIT + InitBwdL::SIZE };
};
static_assert(SweepL::SIZE * 4 <= 160 * 1024, "sweep image must fit LDS");
static_assert(DecBwdBlob::HEAD % 4 == 0 && DecBwdBlob::INIT % 4 == 0, "16-byte aligned images");

static_assert(EdgeL::SIZE * 4 <= 160 * 1024, "edge image must fit LDS");
static_assert(FfnL::SIZE * 4 <= 160 * 1024, "ffn image must fit LDS");
static_assert(EncSdeL::SIZE * 4 <= 160 * 1024, "encoder SDE image must fit LDS");
static_assert(EncGruL::SIZE * 4 <= 160 * 1024, "GRU image must fit LDS");
static_assert(DecSdeL::SIZE * 4 <= 160 * 1024, "decoder SDE image must fit LDS");
static_assert(EncBlob::AL_Q % 4 == 0 && EncBlob::SDE % 4 == 0 && DecBlob::SDE % 4 == 0, "16-byte aligned images");

}  // namespace tsde
