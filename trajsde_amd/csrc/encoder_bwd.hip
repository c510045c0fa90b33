// encoder_bwd.hip -- backward of the LocalEncoderSDESepPara2 stage (reference models/encoders/
// enc_hivt_nusargo_sde_sep2.py:66-202) for gfx950, and the DiffBCE loss on its diffusion outputs
// (losses/diff_BCE.py:11-16).  Third family of SURVEY.md 8(f) rank 1.
//
// Inputs: dL/d local_embed [N,64] (decoder + aggregator) and the weight of the DiffBCE term.  The forward is
// recomputed keeping a tape, then walked backwards:
//   ALEncoder     node block -> attention backward over the stored embedding rows (one wave per target) -> lane embedding
//                 backward -> norm1/lin_q
//   recurrence    21 x { GRU_Unit backward, Euler-Maruyama step backward (drift + the source's diffusion net) } in one
//                 persistent launch; d latent enters at each actor's kept iteration, d DiffBCE/d g at the picked diffusion values
//   AAEncoder     the same attention chain over the 21 snapshots, then the centre embedding / bos tokens
// Matrix weight gradients come from saved (delta, input) rows through run_wgrad; everything is reduced in a fixed
// order (no atomics in this stage).
#include <string>
#include <unordered_map>

#include "attn_common.hpp"
#include "bwd.hpp"
#include "common.hpp"
#include "kernels.hpp"
#include "layouts.hpp"
#include "philox.hpp"
#include "sde_funcs.hpp"
#include "tile.hpp"
#include "tile_bwd.hpp"

namespace tsde {

// ------------------------------------------------------------------ centre embedding (SingleInputEmbedding, EMB:22-40)
// forward: a1 = relu(LN1(W0 xr + b0)); a2 = relu(LN4(W3 a1 + b3)); centre = LN7(W6 a2 + b6), replaced by the bos token
// where bos.  tail: LN7 / W6 / LN4 backward; saves A2, DA3P (for W6), A1, DA2P (for W3), XR (rotated inputs, [R,4]).
__global__ __launch_bounds__(256) void k_aa_center_bwd_tail(const float* __restrict__ img, const float* __restrict__ x,
                                                            const float* __restrict__ x_fake, const float* __restrict__ rot,
                                                            const uint8_t* __restrict__ bos, const int32_t* __restrict__ orig, int N,
                                                            int Nt, int H, const float* __restrict__ dcenter,
                                                            float* __restrict__ A1, float* __restrict__ A2, float* __restrict__ DA3P,
                                                            float* __restrict__ DA2P, float* __restrict__ XR,
                                                            float* __restrict__ vpart) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_blob(lds, img, CenterTailL::SIZE);
  using A = AaCenterL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t rows = int64_t(H) * Nt, ntiles = (rows + 15) / 16;
  f4 dg7[4], db7[4], dg4[4], db4[4];
  zero4(dg7); zero4(db7); zero4(dg4); zero4(db4);
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < rows ? row : rows - 1;
    const int t = int(r / Nt), i = int(r - int64_t(t) * Nt), o = orig[i];
    const float* xp = i < N ? x + (int64_t(i) * H + t) * 2 : x_fake + (int64_t(i - N) * H + t) * 2;
    const f4 Rm = *reinterpret_cast<const f4*>(rot + 4 * o);
    const float x0 = xp[0], x1 = xp[1];
    const float r0 = x0 * Rm[0] + x1 * Rm[2], r1 = x0 * Rm[1] + x1 * Rm[3];
    f4 a1[4], a2[4], a3[4], d[4], t4[4];
    linear_in2(a1, r0, r1, lds + A::W0, lds + A::B0, L.g);
    layer_norm<4>(a1, lds + A::G1, lds + A::E1, L.g);
    relu<4>(a1);
    linear<4, 4>(a2, a1, lds + A::W3, lds + A::B3, L);
    const float rs4 = ln_normalize(a2);                    // a2 = x_hat of LN4
    f4 act2[4];
    bool pos[16];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const f4 ga = *reinterpret_cast<const f4*>(lds + A::G4 + 16 * jt + 4 * L.g);
      const f4 be = *reinterpret_cast<const f4*>(lds + A::E4 + 16 * jt + 4 * L.g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float pre = a2[jt][c] * ga[c] + be[c];
        pos[4 * jt + c] = pre > 0.f;
        act2[jt][c] = fmaxf(pre, 0.f);
      }
    }
    linear<4, 4>(a3, act2, lds + A::W6, lds + A::B6, L);
    const float rs7 = ln_normalize(a3);                    // a3 = x_hat of LN7
    load_row(d, dcenter, r, L.g);
    if (row >= rows || bos[int64_t(o) * H + t]) zero4(d);  // bos rows take the token, not the embedding
    ln_backward(d, a3, rs7, lds + A::G7, L.g, dg7, db7);   // d := d a3p
    linear_t(t4, d, lds + CenterTailL::W6T, L);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!pos[4 * jt + c]) t4[jt][c] = 0.f;
    ln_backward(t4, a2, rs4, lds + A::G4, L.g, dg4, db4);  // t4 := d a2p
    if (row < rows) {
      store_row(a1, A1, row, L.g);
      store_row(act2, A2, row, L.g);
      store_row(d, DA3P, row, L.g);
      store_row(t4, DA2P, row, L.g);
      if (L.g == 0) *reinterpret_cast<f4*>(XR + 4 * row) = f4{r0, r1, 0.f, 0.f};
    }
  }
  float* vp = vpart + int64_t(blockIdx.x * waves + wave) * 256;
  flush_vec(dg7, vp, L);
  flush_vec(db7, vp + 64, L);
  flush_vec(dg4, vp + 128, L);
  flush_vec(db4, vp + 192, L);
}

// d bos_token[t][c] = sum over the rows (t, i) that took the token of dcentre; BOS_SLICES workgroups per t, each over a contiguous
// slice of the rows (most actors are valid from the first step on, so t = 0 holds most of the flagged rows: one workgroup per t left
// that one walking ~5 000 rows while twenty others idled), partial sums to `part[slice][t][64]`, added in slice order by a column
// sum.  Few rows take the token at t > 0, so each wave scans its contiguous share of the rows 64 at a time for the flag and only
// then loads the flagged rows, in ascending order (fixed summation order).
constexpr int BOS_SLICES = 8;
__global__ __launch_bounds__(1024) void k_bos_grad(const float* __restrict__ dcenter, const uint8_t* __restrict__ bos,
                                                   const int32_t* __restrict__ orig, int Nt, int H, float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int t = blockIdx.x, slice = blockIdx.y, c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int per = (((Nt + 16 * BOS_SLICES - 1) / (16 * BOS_SLICES)) + 63) & ~63;
  const int lo = (slice * 16 + part) * per;
  const int r0 = lo < Nt ? lo : Nt, r1 = r0 + per < Nt ? r0 + per : Nt;
  float s = 0.f;
  // the flags of eight 64-row groups at a time: their `orig` loads, then their flag loads, are requested together (two dependent
  // round trips per eight groups instead of two per group: 21 workgroups on the whole chip pay full latency for each)
  for (int base0 = r0; base0 < r1; base0 += 512) {
    int o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base0 + 64 * k + c;
      o[k] = i < r1 ? orig[i] : -1;
    }
    uint8_t fl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fl[k] = o[k] >= 0 ? bos[int64_t(o[k]) * H + t] : uint8_t(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int base = base0 + 64 * k;
      unsigned long long m = __ballot(fl[k] != 0);
      while (m) {                                            // eight flagged rows in flight, added in ascending order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v[q] = 0.f;
          if (m) {
            const int b = __ffsll(m) - 1;
            m &= m - 1;
            v[q] = dcenter[(int64_t(t) * Nt + base + b) * 64 + c];
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
      }
    }
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0) {
    float u = 0.f;
    for (int p = 0; p < 16; ++p) u += red[p][c];
    partial[(int64_t(slice) * H + t) * 64 + c] = u;
  }
}

// ------------------------------------------------------------------ recurrence: forward replay with saves
// one Euler-Maruyama step of iteration idx; slabs are [H][Nt][64] (GS, [H][Nt])
static __device__ __forceinline__ void enc_sde_save_body(const float* lds, const float* __restrict__ h_in,
                                                      const float* __restrict__ hidden0, int Nt, float dt, float sq, float sn,
                                                      float cs, int idx, NoiseArg na, const uint8_t* __restrict__ nus,
                                                      float* __restrict__ HIN, float* __restrict__ H1, float* __restrict__ H2,
                                                      float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ GS,
                                                      float* __restrict__ HODE) {
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  const int64_t off = int64_t(idx) * Nt;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    const bool live = row < Nt;
    f4 y[4], h1[4], h2[4], f[4], z[4];
    if (h_in != nullptr) load_row(y, h_in, r, L.g);
    else load_vec<4>(y, hidden0, L.g);
    if (live) store_row(y, HIN, off + row, L.g);
    const float* F = lds + EncSdeL::F;
    sde_layer0(h1, y, F, DriftL::W0, DriftL::WS, DriftL::WC, DriftL::B0, sn, cs, L);
    tanh_<4>(h1);
    linear<4, 4>(h2, h1, F + DriftL::W2, F + DriftL::B2, L);
    tanh_<4>(h2);
    linear<4, 4>(f, h2, F + DriftL::W4, F + DriftL::B4, L);
    if (live) {
      store_row(h1, H1, off + row, L.g);
      store_row(h2, H2, off + row, L.g);
    }
    const bool is_nus = nus[r] != 0;
    f4 g1[4], g2[4];
    float gs = 0.f;
#pragma unroll
    for (int net = 0; net < 2; ++net) {                        // per-row selection of g_nus / g_argo (ENC:470-482)
      const float* G = lds + (net == 0 ? EncSdeL::GN : EncSdeL::GA);
      f4 a[4], b[4];
      sde_layer0(a, y, G, DiffL::W0, DiffL::WS, DiffL::WC, DiffL::B0, sn, cs, L);
      tanh_<4>(a);
      linear<4, 4>(b, a, G + DiffL::W2, G + DiffL::B2, L);
      tanh_<4>(b);
      const float gv = fast_sigmoid(row_dot(b, G + DiffL::W4, L.g) + G[DiffL::B4]);
      if (is_nus == (net == 0)) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) { g1[jt] = a[jt]; g2[jt] = b[jt]; }
        gs = gv;
      }
    }
    if (live) {
      store_row(g1, G1, off + row, L.g);
      store_row(g2, G2, off + row, L.g);
      if (L.g == 0) GS[off + row] = gs;
    }
    noise_row(z, na, STREAM_ENCODER, idx, r, Nt, L.g);
    em_update(y, f, gs, z, dt, sq);
    if (live) store_row(y, HODE, off + row, L.g);
  }
}

// GRU_Unit of iteration idx with every activation kept; h_out = next state (not a slab)
static __device__ __forceinline__ void enc_gru_save_body(const float* lds, const float* __restrict__ x_t, int Nt,
                                                      int t, int TT, int idx, const uint8_t* __restrict__ pad,
                                                      const int32_t* __restrict__ orig, const float* __restrict__ HODE,
                                                      float* __restrict__ XS, float* __restrict__ U1, float* __restrict__ R1,
                                                      float* __restrict__ UU, float* __restrict__ RR, float* __restrict__ RH,
                                                      float* __restrict__ N1, float* __restrict__ NW, float* __restrict__ h_out) {
  using G = EncGruL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  const int64_t off = int64_t(idx) * Nt;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    f4 h[4], x[4], ur[8];
    load_row(h, HODE, off + r, L.g);
    load_row(x, x_t, r, L.g);
    load_vec<8>(ur, lds + G::BUR, L.g);
    linear_acc<8, 4>(ur, h, lds + G::WUR_H, L.lane);
    linear_acc<8, 4>(ur, x, lds + G::WUR_X, L.lane);
    tanh_<8>(ur);
    f4 u1[4] = {ur[0], ur[1], ur[2], ur[3]}, r1[4] = {ur[4], ur[5], ur[6], ur[7]};
    f4 u[4], rg[4], rh[4], n1[4], nw[4];
    linear<4, 4>(u, u1, lds + G::WU2, lds + G::BU2, L);
    sigmoid_<4>(u);
    linear<4, 4>(rg, r1, lds + G::WR2, lds + G::BR2, L);
    sigmoid_<4>(rg);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) rh[jt] = rg[jt] * h[jt];
    load_vec<4>(n1, lds + G::BN0, L.g);
    linear_acc<4, 4>(n1, x, lds + G::WN_X, L.lane);
    linear_acc<4, 4>(n1, rh, lds + G::WN_H, L.lane);
    tanh_<4>(n1);
    linear<4, 4>(nw, n1, lds + G::WN2, lds + G::BN2, L);
    const bool valid = !pad[int64_t(orig[r]) * TT + t];
    if (row < Nt) {
      const int64_t o = off + row;
      store_row(x, XS, o, L.g);
      store_row(u1, U1, o, L.g);
      store_row(r1, R1, o, L.g);
      store_row(u, UU, o, L.g);
      store_row(rg, RR, o, L.g);
      store_row(rh, RH, o, L.g);
      store_row(n1, N1, o, L.g);
      store_row(nw, NW, o, L.g);
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hn = (1.0f - u[jt][c]) * nw[jt][c] + u[jt][c] * h[jt][c];
        h[jt][c] = valid ? hn : h[jt][c];
      }
    if (row < Nt) store_row(h, h_out, row, L.g);
  }
}

// ------------------------------------------------------------------ recurrence: backward of one iteration
// dh [Nt,64] = gradient w.r.t. this iteration's output state (without the kept-latent term, added here);
// writes DHO (d h_ode), DX (d aa_out[t]) and the GRU deltas of slab idx
static __device__ __forceinline__ void enc_gru_bwd_body(const float* lds, int Nt, int N, int t, int TT, int idx,
                                                     const uint8_t* __restrict__ pad, const int32_t* __restrict__ orig,
                                                     const int32_t* __restrict__ eos, const float* __restrict__ dh,
                                                     const float* __restrict__ dlat, const float* __restrict__ HODE,
                                                     const float* __restrict__ U1, const float* __restrict__ R1,
                                                     const float* __restrict__ UU, const float* __restrict__ RR,
                                                     const float* __restrict__ N1, const float* __restrict__ NW,
                                                     float* __restrict__ DNW, float* __restrict__ DN1P, float* __restrict__ DUP,
                                                     float* __restrict__ DRP, float* __restrict__ DU1, float* __restrict__ DR1,
                                                     float* __restrict__ DHO, float* __restrict__ DX) {
  using G = GruBwdL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  const int64_t off = int64_t(idx) * Nt;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    const bool valid = !pad[int64_t(orig[r]) * TT + t] && row < Nt;
    f4 d[4], h[4], u[4], a[4];
    if (dh != nullptr) load_row(d, dh, r, L.g);
    else zero4(d);
    if (r < N && eos[r] == idx) {                              // ENC:187-188: this iteration's state is the kept latent
      load_row(a, dlat, r, L.g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) d[jt] += a[jt];
    }
    load_row(h, HODE, off + r, L.g);
    load_row(u, UU, off + r, L.g);
    load_row(a, NW, off + r, L.g);
    f4 dnw[4], du[4], dho[4], dx[4], t4[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float dd = valid ? d[jt][c] : 0.f, uu = u[jt][c];
        dnw[jt][c] = dd * (1.0f - uu);
        du[jt][c] = dd * (h[jt][c] - a[jt][c]) * uu * (1.0f - uu);          // through the sigmoid: d u_pre
        dho[jt][c] = valid ? dd * uu : d[jt][c];                            // masked rows pass the state through
      }
    // new_state_net
    linear_t(t4, dnw, lds + G::WN2T, L);
    load_row(a, N1, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) t4[jt][c] *= (1.0f - a[jt][c] * a[jt][c]);  // t4 = d n1_pre
    linear_t(dx, t4, lds + G::WNXT, L);
    f4 drh[4], rr[4], drp[4];
    linear_t(drh, t4, lds + G::WNHT, L);
    load_row(rr, RR, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float rv = rr[jt][c];
        dho[jt][c] = fmaf(drh[jt][c], rv, dho[jt][c]);
        drp[jt][c] = drh[jt][c] * h[jt][c] * rv * (1.0f - rv);                // d r_pre
      }
    if (row < Nt) {
      store_row(dnw, DNW, off + row, L.g);
      store_row(t4, DN1P, off + row, L.g);
      store_row(du, DUP, off + row, L.g);
      store_row(drp, DRP, off + row, L.g);
    }
    // gates: second layers, tanh, first layers on [h, x]
    f4 du1[4], dr1[4];
    linear_t(du1, du, lds + G::WU2T, L);
    load_row(a, U1, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) du1[jt][c] *= (1.0f - a[jt][c] * a[jt][c]);
    linear_t(dr1, drp, lds + G::WR2T, L);
    load_row(a, R1, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) dr1[jt][c] *= (1.0f - a[jt][c] * a[jt][c]);
    linear_adj<4, 4>(dho, du1, lds + G::UHT, L);
    linear_adj<4, 4>(dho, dr1, lds + G::RHT, L);
    linear_adj<4, 4>(dx, du1, lds + G::UXT, L);
    linear_adj<4, 4>(dx, dr1, lds + G::RXT, L);
    if (row < Nt) {
      store_row(du1, DU1, off + row, L.g);
      store_row(dr1, DR1, off + row, L.g);
      store_row(dho, DHO, row, L.g);
      store_row(dx, DX, row, L.g);
    }
  }
}

// d h_in of iteration idx from DHO (d h_ode) and the DiffBCE gradient on the picked diffusion values
static __device__ __forceinline__ void enc_sde_bwd_body(const float* lds, int Nt, float dt, float sq, int idx, NoiseArg na,
                                                     const uint8_t* __restrict__ nus, const int32_t* __restrict__ eos,
                                                     const float* __restrict__ DLDG, const float* __restrict__ DHO,
                                                     const float* __restrict__ H1, const float* __restrict__ H2,
                                                     const float* __restrict__ G1, const float* __restrict__ G2,
                                                     const float* __restrict__ GS, float* __restrict__ DF, float* __restrict__ DH2,
                                                     float* __restrict__ DH1, float* __restrict__ DG2N, float* __restrict__ DG1N,
                                                     float* __restrict__ DG2A, float* __restrict__ DG1A, float* __restrict__ DGPN,
                                                     float* __restrict__ DGPA, float* __restrict__ dh_out) {
  using S = EncSdeBwdL;
  const Lane L;
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int64_t ntiles = (int64_t(Nt) + 15) / 16;
  const int64_t off = int64_t(idx) * Nt;
  for (int64_t tile = int64_t(blockIdx.x) * waves + wave; tile < ntiles; tile += int64_t(gridDim.x) * waves) {
    keep_lds_reads_here();
    const int64_t row = tile * 16 + L.n, r = row < Nt ? row : Nt - 1;
    const bool live = row < Nt;
    f4 dy[4], a[4], d[4], t[4], dyn[4];
    load_row(dy, DHO, r, L.g);
    if (!live) zero4(dy);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[jt][c] = dt * dy[jt][c];
    if (live) store_row(d, DF, off + row, L.g);
    linear_t(t, d, lds + S::F_W4T, L);
    load_row(a, H2, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - a[jt][c] * a[jt][c]);
    if (live) store_row(d, DH2, off + row, L.g);
    linear_t(t, d, lds + S::F_W2T, L);
    load_row(a, H1, off + r, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - a[jt][c] * a[jt][c]);
    if (live) store_row(d, DH1, off + row, L.g);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dyn[jt] = dy[jt];
    linear_adj<4, 4>(dyn, d, lds + S::F_W0T, L);
    // diffusion: g (z sqrt h) with g one scalar per row, from the row's source net
    f4 z[4];
    noise_row(z, na, STREAM_ENCODER, idx, r, Nt, L.g);
    float cdot = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int c = 0; c < 4; ++c) cdot = fmaf(z[jt][c] * sq, dy[jt][c], cdot);
    const float gs = GS[off + r];
    float dg = row_sum(cdot);
    if (live && eos[r] == idx) dg += DLDG[r];                   // ENC:171,190-191: the diffusion value DiffBCE sees
    const float dgp = live ? dg * gs * (1.0f - gs) : 0.f;
    const bool is_nus = nus[r] != 0;
    f4 g1[4], g2[4];
    load_row(g2, G2, off + r, L.g);
    load_row(g1, G1, off + r, L.g);
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const float sel = (is_nus == (net == 0)) ? dgp : 0.f;
      const float* w4 = lds + (net == 0 ? S::GN_W4 : S::GA_W4);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f4 wv = *reinterpret_cast<const f4*>(w4 + 16 * jt + 4 * L.g);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = sel * wv[c] * (1.0f - g2[jt][c] * g2[jt][c]);
      }
      if (live) {
        store_row(d, net == 0 ? DG2N : DG2A, off + row, L.g);
        if (L.g == 0) (net == 0 ? DGPN : DGPA)[off + row] = sel;
      }
      linear_t(t, d, lds + (net == 0 ? S::GN_W2T : S::GA_W2T), L);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[jt][c] = t[jt][c] * (1.0f - g1[jt][c] * g1[jt][c]);
      if (live) store_row(d, net == 0 ? DG1N : DG1A, off + row, L.g);
      linear_adj<4, 4>(dyn, d, lds + (net == 0 ? S::GN_W0T : S::GA_W0T), L);
    }
    if (live) store_row(dyn, dh_out, row, L.g);
  }
}

// ------------------------------------------------------------------ the recurrence as two persistent kernels
// Rows never talk to each other inside the recurrence, so a wave can walk ITS tiles through all H iterations: one launch instead
// of 2 H.  The two weight images of an iteration do not fit LDS together (SDE 7 + GRU 9 matrices), so the workgroup re-stages
// them from L2 every half iteration (256 KB per workgroup and iteration).  A wave reads back only rows it wrote itself (running
// state, d h), in program order.  Measured (64 x 128 agents): 42 + 42 launches of 0.81 + 1.07 ms -> 0.71 + 0.83 ms; what remains is
// the dependent chain of ~30 products per iteration at one tile per wave (35-40 us per iteration), not launches or staging --
// wider workgroups (faster staging) do not help, only splitting a tile's products across waves would (the inference kernel
// k_enc_recur_coop does that; it keeps no tape).
struct RecurTab { float v[32][4]; };            // per iteration: dt, sqrt(dt), sin t, cos t  (H <= 32, trajsde_batch)
struct RecurSaveArgs {
  const float *img_sde, *img_gru, *hidden0, *aa_out;
  int Nt, H, TT;
  NoiseArg na;
  const uint8_t *nus, *pad;
  const int32_t* orig;
  float *HIN, *H1, *H2, *G1, *G2, *GS, *HODE, *XS, *U1, *R1, *UU, *RR, *RH, *N1, *NW, *hcur;
};
__global__ __launch_bounds__(256) void k_enc_recur_save(RecurSaveArgs a, RecurTab tab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int idx = 0; idx < a.H; ++idx) {
    const int t = a.H - 1 - idx;
    if (idx) __syncthreads();                               // every wave is done with the GRU image
    stage_blob(lds, a.img_sde, EncSdeL::SIZE);
    enc_sde_save_body(lds, idx == 0 ? nullptr : a.hcur, a.hidden0, a.Nt, tab.v[idx][0], tab.v[idx][1], tab.v[idx][2], tab.v[idx][3], idx, a.na,
                      a.nus, a.HIN, a.H1, a.H2, a.G1, a.G2, a.GS, a.HODE);
    __syncthreads();
    stage_blob(lds, a.img_gru, EncGruL::SIZE);
    enc_gru_save_body(lds, a.aa_out + int64_t(t) * a.Nt * 64, a.Nt, t, a.TT, idx, a.pad, a.orig, a.HODE, a.XS, a.U1, a.R1, a.UU, a.RR, a.RH, a.N1,
                      a.NW, a.hcur);
  }
}
struct RecurBwdArgs {
  const float *img_gru, *img_sde;
  int Nt, N, H, TT;
  NoiseArg na;
  const uint8_t *pad, *nus;
  const int32_t *orig, *eos;
  const float *dlat, *DLDG, *HODE, *U1, *R1, *UU, *RR, *N1, *NW, *H1, *H2, *G1, *G2, *GS;
  float *DNW, *DN1P, *DUP, *DRP, *DU1, *DR1, *DHO, *DAA, *DF, *DH2, *DH1, *DG2N, *DG1N, *DG2A, *DG1A, *DGPN, *DGPA, *dh;
};
// last iteration first; a.dh [Nt,64] carries d h between iterations (nothing reads the final state except through the kept
// latents, so the first one starts from zero) and holds d h of iteration 0's input on exit
__global__ __launch_bounds__(256) void k_enc_recur_bwd(RecurBwdArgs a, RecurTab tab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int idx = a.H - 1; idx >= 0; --idx) {
    const int t = a.H - 1 - idx;
    if (idx != a.H - 1) __syncthreads();
    stage_blob(lds, a.img_gru, GruBwdL::SIZE);
    enc_gru_bwd_body(lds, a.Nt, a.N, t, a.TT, idx, a.pad, a.orig, a.eos, idx == a.H - 1 ? nullptr : a.dh, a.dlat, a.HODE, a.U1, a.R1, a.UU, a.RR,
                     a.N1, a.NW, a.DNW, a.DN1P, a.DUP, a.DRP, a.DU1, a.DR1, a.DHO, a.DAA + int64_t(t) * a.Nt * 64);
    __syncthreads();
    stage_blob(lds, a.img_sde, EncSdeBwdL::SIZE);
    enc_sde_bwd_body(lds, a.Nt, tab.v[idx][0], tab.v[idx][1], idx, a.na, a.nus, a.eos, a.DLDG, a.DHO, a.H1, a.H2, a.G1, a.G2, a.GS, a.DF, a.DH2,
                     a.DH1, a.DG2N, a.DG1N, a.DG2A, a.DG1A, a.DGPN, a.DGPA, a.dh);
  }
}

// part[b][c] = sum over row slice b of s[r] * A[r][c]  (c < 64),  part[b][64] = sum s[r];  summed by k_colsum afterwards
__global__ __launch_bounds__(1024) void k_rowscale_colsum(const float* __restrict__ A, const float* __restrict__ s, int64_t R,
                                                          float* __restrict__ part) {
  __shared__ float red[16][65];
  const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t per = (R + gridDim.x - 1) / gridDim.x, lo = int64_t(blockIdx.x) * per, hi = lo + per < R ? lo + per : R;
  float acc = 0.f, ss = 0.f;
  int64_t r = lo + sub;
  for (; r + 112 < hi; r += 128) {                       // eight rows in flight, consumed in the order of the plain loop
    float sv[8], av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sv[u] = s[r + 16 * u];
      av[u] = A[(r + 16 * u) * 64 + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = fmaf(sv[u], av[u], acc);
      ss += sv[u];
    }
  }
  for (; r < hi; r += 16) {
    const float sv = s[r];
    acc = fmaf(sv, A[r * 64 + c], acc);
    ss += sv;
  }
  red[sub][c] = acc;
  if (c == 0) red[sub][64] = ss;
  __syncthreads();
  if (sub == 0) {
    float t = 0.f;
    for (int p = 0; p < 16; ++p) t += red[p][c];
    part[blockIdx.x * 128 + c] = t;
    if (c == 0) {
      float u = 0.f;
      for (int p = 0; p < 16; ++p) u += red[p][64];
      part[blockIdx.x * 128 + 64] = u;
    }
  }
}

// DiffBCE (diff_BCE.py:11-16, mean reduction): label 0 for the real target agents' picked diffusion, 1 for their
// perturbed copies.  DLDG[r] = weight * dL/dg of row r (0 for rows that are not picked); loss[0] = weight * L
__global__ __launch_bounds__(1024) void k_diffbce(const float* __restrict__ GS, const int32_t* __restrict__ eos,
                                                  const int32_t* __restrict__ pick_slot, int Nt, int A, float weight,
                                                  float* __restrict__ DLDG, float* __restrict__ loss) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int r = threadIdx.x; r < Nt; r += 1024) {
    const int slot = pick_slot[r];
    float g_ = 0.f;
    if (slot >= 0) {
      const float g = GS[int64_t(eos[r]) * Nt + r];
      if (slot < A) {                                        // diff_in, target 0: -log(1 - g)
        acc += double(-fmaxf(logf(1.0f - g), -100.0f));
        g_ = 1.0f / (1.0f - g);
      } else {                                               // diff_out, target 1: -log(g)
        acc += double(-fmaxf(logf(g), -100.0f));
        g_ = -1.0f / g;
      }
    }
    DLDG[r] = A > 0 ? weight * g_ / float(A) : 0.f;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = A > 0 ? float(double(weight) * red[0] / double(A)) : 0.f;
}

// lat[r] = state after iteration eos[r]: valid ? (1-u) nw + u h_ode : h_ode, from the saved slabs
__global__ void k_gather_latent(const float* __restrict__ HODE, const float* __restrict__ UU, const float* __restrict__ NW,
                                const int32_t* __restrict__ eos, const uint8_t* __restrict__ pad, const int32_t* __restrict__ orig, int N,
                                int Nt, int H, int TT, float* __restrict__ lat) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= int64_t(N) * 64) return;
  const int r = int(i >> 6), c = int(i & 63), idx = eos[r], t = H - 1 - idx;
  const int64_t o = (int64_t(idx) * Nt + r) * 64 + c;
  const bool valid = !pad[int64_t(orig[r]) * TT + t];
  const float h = HODE[o], u = UU[o];
  lat[i] = valid ? (1.0f - u) * NW[o] + u * h : h;
}

// ------------------------------------------------------------------ workspace
struct EncBwdWs {
  // AA tape
  float *center, *cn, *q, *emb, *stats, *agg, *x1, *xn2, *aa_out;      // emb [E_aa,64]: nbr_embed rows; stats [R,8,2]: softmax (max, 1/sum)
  // recurrence tape (slabs [H][Nt][64]) and running states
  float *HIN, *H1, *H2, *G1, *G2, *GS, *HODE, *XS, *U1, *R1, *UU, *RR, *RH, *N1, *NW, *hcur, *lat;
  // AL tape
  float *al_xn, *al_q, *al_emb, *al_stats, *al_agg, *al_x1, *al_xn2;
  // backward: recurrence deltas
  float *DF, *DH2, *DH1, *DG2N, *DG1N, *DG2A, *DG1A, *DGPN, *DGPA, *DNW, *DN1P, *DUP, *DRP, *DU1, *DR1, *DHO, *dhA, *dhB, *DLDG, *DLAT, *DAA;
  // backward: attention chain scratch (sized for the larger of the AA / AL problems)
  float* rec;                  // (target, stream) records of the fused forward attention
  float *dagg, *dxn, *DQ, *DCENTER, *EA, *ED, *RL, *SS, *DAGGM, *A1, *A2, *DA3P, *DA2P, *XR, *part, *cs, *scal, *varena;
  NodeBlockScratch nb;
  EdgeEmbedScratch ee;
  int64_t total, tape_total, scratch_total, parts;
  bool ok;
  // The forward tape (what trajsde_encoder_forward_train writes and the backward reads) is carved first, the backward's own
  // scratch after it.  With `scratch` given the two live in separate buffers: a training step then holds only the tape between
  // its forward and the encoder backward, and the scratch is allocated when the decoder's and aggregator's workspaces are gone.
  EncBwdWs(const trajsde_batch* b, const trajsde_graph* g, void* ws, int64_t bytes, void* scratch = nullptr, int64_t scratch_bytes = 0) {
    Carver t(ws, bytes);
    const int64_t H = b->H, Nt = g->Nt, N = b->N, R = H * Nt, Eaa = g->E_aa, Ela = g->E_la, E = (Eaa > Ela ? Eaa : Ela) + 1;
    // ---- tape
    float** tape_R[] = {&center, &cn, &q, &agg, &x1, &xn2, &aa_out, &HIN, &H1, &H2, &G1, &G2, &HODE, &XS, &U1, &R1, &UU, &RR, &RH, &N1, &NW};
    for (float** p : tape_R) *p = t.take<float>(R * 64);
    GS = t.take<float>(R);
    hcur = t.take<float>(Nt * 64);
    float** tape_N[] = {&lat, &al_xn, &al_q, &al_agg, &al_x1, &al_xn2};
    for (float** p : tape_N) *p = t.take<float>(N * 64);
    emb = t.take<float>(Eaa * 64 + 64);
    stats = t.take<float>(R * 16);
    al_emb = t.take<float>(Ela * 64 + 64);
    al_stats = t.take<float>(N * 16);
    {
      const int64_t ra = fused_rec_floats(Eaa, true, R), rl = fused_rec_floats(Ela, true, N);
      rec = t.take<float>(ra > rl ? ra : rl);                // (forward only)
    }
    tape_total = t.off + 256;
    // ---- backward scratch
    Carver c = scratch ? Carver(scratch, scratch_bytes) : t;
    const int64_t s_begin = scratch ? 0 : t.off;
    float** rows_R[] = {&DF, &DH2, &DH1, &DG2N, &DG1N, &DG2A, &DG1A, &DNW, &DN1P, &DUP, &DRP, &DU1, &DR1, &DAA, &dagg, &dxn, &DQ,
                        &DCENTER, &DAGGM, &nb.dx1, &nb.UPD, &nb.DGP, &nb.DS};
    for (float** p : rows_R) *p = c.take<float>(R * 64);
    // the centre-embedding backward runs last, after the recurrence's weight gradients (the last readers of its delta slabs)
    // were enqueued on the same stream: its four slabs reuse four of those
    A1 = DF; A2 = DH2; DA3P = DH1; DA2P = DG2N;
    float** rows_Rs[] = {&DGPN, &DGPA};
    for (float** p : rows_Rs) *p = c.take<float>(R);
    XR = c.take<float>(R * 4);
    nb.H = c.take<float>(R * 256);
    nb.DH = c.take<float>(R * 256);
    float** rows_Nt[] = {&DHO, &dhA, &dhB};
    for (float** p : rows_Nt) *p = c.take<float>(Nt * 64);
    DLDG = c.take<float>(Nt);
    DLAT = c.take<float>(N * 64);
    EA = c.take<float>(E * 8 + 64);
    ED = c.take<float>(E * 8 + 64);
    // per (target, head) sums of the embedding rows (run_edge_attn_bwd), [R,8,64] each.  RL lives in the node block's two
    // [R,256] slabs (contiguous), which the block's own kernels and weight gradients have finished with by then
    RL = nb.H;
    SS = c.take<float>(R * 512);
    float** rows_E[] = {&ee.DEP, &ee.DSP};               // (ee.S: the chain's embedding slab, dead by then -- run_attn_chain)
    for (float** p : rows_E) *p = c.take<float>(E * 64);
    ee.S = nullptr;
    nb.vpart = ee.vpart = c.take<float>(VPART_FLOATS);
    varena = c.take<float>(VPART_ARENA_SLABS * VPART_FLOATS);       // the producers' own slabs while the sums are deferred (bwd.hpp)
    const int64_t rows = E > R ? E : R;
    parts = wgrad_max_parts(rows, H);
    part = c.take<float>(parts * 4096);
    cs = c.take<float>(parts * 64);
    scal = c.take<float>(64);
    scratch_total = c.off - s_begin + 256;
    total = scratch ? tape_total : c.off + 256;               // single-buffer form: tape and scratch in one workspace
    ok = t.ok && c.ok;
  }
};

// attention chain of one encoder (AA or AL) given d out rows: node block, segment softmax, edge kernel, projection.
// Returns d x (the block's input rows) in `dx_out`.
struct AttnChain {
  const float *img_node, *img_proj, *img_attn, *img_emb;          // backward images (img_attn: GAttnL of lin_k / lin_v, forward blob)
  const float *geom, *q, *emb, *stats, *x;                        // graph + tape (emb: embedding rows, stats: softmax max | 1/sum)
  const int32_t *dst, *segptr;
  NodeBlockTape tp;
  int64_t R, E;
  std::string prefix, embed;                                      // parameter names: <prefix>.lin_k..., <prefix>.<embed>...
  int heads = 8;
  DropArg drop = no_drop();                                       // train-mode dropout of this block (dropout.hpp)
  const float* img_kvt = nullptr;                                 // lin_k^T | lin_v^T (EdgeKvBwdL::WKT of the backward blob)
};

}  // namespace tsde

using namespace tsde;

namespace {

struct GradTable {
  std::unordered_map<std::string, float*> slot;
  bool missing = false;
  std::string missing_name;
  float* operator()(const std::string& n) {
    auto it = slot.find(n);
    if (it == slot.end()) {
      if (!missing) missing_name = n;
      missing = true;
      return nullptr;
    }
    return it->second;
  }
};

int run_attn_chain(const AttnChain& c, const float* dout, EncBwdWs& w, const WgradCtx& wc, GradTable& G, float* dx_out,
                   hipStream_t st) {
  const std::string& p = c.prefix;
  NodeBlockGrads gr{G(p + ".lin_ih.weight"), G(p + ".lin_ih.bias"), G(p + ".lin_hh.weight"), G(p + ".lin_hh.bias"),
                    G(p + ".lin_self.weight"), G(p + ".lin_self.bias"), G(p + ".out_proj.weight"), G(p + ".out_proj.bias"),
                    G(p + ".norm2.weight"), G(p + ".norm2.bias"), G(p + ".mlp.0.weight"), G(p + ".mlp.0.bias"),
                    G(p + ".mlp.3.weight"), G(p + ".mlp.3.bias")};
  const std::string e = p + "." + c.embed;
  EdgeEmbedGrads eg{G(e + ".module_list.0.0.weight"), G(e + ".module_list.0.0.bias"), G(e + ".module_list.0.1.weight"),
                    G(e + ".module_list.0.1.bias"), G(e + ".module_list.1.0.weight"), G(e + ".module_list.1.0.bias"),
                    G(e + ".module_list.1.1.weight"), G(e + ".module_list.1.1.bias"), G(e + ".module_list.0.3.weight"),
                    G(e + ".module_list.0.3.bias"), G(e + ".module_list.1.3.weight"), G(e + ".module_list.1.3.bias"),
                    G(e + ".aggr_embed.0.weight"), G(e + ".aggr_embed.0.bias"), G(e + ".aggr_embed.2.weight"),
                    G(e + ".aggr_embed.2.bias"), G(e + ".aggr_embed.3.weight"), G(e + ".aggr_embed.3.bias")};
  float *wk = G(p + ".lin_k.weight"), *bk = G(p + ".lin_k.bias"), *wv = G(p + ".lin_v.weight"), *bv = G(p + ".lin_v.bias");
  float *wq = G(p + ".lin_q.weight"), *bq = G(p + ".lin_q.bias"), *n1g = G(p + ".norm1.weight"), *n1b = G(p + ".norm1.bias");
  TS_REQUIRE(!G.missing, "encoder_backward: parameter table lacks " + G.missing_name);
  const int64_t R = c.R, E = c.E;
  if (int rc = node_block_backward(c.img_node, c.tp, dout, R, w.nb, wc, gr, w.dagg, w.dxn, st, c.drop)) return rc;
  // Attention over the stored embedding rows, one wave per target (aggregator_bwd.hip k_gattn_bwd<.., NODE = false>): with
  // k_e = lin_k(emb_e), v_e = lin_v(emb_e) the per-edge matrices drop out --
  //   d lin_k.weight = headwise outer(q, RL),  RL_h = sum_e dlogit_e,h emb_e        d lin_k.bias = 0 (shifts a target's logits alike)
  //   d lin_v.weight = headwise outer(dagg, SS), SS_h = sum_e alpha_e,h emb_e       d lin_v.bias = column sum of DAGGM
  //   d emb_e        = Wk^T (dlogit_e,h q) + Wv^T (alpha_e,h dagg)   built inside the embedding backward (EdgeAttnGrad)
  // so no [E,64] row of k, v, d k, d v or d emb is ever written.  A target without edges gets DQ = 0, RL = SS = 0 from the kernel.
  bool weights_done = false;
  if (int rc = run_edge_attn_bwd(st, c.heads, c.img_attn, c.segptr, c.emb, c.q, c.tp.agg, w.dagg, c.stats, R, w.DQ, w.RL, w.SS, w.DAGGM,
                                 w.EA, w.ED, c.drop, &wc, wk, wv, bv, &weights_done))
    return rc;
  if (!weights_done) {
    if (int rc = run_headwise_outer(wc, c.q, w.RL, R, wk, c.heads)) return rc;
    if (int rc = run_headwise_outer(wc, w.dagg, w.SS, R, wv, c.heads)) return rc;
  }
  TS_HIP(hipMemsetAsync(bk, 0, 64 * sizeof(float), st));
  if (!weights_done)
    if (int rc = run_colsum_tall(st, w.DAGGM, R, 64, 64, bv, w.nb.vpart)) return rc;
  if (E > 0) {
    const EdgeAttnGrad ag{c.dst, c.q, w.dagg, w.EA, w.ED, c.img_kvt, c.heads};
    // the embedding rows were last read by the attention backward above: their slab becomes the embedding backward's S slab
    EdgeEmbedScratch ee = w.ee;
    ee.S = const_cast<float*>(c.emb);
    if (int rc = edge_embed_backward(c.img_emb, c.geom, nullptr, E, ee, wc, eg, st, &ag)) return rc;
  }
  const int gp = vec_grid((R + 15) / 16, 256, ProjBwdL<1>::SIZE * 4);
  float* const vp = vpart_slab(w.nb.vpart, int64_t(gp) * 4, 128);
  TS_LAUNCH(k_node_proj_bwd<1>, gp, 256, ProjBwdL<1>::SIZE * 4, st, c.img_proj, c.x, w.nb.dx1, w.dxn, w.DQ, nullptr, nullptr, R, dx_out,
            nullptr, vp);
  {
    ColsumBatch cb(st, gp * 4, 128);
    cb.add(vp, 64, n1g);
    cb.add(vp + 64, 64, n1b);
    if (int rc = cb.flush()) return rc;
  }
  return run_wgrad(wc, w.DQ, 64, c.tp.xn, 64, R, R, wq, 64, 0, bq, 0);
}

// AAEncoder backward over the H snapshots: w.DAA = d aa_out [H,Nt,64] on entry; attention chain, centre embedding, bos tokens
int aa_encoder_backward(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd, const float* blob_bwd, EncBwdWs& w,
                        const WgradCtx& wc, GradTable& G, int heads, hipStream_t st, const DropArg& drop = no_drop()) {
  using BB = EncBwdBlob;
  const int N = b->N, Nt = g->Nt, H = b->H;
  const int64_t R = int64_t(H) * Nt, Eaa = g->E_aa;
  AttnChain c{blob_bwd + BB::AA_NODE, blob_bwd + BB::AA_PROJ, blob_fwd + EncBlob::AA_ATTN, blob_bwd + BB::AA_EDGEEMB,
              g->aa_geom, w.q, w.emb, w.stats, w.center, g->aa_dst, g->aa_segptr, NodeBlockTape{w.agg, w.cn, w.x1, w.xn2}, R, Eaa,
              "aa_encoder", "nbr_embed", heads};
  c.drop = drop;
  c.img_kvt = blob_bwd + BB::AA_EDGEKV + EdgeKvBwdL::WKT;
  if (int rc = run_attn_chain(c, w.DAA, w, wc, G, w.DCENTER, st)) return rc;
  const std::string ce = "aa_encoder.center_embed.embed.";
  float *w0 = G(ce + "0.weight"), *b0 = G(ce + "0.bias"), *g1 = G(ce + "1.weight"), *e1 = G(ce + "1.bias");
  float *w3 = G(ce + "3.weight"), *b3 = G(ce + "3.bias"), *g4 = G(ce + "4.weight"), *e4 = G(ce + "4.bias");
  float *w6 = G(ce + "6.weight"), *b6 = G(ce + "6.bias"), *g7 = G(ce + "7.weight"), *e7 = G(ce + "7.bias");
  float* tok = G("aa_encoder.bos_token");
  TS_REQUIRE(!G.missing, "encoder_backward: parameter table lacks " + G.missing_name);
  const int gt = vec_grid((R + 15) / 16, 256, CenterTailL::SIZE * 4);
  float* vp = vpart_slab(w.nb.vpart, int64_t(gt) * 4, 256);
  TS_LAUNCH(k_aa_center_bwd_tail, gt, 256, CenterTailL::SIZE * 4, st, blob_bwd + BB::AA_CTAIL, b->x, g->x_fake, rot, b->bos_mask, g->orig, N,
            Nt, H, w.DCENTER, w.A1, w.A2, w.DA3P, w.DA2P, w.XR, vp);
  float* const tv[4] = {g7, e7, g4, e4};
  {
    ColsumBatch cb(st, gt * 4, 256);
    for (int i = 0; i < 4; ++i) cb.add(vp + 64 * i, 64, tv[i]);
    if (int rc = cb.flush()) return rc;
  }
  if (int rc = run_wgrad(wc, w.DA3P, 64, w.A2, 64, R, R, w6, 64, 0, b6, 0)) return rc;
  if (int rc = run_wgrad(wc, w.DA2P, 64, w.A1, 64, R, R, w3, 64, 0, b3, 0)) return rc;
  const int lds_br = (EdgeL::WA3 + MAT64) * 4;
  const int gb = vec_grid((R + 15) / 16, 256, lds_br);
  vp = vpart_slab(w.nb.vpart, int64_t(gb) * 4, 320);
  TS_LAUNCH(k_edge_embed_bwd_branch<0>, gb, 256, lds_br, st, blob_bwd + BB::AA_CHEAD, w.XR, w.DA2P, R, vp);
  {
    ColsumBatch cb(st, gb * 4, 320);
    cb.add(vp, 64, g1);
    cb.add(vp + 64, 64, e1);
    cb.add(vp + 128, 64, w0, 2);
    cb.add(vp + 192, 64, w0 + 1, 2);
    cb.add(vp + 256, 64, b0);
    if (int rc = cb.flush()) return rc;
  }
  // (the shared slab is free here: every vector sum of this call that fell back to it has run)
  TS_LAUNCH(k_bos_grad, dim3(H, BOS_SLICES), 1024, 0, st, w.DCENTER, b->bos_mask, g->orig, Nt, H, w.nb.vpart);
  return run_colsum(st, w.nb.vpart, BOS_SLICES, H * 64, H * 64, tok);
}

}  // namespace

// training-path attention of an encoder (AA / AL): the edge embedding rows are computed once and KEPT (the tape), the attention
// itself runs one wave per target over those rows (k_global_attn<.., NODE = false>: lin_k / lin_v folded into per-target
// vectors) and leaves the softmax statistics for the backward
static int edge_attention_tape(const char* tag, const float* img_fused, const float* img_edge6, const float* img_attn, const float* geom, const int32_t* dst, int64_t E,
                               const int32_t* segptr, const float* q, int64_t R, float* emb, float* stats, float* agg, float* rec, int heads,
                               const DropArg& drop, hipStream_t st) {
  if (attn_fused_enabled())     // the inference forward's own kernels, which also write the embedding rows and the statistics
    return fused_edge_attention(tag, false, img_fused, geom, dst, q, EdgeCount{E, nullptr, 0}, segptr, R, rec, agg, heads, st, drop, emb, stats);
  if (E > 0)
    TS_LAUNCH(k_edge_embed<true>, tile_grid((E + 15) / 16, 1024, EdgeL6::EMB_SIZE * 4), 1024, EdgeL6::EMB_SIZE * 4, st, img_edge6, geom,
              EdgeCount{E, nullptr, 0}, emb, 0);
  TS_EDGE_ATTN(heads, drop, xcd_grid(cdiv(R, 4)), 256, 0, st, img_attn, segptr, emb, q, R, agg, stats);
  return TRAJSDE_OK;
}

// The encoder's forward with every activation the backward needs kept in `w` (the "tape"): the split-precision kernels the
// inference forward runs -- the fused edge attention also writes the embedding rows and the softmax statistics, which is all
// the attention backward needs -- and the recurrence as one persistent launch that saves its intermediates.  Run once per training step: by
// trajsde_encoder_forward_train (which also finishes local_embed / diff_pick), or by trajsde_encoder_backward itself when no
// tape was handed over.
static int encoder_tape(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd, const float* step_tab,
                        const NoiseArg& na, EncBwdWs& w, const DropArg& drop_aa, const DropArg& drop_al, hipStream_t st) {
  const int N = b->N, Nt = g->Nt, H = b->H;
  const int64_t R = int64_t(H) * Nt, Eaa = g->E_aa, Ela = g->E_la;
  const int64_t rtiles = (int64_t(Nt) + 15) / 16;
  using FB = EncBlob;
  TS_LAUNCH(k_aa_center, tile_grid((R + 15) / 16, 512, AaCenterL::SIZE * 4), 512, AaCenterL::SIZE * 4, st, blob_fwd + FB::AA_CENTER, b->x,
            g->x_fake, rot, b->bos_mask, g->orig, N, Nt, H, w.center, w.cn, w.q);
  if (int rc = edge_attention_tape("k_edge_kv[aa]+emb", blob_fwd + FB::AA_EDGE6F, blob_fwd + FB::AA_EDGE6, blob_fwd + FB::AA_ATTN, g->aa_geom, g->aa_dst, Eaa, g->aa_segptr, w.q, R,
                                   w.emb, w.stats, w.agg, w.rec, 8, drop_aa, st))
    return rc;
  TS_LAUNCH(k_node_update<true>, tile_grid((R + 15) / 16, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, blob_fwd + FB::AA_UPD6, w.agg, w.cn,
            w.center, R, w.x1, w.xn2, drop_aa, no_merge());
  TS_LAUNCH(k_ffn6, tile_grid((R + 15) / 16, 512, FfnL6::HALF * 4), 512, FfnL6::HALF * 4, st, blob_fwd + FB::AA_FFN6, w.x1, w.xn2, R, w.aa_out, drop_aa);
  {
    RecurTab tab;
    for (int idx = 0; idx < H; ++idx) {
      const float* e = step_tab + 8 * idx;
      tab.v[idx][0] = e[1]; tab.v[idx][1] = e[2]; tab.v[idx][2] = e[3]; tab.v[idx][3] = e[4];
    }
    // The inference kernel (recur.hip k_enc_recur_coop: a tile's products split over four waves, weights in registers) with tape
    // stores beside it: 0.71 -> ~0.25 ms at 64 x 128 agents against the one-tile-per-wave form below, which stays for shapes the
    // cooperative kernel does not take (more than COOP_TMAX x 256 row tiles) and under TRAJSDE_RECUR_LEGACY=1.
    static const bool legacy = []() { const char* e = getenv("TRAJSDE_RECUR_LEGACY"); return e && atoi(e) != 0; }();
    static const int force_tw = []() { const char* e = getenv("TRAJSDE_RECUR_TW"); return e ? atoi(e) : 0; }();
    const int tiles_per_wg = force_tw > 0 ? force_tw : int((rtiles + 255) / 256);
    if (TSDE_SPLIT_H3 && !legacy && tiles_per_wg <= COOP_TMAX && H <= 32) {
      StepTab stab;
      for (int i = 0; i < H; ++i) {
        stab.dt[i] = step_tab[8 * i + 1]; stab.sq[i] = step_tab[8 * i + 2]; stab.sn[i] = step_tab[8 * i + 3]; stab.cs[i] = step_tab[8 * i + 4];
      }
      const RecurTape tp{w.HIN, w.H1, w.H2, w.G1, w.G2, w.GS, w.HODE, w.XS, w.U1, w.R1, w.UU, w.RR, w.RH, w.N1, w.NW};
      const int grid = int((rtiles + tiles_per_wg - 1) / tiles_per_wg);
      const int lds = coop_lds_floats(tiles_per_wg) * 4;
#define TS_COOP_SAVE(TW) TS_LAUNCH_TAG("k_enc_recur_coop<save>", false, (k_enc_recur_coop<TW, true>), grid, 256, lds, st, blob_fwd + FB::SDE, blob_fwd + FB::GRU, \
              blob_fwd + FB::COOP6, blob_fwd + FB::HIDDEN, w.aa_out, Nt, N, H, b->TT, tiles_per_wg, stab, 0, na, g->nus_mask, b->padding_mask, g->orig, g->eos_idx,     \
              g->pick_slot, w.lat, nullptr, nullptr, 0, tp)
      switch (tiles_per_wg) {
        case 1: TS_COOP_SAVE(1); break;
        case 2: TS_COOP_SAVE(2); break;
        case 3: TS_COOP_SAVE(3); break;
        default: TS_COOP_SAVE(4); break;
      }
#undef TS_COOP_SAVE
    } else {
      const RecurSaveArgs ra{blob_fwd + FB::SDE, blob_fwd + FB::GRU, blob_fwd + FB::HIDDEN, w.aa_out, Nt, H, b->TT, na, g->nus_mask, b->padding_mask,
                             g->orig, w.HIN, w.H1, w.H2, w.G1, w.G2, w.GS, w.HODE, w.XS, w.U1, w.R1, w.UU, w.RR, w.RH, w.N1, w.NW, w.hcur};
      const int lds_r = (EncSdeL::SIZE > EncGruL::SIZE ? EncSdeL::SIZE : EncGruL::SIZE) * 4;
      TS_LAUNCH(k_enc_recur_save, tile_grid(rtiles, 256, lds_r), 256, lds_r, st, ra, tab);
    }
  }
  // the kept latent of actor r is the state after iteration eos[r] (ENC:187-188): rebuilt from the saved slabs
  k_gather_latent<<<cdiv(int64_t(N) * 64, 256), 256, 0, st>>>(w.HODE, w.UU, w.NW, g->eos_idx, b->padding_mask, g->orig, N, Nt, H, b->TT, w.lat);
  TS_LAUNCH_CHECK("k_gather_latent");
  TS_LAUNCH(k_node_proj<1>, tile_grid((int64_t(N) + 15) / 16, 512, NodeProjL<1>::SIZE * 4), 512, NodeProjL<1>::SIZE * 4, st,
            blob_fwd + FB::AL_Q, w.lat, int64_t(N), w.al_xn, w.al_q, nullptr, nullptr);
  if (int rc = edge_attention_tape("k_edge_kv[al]+emb", blob_fwd + FB::AL_EDGE6F, blob_fwd + FB::AL_EDGE6, blob_fwd + FB::AL_ATTN, g->la_geom, g->la_dst, Ela, g->la_segptr, w.al_q,
                                   int64_t(N), w.al_emb, w.al_stats, w.al_agg, w.rec, 8, drop_al, st))
    return rc;
  TS_LAUNCH(k_node_update<true>, tile_grid((int64_t(N) + 15) / 16, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, blob_fwd + FB::AL_UPD6,
            w.al_agg, w.al_xn, w.lat, int64_t(N), w.al_x1, w.al_xn2, drop_al, no_merge());

  return TRAJSDE_OK;
}

// diff_pick[slot] = the diffusion value of the kept iteration of the rows that have a slot (ENC:171,190-191), broadcast over
// the 64 channels -- from the tape's GS [H][Nt]
__global__ void k_pick_diff(const float* __restrict__ GS, const int32_t* __restrict__ eos, const int32_t* __restrict__ pick_slot, int Nt,
                            float* __restrict__ diff_pick) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const int slot = pick_slot[r];
  if (slot >= 0) diff_pick[int64_t(slot) * 64 + lane] = GS[int64_t(eos[r]) * Nt + r];
}

extern "C" {

int64_t trajsde_encoder_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EncBwdWs w(b, g, nullptr, 0);
  return w.total;
}
int64_t trajsde_encoder_tape_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EncBwdWs w(b, g, nullptr, 0);
  return w.tape_total;
}
int64_t trajsde_encoder_backward_scratch_bytes(const trajsde_batch* b, const trajsde_graph* g) {
  if (!b || !g) return -1;
  EncBwdWs w(b, g, nullptr, 0);
  return w.scratch_total;
}

int trajsde_encoder_forward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd,
                                  const float* step_tab /*HOST [H,8]*/, const trajsde_noise* noise, void* ws, int64_t ws_bytes,
                                  float* local_embed, float* diff_pick, const trajsde_dropout* dropout, void* stream_) {
  TS_REQUIRE(b && g && rot && blob_fwd && step_tab && ws && local_embed && diff_pick, "encoder_forward_train: null pointer");
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_forward_train: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(g->exact, "encoder_forward_train: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(b->A > 0 && g->Nt == b->N + b->A, "encoder_forward_train: graph was prepared without the fake-agent rows");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "encoder_forward_train: dropout p must be in [0, 1)");
  TS_REQUIRE(!state_bf16(), "encoder_forward_train: the training tape is fp32; switch trajsde_state_storage(0)");
  const DropArg drop_aa = dropout ? make_drop(dropout->p, dropout->seed, 0) : no_drop();
  const DropArg drop_al = dropout ? make_drop(dropout->p, dropout->seed, 1) : no_drop();
  EncBwdWs w(b, g, ws, int64_t(1) << 60);                 // only the tape fields are touched here
  if (ws_bytes < w.tape_total) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_forward_train: workspace too small (trajsde_encoder_tape_bytes)");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  NoiseArg na{0, nullptr, nullptr};
  if (noise) { na.seed = noise->seed; na.z = noise->z; na.row_ids = noise->row_ids; na.seed_dev = noise->seed_dev; }
  if (int rc = encoder_tape(b, g, rot, blob_fwd, step_tab, na, w, drop_aa, drop_al, st)) return rc;
  const int N = b->N;
  TS_LAUNCH(k_ffn6, tile_grid((int64_t(N) + 15) / 16, 512, FfnL6::HALF * 4), 512, FfnL6::HALF * 4, st, blob_fwd + EncBlob::AL_FFN6, w.al_x1, w.al_xn2,
            int64_t(N), local_embed, drop_al);
  TS_HIP(hipMemsetAsync(diff_pick, 0, size_t(2) * b->A * 64 * sizeof(float), st));
  k_pick_diff<<<g->Nt, 64, 0, st>>>(w.GS, g->eos_idx, g->pick_slot, g->Nt, diff_pick);
  TS_LAUNCH_CHECK("k_pick_diff");
  return TRAJSDE_OK;
}

int trajsde_encoder_backward(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd,
                             const float* blob_bwd, const float* step_tab /*HOST [H,8]*/, const float* step_tab_dev,
                             const trajsde_noise* noise, const float* d_local, float diff_weight, void* ws, int64_t ws_bytes,
                             float* diff_loss, float* const* grads, int n_grads, float* d_latent, float* d_aa_out,
                             const trajsde_dropout* dropout, int tape_valid, void* scratch, int64_t scratch_bytes, void* stream_) {
  TS_REQUIRE(b && g && rot && blob_fwd && blob_bwd && step_tab && step_tab_dev && d_local && ws && diff_loss && grads,
             "encoder_backward: null pointer");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "encoder_backward: dropout p must be in [0, 1)");
  TS_REQUIRE(!state_bf16(), "encoder_backward: the backward pass keeps its tape in fp32; switch trajsde_state_storage(0) for training");
  const DropArg drop_aa = dropout ? make_drop(dropout->p, dropout->seed, 0) : no_drop();     // block ids of dropout.hpp
  const DropArg drop_al = dropout ? make_drop(dropout->p, dropout->seed, 1) : no_drop();
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_backward: graph not compacted (call trajsde_graph_compact)");
  TS_REQUIRE(g->exact, "encoder_backward: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(b->A > 0 && g->Nt == b->N + b->A, "encoder_backward: graph was prepared without the fake-agent rows");
  const std::vector<std::string> names = stage_param_names(TRAJSDE_STAGE_ENCODER_BWD, 0, 0);
  TS_REQUIRE(n_grads == int(names.size()), "encoder_backward: gradient count does not match trajsde_param_count(ENCODER_BWD)");
  GradTable G;
  for (int i = 0; i < n_grads; ++i) {
    TS_REQUIRE(grads[i] != nullptr, "encoder_backward: null gradient buffer " + names[i]);
    G.slot[names[i]] = grads[i];
  }
  EncBwdWs w(b, g, ws, ws_bytes, scratch, scratch_bytes);
  if (!w.ok) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_backward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int N = b->N, Nt = g->Nt, H = b->H, A = b->A;
  const int64_t R = int64_t(H) * Nt, Ela = g->E_la;
  const int64_t rtiles = (int64_t(Nt) + 15) / 16;
  NoiseArg na{0, nullptr, nullptr};
  if (noise) { na.seed = noise->seed; na.z = noise->z; na.row_ids = noise->row_ids; na.seed_dev = noise->seed_dev; }
  const WgradCtx wc{st, w.part, w.cs, step_tab_dev, w.parts};
  DeferredSums sums(st, w.part, w.cs, w.parts, step_tab_dev, w.varena, VPART_ARENA_SLABS * VPART_FLOATS);   // this call's reductions: at the end
  using BB = EncBwdBlob;

  if (!tape_valid)
    if (int rc = encoder_tape(b, g, rot, blob_fwd, step_tab, na, w, drop_aa, drop_al, st)) return rc;

  // ================= backward =================
  // ---- ALEncoder: d local_embed -> d latent
  {
    AttnChain c{blob_bwd + BB::AL_NODE, blob_bwd + BB::AL_PROJ, blob_fwd + EncBlob::AL_ATTN, blob_bwd + BB::AL_EDGEEMB,
                g->la_geom, w.al_q, w.al_emb, w.al_stats, w.lat, g->la_dst, g->la_segptr,
                NodeBlockTape{w.al_agg, w.al_xn, w.al_x1, w.al_xn2}, int64_t(N), Ela, "al_encoder", "lane_embed"};
    c.drop = drop_al;
    c.img_kvt = blob_bwd + BB::AL_EDGEKV + EdgeKvBwdL::WKT;
    if (int rc = run_attn_chain(c, d_local, w, wc, G, w.DLAT, st)) return rc;
    if (d_latent) TS_HIP(hipMemcpyAsync(d_latent, w.DLAT, size_t(N) * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  // ---- DiffBCE on the picked diffusion values
  TS_LAUNCH(k_diffbce, 1, 1024, 0, st, w.GS, g->eos_idx, g->pick_slot, Nt, A, diff_weight, w.DLDG, w.scal);
  TS_HIP(hipMemcpyAsync(diff_loss, w.scal, sizeof(float), hipMemcpyDeviceToDevice, st));
  // ---- recurrence, last iteration first
  {
    RecurTab tab;
    for (int idx = 0; idx < H; ++idx) {
      const float* e = step_tab + 8 * idx;
      tab.v[idx][0] = e[1]; tab.v[idx][1] = e[2]; tab.v[idx][2] = e[3]; tab.v[idx][3] = e[4];
    }
    const RecurBwdArgs rb{blob_bwd + BB::GRU, blob_bwd + BB::SDE, Nt, N, H, b->TT, na, b->padding_mask, g->nus_mask, g->orig, g->eos_idx,
                          w.DLAT, w.DLDG, w.HODE, w.U1, w.R1, w.UU, w.RR, w.N1, w.NW, w.H1, w.H2, w.G1, w.G2, w.GS,
                          w.DNW, w.DN1P, w.DUP, w.DRP, w.DU1, w.DR1, w.DHO, w.DAA, w.DF, w.DH2, w.DH1, w.DG2N, w.DG1N, w.DG2A, w.DG1A,
                          w.DGPN, w.DGPA, w.dhA};
    const int lds_r = (GruBwdL::SIZE > EncSdeBwdL::SIZE ? GruBwdL::SIZE : EncSdeBwdL::SIZE) * 4;
    // the cooperative form (recur.hip k_enc_recur_bwd_coop: the transposed matrices in registers, no staging) where it applies
    static const bool legacy_bwd = []() { const char* e = getenv("TRAJSDE_RECUR_LEGACY"); return e && atoi(e) != 0; }();
    // Row tiles per workgroup: unlike the forward kernel, whose three interleaved tiles hide each other's latencies, this one slows
    // down more than in proportion (measured per pass over 256 workgroups at 21 iterations: 0.146 / 0.27 / 0.70 / 1.26 ms with 1 / 2 /
    // 3 / 4 tiles: 404 / 508 registers, then 127 / 279 spilled, and a loop body that outgrows the instruction cache), so it runs one
    // or two tiles and as many rounds of 256 workgroups as that takes -- 515 tiles: 3 rounds of one, 0.44 ms (legacy kernel 0.83)
    static const int force_tw = []() { const char* e = getenv("TRAJSDE_RECUR_BWD_TW"); return e ? atoi(e) : 0; }();      // experiments
    const int64_t rounds1 = (rtiles + 255) / 256, rounds2 = (rtiles + 511) / 512;
    const int tiles_per_wg = force_tw > 0 ? force_tw : (rounds2 * 185 < rounds1 * 100 ? 2 : 1);
    if (TSDE_SPLIT_H3 && !legacy_bwd && tiles_per_wg <= COOP_TMAX && H <= 32) {
      RecurBwdCoopArgs ca{};
      ca.gru_t = blob_bwd + BB::GRU; ca.sde_t = blob_bwd + BB::SDE;
      ca.Nt = Nt; ca.N = N; ca.H = H; ca.TT = b->TT; ca.na = na;
      ca.pad = b->padding_mask; ca.nus = g->nus_mask; ca.orig = g->orig; ca.eos = g->eos_idx;
      ca.dlat = w.DLAT; ca.DLDG = w.DLDG;
      ca.tp = RecurTape{w.HIN, w.H1, w.H2, w.G1, w.G2, w.GS, w.HODE, w.XS, w.U1, w.R1, w.UU, w.RR, w.RH, w.N1, w.NW};
      ca.DNW = w.DNW; ca.DN1P = w.DN1P; ca.DUP = w.DUP; ca.DRP = w.DRP; ca.DU1 = w.DU1; ca.DR1 = w.DR1; ca.DAA = w.DAA;
      ca.DF = w.DF; ca.DH2 = w.DH2; ca.DH1 = w.DH1; ca.DG2N = w.DG2N; ca.DG1N = w.DG1N; ca.DG2A = w.DG2A; ca.DG1A = w.DG1A;
      ca.DGPN = w.DGPN; ca.DGPA = w.DGPA; ca.dh_out = w.dhA;
      for (int i = 0; i < H; ++i) { ca.dt[i] = step_tab[8 * i + 1]; ca.sq[i] = step_tab[8 * i + 2]; }
      const int grid = int((rtiles + tiles_per_wg - 1) / tiles_per_wg);
      const int lds = coop_bwd_lds_floats(tiles_per_wg) * 4;
      switch (tiles_per_wg) {
        case 1: TS_LAUNCH(k_enc_recur_bwd_coop<1>, grid, 256, lds, st, ca); break;
        case 2: TS_LAUNCH(k_enc_recur_bwd_coop<2>, grid, 256, lds, st, ca); break;
        case 3: TS_LAUNCH(k_enc_recur_bwd_coop<3>, grid, 256, lds, st, ca); break;
        default: TS_LAUNCH(k_enc_recur_bwd_coop<4>, grid, 256, lds, st, ca); break;
      }
    } else {
      TS_LAUNCH(k_enc_recur_bwd, tile_grid(rtiles, 256, lds_r), 256, lds_r, st, rb, tab);
    }
    const float* dh = w.dhA;
    // iteration 0 started from the learned initial state, broadcast to every row (ENC:78)
    if (int rc = run_colsum_tall(st, dh, Nt, 64, 64, G("hidden"), w.nb.vpart)) return rc;
    const std::string lf = "lsde_func.", gu = "gru_unit.";
    struct WG { const float* d; const float* a; const char* w; const char* bias; int ldw, col0, tc; };
    const WG jobs[] = {
        {w.DH1, w.HIN, "lsde_func.f_func.net.0.weight", "lsde_func.f_func.net.0.bias", 66, 0, 1},
        {w.DH2, w.H1, "lsde_func.f_func.net.2.weight", "lsde_func.f_func.net.2.bias", 64, 0, 0},
        {w.DF, w.H2, "lsde_func.f_func.net.4.weight", "lsde_func.f_func.net.4.bias", 64, 0, 0},
        {w.DG1N, w.HIN, "lsde_func.g_nus.net.0.weight", "lsde_func.g_nus.net.0.bias", 66, 0, 1},
        {w.DG2N, w.G1, "lsde_func.g_nus.net.2.weight", "lsde_func.g_nus.net.2.bias", 64, 0, 0},
        {w.DG1A, w.HIN, "lsde_func.g_argo.net.0.weight", "lsde_func.g_argo.net.0.bias", 66, 0, 1},
        {w.DG2A, w.G1, "lsde_func.g_argo.net.2.weight", "lsde_func.g_argo.net.2.bias", 64, 0, 0},
        {w.DNW, w.N1, "gru_unit.new_state_net.2.weight", "gru_unit.new_state_net.2.bias", 64, 0, 0},
        {w.DN1P, w.XS, "gru_unit.new_state_net.0.weight", "gru_unit.new_state_net.0.bias", 128, 0, 0},
        {w.DN1P, w.RH, "gru_unit.new_state_net.0.weight", nullptr, 128, 64, 0},
        {w.DUP, w.U1, "gru_unit.update_gate.2.weight", "gru_unit.update_gate.2.bias", 64, 0, 0},
        {w.DRP, w.R1, "gru_unit.reset_gate.2.weight", "gru_unit.reset_gate.2.bias", 64, 0, 0},
        {w.DU1, w.HODE, "gru_unit.update_gate.0.weight", "gru_unit.update_gate.0.bias", 128, 0, 0},
        {w.DU1, w.XS, "gru_unit.update_gate.0.weight", nullptr, 128, 64, 0},
        {w.DR1, w.HODE, "gru_unit.reset_gate.0.weight", "gru_unit.reset_gate.0.bias", 128, 0, 0},
        {w.DR1, w.XS, "gru_unit.reset_gate.0.weight", nullptr, 128, 64, 0},
    };
    WgradBatch wb(wc, R, Nt);                                  // sixteen problems over the same (iteration, row) slabs
    for (const WG& j : jobs) {
      float* W = G(j.w);
      float* bias = j.bias ? G(j.bias) : nullptr;
      TS_REQUIRE(!G.missing, "encoder_backward: parameter table lacks " + G.missing_name);
      if (int rc = wb.add(j.d, 64, j.a, 64, W, j.ldw, j.col0, bias, j.tc)) return rc;
    }
    if (int rc = wb.flush()) return rc;
    // last diffusion layers (64 -> 1): d w4 = sum dgp * g2, d b4 = sum dgp, per net (rows of the other source carry 0)
    float *n4w = G("lsde_func.g_nus.net.4.weight"), *n4b = G("lsde_func.g_nus.net.4.bias");
    float *a4w = G("lsde_func.g_argo.net.4.weight"), *a4b = G("lsde_func.g_argo.net.4.bias");
    TS_REQUIRE(!G.missing, "encoder_backward: parameter table lacks " + G.missing_name);
    for (int net = 0; net < 2; ++net) {
      const int blocks = R >= 65536 ? 128 : 8;
      float* const vp = vpart_slab(w.nb.vpart, blocks, 128);
      TS_LAUNCH(k_rowscale_colsum, blocks, 1024, 0, st, w.G2, net == 0 ? w.DGPN : w.DGPA, R, vp);
      {
        ColsumBatch cb(st, blocks, 128);
        cb.add(vp, 64, net == 0 ? n4w : a4w);
        cb.add(vp + 64, 1, net == 0 ? n4b : a4b);
        if (int rc = cb.flush()) return rc;
      }
    }
  }
  if (d_aa_out) TS_HIP(hipMemcpyAsync(d_aa_out, w.DAA, size_t(R) * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (int rc = aa_encoder_backward(b, g, rot, blob_fwd, blob_bwd, w, wc, G, 8, st, drop_aa)) return rc;
  return sums.finish();
}

// ------------------------------------------------------------------ vanilla LocalEncoder backward (GENC:52-93)
namespace {
struct TrLayerTape { float *xn, *q, *k, *v, *o, *x1, *xn2, *out; };
struct GridBwdExtra {      // buffers of the vanilla encoder backward beyond EncBwdWs: the temporal tape and its gradients
  float *x0, *dxa, *dxb, *dq, *dk, *dv, *dO, *tout, *dtout, *H, *DH, *dx1;
  std::vector<TrLayerTape> tp;
  int64_t bytes;
  bool ok;
  GridBwdExtra(void* base, int64_t size, int64_t N, int nl) : tp(nl) {
    Carver ex(base, size);
    const int64_t RT = N * 22;
    x0 = ex.take<float>(RT * 64);
    for (auto& l : tp) {
      l.xn = ex.take<float>(RT * 64); l.q = ex.take<float>(RT * 64); l.k = ex.take<float>(RT * 64); l.v = ex.take<float>(RT * 64);
      l.o = ex.take<float>(RT * 64); l.x1 = ex.take<float>(RT * 64); l.xn2 = ex.take<float>(RT * 64); l.out = ex.take<float>(RT * 64);
    }
    float** slabs[] = {&dxa, &dxb, &dq, &dk, &dv, &dO, &dx1};
    for (float** p : slabs) *p = ex.take<float>(RT * 64);
    tout = ex.take<float>(N * 64);
    dtout = ex.take<float>(N * 64);
    H = ex.take<float>(RT * 256);           // the temporal rows (22 N) outnumber the snapshot rows EncBwdWs sizes its node scratch for
    DH = ex.take<float>(RT * 256);
    bytes = ex.off + 256;
    ok = ex.ok;
  }
};
}  // namespace

int64_t trajsde_encoder_grid_backward_ws_bytes(const trajsde_batch* b, const trajsde_graph* g, int num_temporal_layers) {
  if (!b || !g || num_temporal_layers < 1 || num_temporal_layers > 16) return -1;
  EncBwdWs w(b, g, nullptr, 0);
  return align_up(w.total, 256) + GridBwdExtra(nullptr, 0, b->N, num_temporal_layers).bytes;
}

int trajsde_encoder_grid_backward(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd,
                                  const float* blob_bwd, int num_heads, int num_temporal_layers, const float* d_local, void* ws,
                                  int64_t ws_bytes, float* const* grads, int n_grads, void* stream_) {
  return trajsde_encoder_grid_backward_train(b, g, rot, blob_fwd, blob_bwd, num_heads, num_temporal_layers, d_local, ws, ws_bytes, grads, n_grads,
                                             nullptr, stream_);
}

// the same backward for a forward that ran with dropout (trajsde_encoder_grid_forward_train, the same `dropout`): the masks are
// regenerated from their counters in the recomputed forward and in every backward kernel that needs them
int trajsde_encoder_grid_backward_train(const trajsde_batch* b, const trajsde_graph* g, const float* rot, const float* blob_fwd,
                                        const float* blob_bwd, int num_heads, int num_temporal_layers, const float* d_local, void* ws,
                                        int64_t ws_bytes, float* const* grads, int n_grads, const trajsde_dropout* dropout, void* stream_) {
  TS_REQUIRE(b && g && rot && blob_fwd && blob_bwd && d_local && ws && grads, "encoder_grid_backward: null pointer");
  TS_REQUIRE(!dropout || (dropout->p >= 0.f && dropout->p < 1.f), "encoder_grid_backward: dropout p must be in [0, 1)");
  const bool dropping = dropout && dropout->p > 0.f;
  auto drop_of = [&](int block) { return dropping ? make_drop(dropout->p, dropout->seed, block) : no_drop(); };
  const DropArg drop_aa = drop_of(0), drop_al = drop_of(1);
  TS_REQUIRE(g->aa_dst && g->la_dst && g->orig, "encoder_grid_backward: graph not compacted");
  TS_REQUIRE(g->exact, "encoder_grid_backward: needs exact list lengths (trajsde_graph_prepare, not _async)");
  TS_REQUIRE(b->A == 0 && g->Nt == b->N && b->H == 21, "encoder_grid_backward: graph with A = 0 and 21 history steps expected");
  TS_REQUIRE(num_heads == 8 || num_heads == 4, "encoder_grid_backward: num_heads must be 8 or 4");
  const int nl = num_temporal_layers;
  TS_REQUIRE(nl >= 1 && nl <= 16, "encoder_grid_backward: 1..16 temporal layers");
  const std::vector<std::string> names = stage_param_names(TRAJSDE_STAGE_ENCODER_GRID_BWD, nl, 0);
  TS_REQUIRE(n_grads == int(names.size()), "encoder_grid_backward: gradient count does not match trajsde_param_count(ENCODER_GRID_BWD)");
  GradTable G;
  for (int i = 0; i < n_grads; ++i) {
    TS_REQUIRE(grads[i] != nullptr, "encoder_grid_backward: null gradient buffer " + names[i]);
    G.slot[names[i]] = grads[i];
  }
  if (ws_bytes < trajsde_encoder_grid_backward_ws_bytes(b, g, nl)) return fail(TRAJSDE_ERR_WORKSPACE, "encoder_grid_backward: workspace too small");
  EncBwdWs w(b, g, ws, ws_bytes);
  GridBwdExtra ex(static_cast<char*>(ws) + align_up(w.total, 256), ws_bytes - align_up(w.total, 256), b->N, nl);
  TS_REQUIRE(w.ok && ex.ok, "encoder_grid_backward: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int N = b->N, H = b->H;
  const int64_t R = int64_t(H) * N, RT = int64_t(N) * 22, Eaa = g->E_aa, Ela = g->E_la, rtiles = (RT + 15) / 16;
  std::vector<TrLayerTape>& tp = ex.tp;
  float *x0 = ex.x0, *dxa = ex.dxa, *dxb = ex.dxb, *dq = ex.dq, *dk = ex.dk, *dv = ex.dv, *dO = ex.dO, *tout = ex.tout, *dtout = ex.dtout;
  const WgradCtx wc{st, w.part, w.cs, nullptr, w.parts};
  using FB = EncBlob;
  using BB = EncBwdBlob;
  // ================= forward recompute =================
  TS_LAUNCH(k_aa_center, tile_grid((R + 15) / 16, 512, AaCenterL::SIZE * 4), 512, AaCenterL::SIZE * 4, st, blob_fwd + FB::AA_CENTER, b->x,
            g->x_fake, rot, b->bos_mask, g->orig, N, N, H, w.center, w.cn, w.q);
  if (int rc = edge_attention_tape("k_edge_kv[aa]+emb", blob_fwd + FB::AA_EDGE6F, blob_fwd + FB::AA_EDGE6, blob_fwd + FB::AA_ATTN, g->aa_geom, g->aa_dst, Eaa, g->aa_segptr, w.q, R,
                                   w.emb, w.stats, w.agg, w.rec, num_heads, drop_aa, st))
    return rc;
  TS_LAUNCH(k_node_update<true>, tile_grid((R + 15) / 16, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, blob_fwd + FB::AA_UPD6, w.agg, w.cn,
            w.center, R, w.x1, w.xn2, drop_aa, no_merge());
  TS_LAUNCH(k_ffn6, tile_grid((R + 15) / 16, 512, FfnL6::HALF * 4), 512, FfnL6::HALF * 4, st, blob_fwd + FB::AA_FFN6, w.x1, w.xn2, R, w.aa_out, drop_aa);
  TS_LAUNCH(k_tr_prep, cdiv(RT * 64, 256), 256, 0, st, w.aa_out, b->padding_mask, blob_fwd + EncGridBlob::TOK, N, b->TT, x0);
  const float* x = x0;
  for (int l = 0; l < nl; ++l) {
    const float* lb = blob_fwd + EncGridBlob::layer(l);
    TS_LAUNCH(k_node_proj<3>, tile_grid(rtiles, 512, NodeProjL<3>::SIZE * 4), 512, NodeProjL<3>::SIZE * 4, st, lb + TrLayerL::QKV, x, RT,
              tp[l].xn, tp[l].q, tp[l].k, tp[l].v);
    const DropArg dl = drop_of(DROP_TEMPORAL_BLOCK0 + l);
    if (int rc = launch_tr_attention(num_heads, tp[l].q, tp[l].k, tp[l].v, N, tp[l].o, dl, st)) return rc;
    TS_LAUNCH(k_tr_outproj, tile_grid(rtiles, 512, TrOutL::SIZE * 4), 512, TrOutL::SIZE * 4, st, lb + TrLayerL::OUT, tp[l].o, x, RT, tp[l].x1,
              tp[l].xn2, dl);
    TS_LAUNCH(k_ffn, tile_grid(rtiles, 512, FfnL::SIZE * 4), 512, FfnL::SIZE * 4, st, lb + TrLayerL::FFN, tp[l].x1, tp[l].xn2, RT, tp[l].out, dl, 0);
    x = tp[l].out;
  }
  TS_LAUNCH(k_tr_final, tile_grid((int64_t(N) + 15) / 16, 256, 0), 256, 0, st, blob_fwd + EncGridBlob::norm(nl), x, N, tout);
  TS_LAUNCH(k_node_proj<1>, tile_grid((int64_t(N) + 15) / 16, 512, NodeProjL<1>::SIZE * 4), 512, NodeProjL<1>::SIZE * 4, st,
            blob_fwd + FB::AL_Q, tout, int64_t(N), w.al_xn, w.al_q, nullptr, nullptr);
  if (int rc = edge_attention_tape("k_edge_kv[al]+emb", blob_fwd + FB::AL_EDGE6F, blob_fwd + FB::AL_EDGE6, blob_fwd + FB::AL_ATTN, g->la_geom, g->la_dst, Ela, g->la_segptr, w.al_q,
                                   int64_t(N), w.al_emb, w.al_stats, w.al_agg, w.rec, num_heads, drop_al, st))
    return rc;
  TS_LAUNCH(k_node_update<true>, tile_grid((int64_t(N) + 15) / 16, 512, UpdL6::SIZE * 4), 512, UpdL6::SIZE * 4, st, blob_fwd + FB::AL_UPD6,
            w.al_agg, w.al_xn, tout, int64_t(N), w.al_x1, w.al_xn2, drop_al, no_merge());
  // ================= backward =================
  {
    AttnChain c{blob_bwd + BB::AL_NODE, blob_bwd + BB::AL_PROJ, blob_fwd + EncBlob::AL_ATTN, blob_bwd + BB::AL_EDGEEMB,
                g->la_geom, w.al_q, w.al_emb, w.al_stats, tout, g->la_dst, g->la_segptr,
                NodeBlockTape{w.al_agg, w.al_xn, w.al_x1, w.al_xn2}, int64_t(N), Ela, "al_encoder", "lane_embed", num_heads};
    c.img_kvt = blob_bwd + BB::AL_EDGEKV + EdgeKvBwdL::WKT;
    c.drop = drop_al;
    if (int rc = run_attn_chain(c, d_local, w, wc, G, dtout, st)) return rc;
  }
  const std::string te = "temporal_encoder.";
  float* dcur = dxa;
  float* dnext = dxb;
  TS_HIP(hipMemsetAsync(dcur, 0, size_t(RT) * 64 * sizeof(float), st));
  {
    const int gf = vec_grid((int64_t(N) + 15) / 16, 256, 0);
    float *ng = G(te + "transformer_encoder.norm.weight"), *nb = G(te + "transformer_encoder.norm.bias");
    TS_REQUIRE(!G.missing, "encoder_grid_backward: parameter table lacks " + G.missing_name);
    TS_LAUNCH(k_tr_final_bwd, gf, 256, 0, st, blob_bwd + EncGridBwdBlob::norm(nl), x, dtout, N, dcur, w.nb.vpart);
    {
      ColsumBatch cb(st, gf * 4, 128);
      cb.add(w.nb.vpart, 64, ng);
      cb.add(w.nb.vpart + 64, 64, nb);
      if (int rc = cb.flush()) return rc;
    }
  }
  NodeBlockScratch sc = w.nb;
  sc.H = ex.H;
  sc.DH = ex.DH;
  sc.dx1 = ex.dx1;
  for (int l = nl - 1; l >= 0; --l) {
    const std::string p = te + "transformer_encoder.layers." + std::to_string(l);
    const float* lb = blob_bwd + EncGridBwdBlob::layer(l);
    const float* x_in = l == 0 ? x0 : tp[l - 1].out;
    NodeBlockGrads gr{};
    gr.w1 = G(p + ".linear1.weight"); gr.b1 = G(p + ".linear1.bias"); gr.w2 = G(p + ".linear2.weight"); gr.b2 = G(p + ".linear2.bias");
    gr.n2g = G(p + ".norm2.weight"); gr.n2b = G(p + ".norm2.bias");
    float *wo = G(p + ".self_attn.out_proj.weight"), *bo = G(p + ".self_attn.out_proj.bias");
    float *wi = G(p + ".self_attn.in_proj_weight"), *bi = G(p + ".self_attn.in_proj_bias");
    float *n1g = G(p + ".norm1.weight"), *n1b = G(p + ".norm1.bias");
    TS_REQUIRE(!G.missing, "encoder_grid_backward: parameter table lacks " + G.missing_name);
    const DropArg dl = drop_of(DROP_TEMPORAL_BLOCK0 + l);
    if (int rc = ffn_block_backward(lb + TrLayerBwdL::FFN_A, lb + TrLayerBwdL::FFN_B, tp[l].xn2, tp[l].x1, dcur, RT, sc, wc, gr, st, dl)) return rc;
    // dropout1: out_proj saw dx1 m, the residual dx1 (k_drop_rows; dq is free until the attention backward writes it)
    const float* dx1m = sc.dx1;
    if (dropping) {
      TS_LAUNCH(k_drop_rows, tile_grid(rtiles, 256, 0), 256, 0, st, sc.dx1, RT, dq, dl, int(DK_PROJ));
      dx1m = dq;
    }
    TS_LAUNCH(k_lin_t_acc, tile_grid(rtiles, 256, MAT64 * 4), 256, MAT64 * 4, st, lb + TrLayerBwdL::WOUT_T, dx1m, RT, dO, 0);
    if (int rc = run_wgrad(wc, dx1m, 64, tp[l].o, 64, RT, RT, wo, 64, 0, bo, 0)) return rc;
    if (num_heads == 4) {
      if (dropping) TS_LAUNCH_TAG("k_tr_attention_bwd<drop>", false, (k_tr_attention_bwd<4, true>), cdiv(N, 4), 256, 0, st, tp[l].q, tp[l].k, tp[l].v, dO, N, dq, dk, dv, dl);
      else TS_LAUNCH_TAG("k_tr_attention_bwd", false, (k_tr_attention_bwd<4, false>), cdiv(N, 4), 256, 0, st, tp[l].q, tp[l].k, tp[l].v, dO, N, dq, dk, dv, dl);
    } else {
      if (dropping) TS_LAUNCH_TAG("k_tr_attention_bwd<drop>", false, (k_tr_attention_bwd<8, true>), cdiv(N, 4), 256, 0, st, tp[l].q, tp[l].k, tp[l].v, dO, N, dq, dk, dv, dl);
      else TS_LAUNCH_TAG("k_tr_attention_bwd", false, (k_tr_attention_bwd<8, false>), cdiv(N, 4), 256, 0, st, tp[l].q, tp[l].k, tp[l].v, dO, N, dq, dk, dv, dl);
    }
    const int gp = vec_grid(rtiles, 256, ProjBwdL<3>::SIZE * 4);
    TS_LAUNCH(k_node_proj_bwd<3>, gp, 256, ProjBwdL<3>::SIZE * 4, st, lb + TrLayerBwdL::PROJ, x_in, sc.dx1, nullptr, dq, dk, dv, RT, dnext,
              nullptr, w.nb.vpart);
    {
      ColsumBatch cb(st, gp * 4, 128);
      cb.add(w.nb.vpart, 64, n1g);
      cb.add(w.nb.vpart + 64, 64, n1b);
      if (int rc = cb.flush()) return rc;
    }
    const float* dps[3] = {dq, dk, dv};
    WgradBatch wb(wc, RT, RT);
    for (int j = 0; j < 3; ++j)
      if (int rc = wb.add(dps[j], 64, tp[l].xn, 64, wi + int64_t(j) * MAT64, 64, 0, bi + 64 * j, 0)) return rc;
    if (int rc = wb.flush()) return rc;
    float* t = dcur; dcur = dnext; dnext = t;
  }
  {
    float *gpad = G(te + "padding_token"), *gcls = G(te + "cls_token"), *gpos = G(te + "pos_embed");
    TS_REQUIRE(!G.missing, "encoder_grid_backward: parameter table lacks " + G.missing_name);
    TS_LAUNCH(k_tr_tok_grad, 22, 1024, 0, st, dcur, b->padding_mask, N, b->TT, gpad, gcls, gpos);
    TS_LAUNCH(k_tr_prep_bwd, cdiv(R * 64, 256), 256, 0, st, dcur, b->padding_mask, N, b->TT, w.DAA);
  }
  return aa_encoder_backward(b, g, rot, blob_fwd, blob_bwd, w, wc, G, num_heads, st, drop_aa);
}

}  // extern "C"
