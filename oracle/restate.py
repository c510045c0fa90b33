"""CPU restatement of TrajSDE's forward hot path -- TEST INFRASTRUCTURE ONLY.

Plain torch fp32 on the host.  This file is the *checker*: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product path (trajsde_amd/) never does.

Parity pin: validated <= 1e-5 against golden vectors produced by running the reference's own modules
(imported from /root/reference over oracle/shims, with injected noise) -- see oracle/make_golden.py and
tests/test_oracle_golden.py.  The third-party arithmetic underneath the reference (torchsde 0.2.5,
torch-geometric 2.2.0) is absent from /root/reference and from this image; its published semantics
are restated in oracle/shims, and the reference holds no tests for it, so at that
boundary parity is "unpinned by the reference" and pinned only by our own known-answer tests.

Every function cites the reference lines it follows.  Abbreviations:
  MODEL = models/model_base_mix_sde.py      ENC = models/encoders/enc_hivt_nusargo_sde_sep2.py
  AGG   = models/aggregators/agg_hivt.py    DEC = models/decoders/dec_hivt_nusargo_sde.py
  EMB   = models/utils/embedding.py         ODEU = models/utils/ode_utils.py
  SDEINT = models/utils/sdeint.py           UTIL = models/utils/util.py
Weights come as a flat {state_dict key: tensor} dict with the reference's key names.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

D = 64


# ----------------------------------------------------------------------------- small pieces
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _ln(P, name, x):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-5)


def single_input_embedding(P, pre, x):
    """EMB:22-40: Linear-LN-ReLU-Linear-LN-ReLU-Linear-LN."""
    h = F.relu(_ln(P, pre + ".embed.1", _lin(P, pre + ".embed.0", x)))
    h = F.relu(_ln(P, pre + ".embed.4", _lin(P, pre + ".embed.3", h)))
    return _ln(P, pre + ".embed.7", _lin(P, pre + ".embed.6", h))


def multiple_input_embedding(P, pre, inputs):
    """EMB:43-70: per-input Linear-LN-ReLU-Linear, summed, then LN-ReLU-Linear-LN."""
    acc = None
    for i, x in enumerate(inputs):
        b = f"{pre}.module_list.{i}"
        h = _lin(P, b + ".3", F.relu(_ln(P, b + ".1", _lin(P, b + ".0", x))))
        acc = h if acc is None else acc + h
    h = F.relu(_ln(P, pre + ".aggr_embed.0", acc))
    return _ln(P, pre + ".aggr_embed.3", _lin(P, pre + ".aggr_embed.2", h))


def segment_softmax(logits, dst, n_dst):
    """torch_geometric.utils.softmax semantics: exp(x-max_seg)/(sum_seg+1e-16)."""
    idx = dst.view(-1, 1).expand_as(logits)
    m = logits.new_full((n_dst, logits.shape[1]), float("-inf")).scatter_reduce(0, idx, logits, "amax")
    e = (logits - m.gather(0, idx)).exp()
    s = logits.new_zeros((n_dst, logits.shape[1])).scatter_add_(0, idx, e)
    return e / (s.gather(0, idx) + 1e-16)


class PhiloxDropout:
    """Train-mode dropout masks of the attention blocks from the host twin of csrc/dropout.hpp (trajsde_amd/philox.py): the
    masks the HIP kernels regenerate in-kernel.  `block`: 0 AAEncoder, 1 ALEncoder, 2 + i global layer i."""

    def __init__(self, seed, p):
        self.seed, self.p = int(seed), float(p)

    def attn(self, block, src, dst, heads, like):
        from trajsde_amd import philox
        rank = philox.segment_ranks(src.numpy(), dst.numpy())
        return torch.from_numpy(philox.dropout_attn_mask(self.seed, block, dst.numpy(), rank, heads, self.p)).to(like.dtype)

    def feat(self, block, kind, like):
        from trajsde_amd import philox
        return torch.from_numpy(philox.dropout_feature_mask(self.seed, block, kind, like.shape[0], like.shape[1], self.p)).to(like.dtype)


def attention_aggregate(q_dst, k_e, v_e, dst, n_dst, heads=8, attn_keep=None):
    """message + add-aggregate shared by ENC:586-593, ENC:765-772, AGG:108-117; `attn_keep` [E, heads]: attn_drop (ENC:592)
    as a mask of {0, 1/(1-p)} factors (train mode)."""
    dh = q_dst.shape[1] // heads
    q = q_dst.index_select(0, dst).view(-1, heads, dh)
    alpha = (q * k_e.view(-1, heads, dh)).sum(-1) / (dh ** 0.5)
    alpha = segment_softmax(alpha, dst, n_dst)
    if attn_keep is not None:
        alpha = alpha * attn_keep
    msg = (v_e.view(-1, heads, dh) * alpha.unsqueeze(-1)).reshape(-1, heads * dh)
    return msg.new_zeros((n_dst, heads * dh)).index_add_(0, dst, msg)


def gated_update(P, pre, agg, x_norm):
    """update(): ENC:595-600, ENC:774-780, AGG:119-124."""
    gate = torch.sigmoid(_lin(P, pre + ".lin_ih", agg) + _lin(P, pre + ".lin_hh", x_norm))
    return agg + gate * (_lin(P, pre + ".lin_self", x_norm) - agg)


def ff_block(P, pre, x, drop=None, block=0):
    """_ff_block: Linear(64,256)-ReLU-Dropout-Linear(256,64)-Dropout (ENC:529-533; the dropouts act in train mode only)."""
    h = F.relu(_lin(P, pre + ".mlp.0", x))
    if drop is not None:
        h = h * drop.feat(block, 2, h)
    o = _lin(P, pre + ".mlp.3", h)
    return o * drop.feat(block, 3, o) if drop is not None else o


def proj_drop(x, drop, block):
    """proj_drop(out_proj(...)) (ENC:611, ENC:794, AGG:132)"""
    return x * drop.feat(block, 1, x) if drop is not None else x


def rotate2(vec, rot):
    """row-vector times per-row 2x2 matrix, i.e. torch.bmm(v.unsqueeze(-2), R).squeeze(-2)."""
    return torch.stack((vec[:, 0] * rot[:, 0, 0] + vec[:, 1] * rot[:, 1, 0],
                        vec[:, 0] * rot[:, 0, 1] + vec[:, 1] * rot[:, 1, 1]), dim=-1)


def sde_time_mlp_in(y, sin_t, cos_t):
    """cat(y, sin t, cos t): ENC:395-397, DEC:124-126."""
    col = y.new_ones(y.shape[0], 1)
    return torch.cat((y, col * sin_t, col * cos_t), dim=-1)


def drift(P, pre, y, sin_t, cos_t):
    """FFunc: ENC:372-398 (sde_layers=2) == DEC:107-127: 66->64 tanh 64->64 tanh 64->64."""
    h = torch.tanh(_lin(P, pre + ".net.0", sde_time_mlp_in(y, sin_t, cos_t)))
    h = torch.tanh(_lin(P, pre + ".net.2", h))
    return _lin(P, pre + ".net.4", h)


def diffusion(P, pre, y, sin_t, cos_t):
    """GFunc: ENC:412-440 == DEC:141-158: sigmoid(66->64 tanh 64->64 tanh 64->1), one scalar per row."""
    h = torch.tanh(_lin(P, pre + ".net.0", sde_time_mlp_in(y, sin_t, cos_t)))
    h = torch.tanh(_lin(P, pre + ".net.2", h))
    return torch.sigmoid(_lin(P, pre + ".net.4", h))


def gru_unit(P, pre, h_cur, x, mask):
    """GRU_Unit.forward, ODEU:136-152."""
    yc = torch.cat((h_cur, x), -1)
    u = torch.sigmoid(_lin(P, pre + ".update_gate.2", torch.tanh(_lin(P, pre + ".update_gate.0", yc))))
    r = torch.sigmoid(_lin(P, pre + ".reset_gate.2", torch.tanh(_lin(P, pre + ".reset_gate.0", yc))))
    comb = torch.cat((x, r * h_cur), dim=1)
    new = _lin(P, pre + ".new_state_net.2", torch.tanh(_lin(P, pre + ".new_state_net.0", comb)))
    h_next = (1 - u) * new + u * h_cur
    m = mask.unsqueeze(-1)
    return m * h_next + ~m * h_cur


# ----------------------------------------------------------------------------- stages
def rotate_inputs(batch):
    """MODEL:75-85: rotate_mat[n] = [[cos,-sin],[sin,cos]]; y <- y @ R_n."""
    ang = batch["rotate_angles"]
    s, c = torch.sin(ang), torch.cos(ang)
    rot = torch.stack((torch.stack((c, -s), -1), torch.stack((s, c), -1)), -2)
    y = batch["y"] if "y" in batch else None
    y_rot = torch.bmm(y, rot) if y is not None else None
    return rot, y_rot


def local_encoder(P, cfg, batch, rot, noise, enc_sched, want_intermediates=False, drop=None):
    """LocalEncoderSDESepPara2.forward, ENC:66-202.  `drop`: a PhiloxDropout (train mode) or None (eval)."""
    pre = "encoder"
    H = cfg["historical_steps"]
    ref_time = cfg["ref_time"]
    radius = cfg["local_radius"]
    x, pos = batch["x"], batch["positions"]
    pad = batch["padding_mask"]
    agent_index = batch["agent_index"]
    N, A = x.shape[0], agent_index.shape[0]
    Nt = N + A
    inter = {}

    # ENC:68-71 lane feature = last valid point - first point
    lane_len = (1 - batch["lane_paddings"]).sum(-1)
    lp = batch["lane_positions"]
    ar = torch.arange(lp.size(0))
    lane_feat = lp[ar, (lane_len - 1).long()] - lp[ar, 0]

    # ENC:73-74,103 source mask, extended by the fake copies of the target agents
    nus_mask = torch.cat((batch["source"][batch["batch"]] == 0, batch["source"] == 0))

    # ENC:88-103 fake agents: x + 2*randn, duplicated incoming edges, copied per-actor rows
    ei = batch["edge_index"]
    to_agent = torch.isin(ei[1], agent_index)
    _, inv = torch.unique(ei[1][to_agent], return_inverse=True)
    ei_ext = torch.cat((ei, torch.stack((ei[0][to_agent], inv + N))), dim=-1)
    z_fake = noise.fake_agent((A, H, 2))
    x_ext = torch.cat((x, x[agent_index] + 2 * z_fake), 0)
    orig = torch.cat((torch.arange(N), agent_index))
    pad_ext = pad[orig]
    valid = ~pad_ext[:, :ref_time + 1]                      # [Nt, 21]
    pos_ext = pos[orig]
    bos_ext = batch["bos_mask"][orig]
    rot_ext = rot[orig]

    # ENC:107-121 per-step subgraph (both endpoints valid) + radius filter, 21 snapshots in one graph
    src, dst = ei_ext
    srcs, dsts, attrs = [], [], []
    for t in range(H):
        keep = valid[src, t] & valid[dst, t]
        s_t, d_t = src[keep], dst[keep]
        attr = pos_ext[s_t, t] - pos_ext[d_t, t]
        near = torch.norm(attr, p=2, dim=-1) < radius      # UTIL:83-92
        srcs.append(s_t[near] + t * Nt)
        dsts.append(d_t[near] + t * Nt)
        attrs.append(attr[near])
    e_src, e_dst, e_attr = torch.cat(srcs), torch.cat(dsts), torch.cat(attrs)
    inter["aa_edges"] = int(e_src.numel())
    if want_intermediates:
        inter["aa_edge_list"] = torch.stack((e_src, e_dst))    # snapshot-node ids t*Nt + n, the reference's order

    # AAEncoder.forward ENC:546-566 on the 21*Nt snapshot nodes (node id = t*Nt + n)
    a = pre + ".aa_encoder"
    xt = x_ext.transpose(0, 1).reshape(H * Nt, 2)           # [t, n] order
    rot_rep = rot_ext.repeat(H, 1, 1)
    center = single_input_embedding(P, a + ".center_embed", rotate2(xt, rot_rep))
    bos = bos_ext.t().reshape(H * Nt)
    tok = P[a + ".bos_token"].repeat_interleave(Nt, 0)
    center = torch.where(bos.unsqueeze(-1), tok, center)
    cn = _ln(P, a + ".norm1", center)
    r_e = rot_rep[e_dst]
    nbr = multiple_input_embedding(P, a + ".nbr_embed", [rotate2(xt[e_src], r_e), rotate2(e_attr, r_e)])
    agg = attention_aggregate(_lin(P, a + ".lin_q", cn), _lin(P, a + ".lin_k", nbr), _lin(P, a + ".lin_v", nbr),
                              e_dst, H * Nt, attn_keep=drop.attn(0, e_src, e_dst, 8, cn) if drop is not None else None)
    center = center + proj_drop(_lin(P, a + ".out_proj", gated_update(P, a, agg, cn)), drop, 0)
    center = center + ff_block(P, a, _ln(P, a + ".norm2", center), drop, 0)
    aa_out = center.view(H, Nt, D)
    if want_intermediates:
        inter["aa_out"] = aa_out

    # ENC:128-182 latent SDE + GRU recurrence, iteration idx consumes history step t = 20 - idx
    h = P[pre + ".hidden"].unsqueeze(0).repeat(Nt, 1)
    sel = torch.cat((agent_index, torch.arange(N, Nt)))     # ENC:101 new_agent_index
    latent, diffs = [], []
    lf = pre + ".lsde_func"
    for idx in range(H):
        t = H - 1 - idx
        s_t, c_t = float(enc_sched.sin_t0[idx]), float(enc_sched.cos_t0[idx])
        f = drift(P, lf + ".f_func", h, s_t, c_t)
        g = torch.empty(Nt, D)
        g[nus_mask] = diffusion(P, lf + ".g_nus", h[nus_mask], s_t, c_t).repeat(1, D)      # ENC:470-482
        g[~nus_mask] = diffusion(P, lf + ".g_argo", h[~nus_mask], s_t, c_t).repeat(1, D)
        dW = noise.encoder(idx, (Nt, D)) * float(enc_sched.sqrt_h[idx])
        # Euler_private.step SDEINT:477-485: y1 = y0 + f*dt + g*dW (one step per interval, App. D)
        h_ode = h + f * torch.tensor(enc_sched.dt[idx]) + g * dW
        h = gru_unit(P, pre + ".gru_unit", h_ode, aa_out[t], valid[:, t])
        latent.append(h)
        diffs.append(g[sel])
    latent = torch.stack(latent)[:, :N]
    diffs = torch.stack(diffs)
    if want_intermediates:
        inter["latent_ys"] = latent

    # ENC:187-196 pick the state of the iteration that processed each actor's first valid step
    eos = ref_time - torch.argmax(batch["bos_mask"].float(), dim=1)
    out = latent[eos, torch.arange(N)]
    diff_pick = diffs[eos[agent_index].repeat(2), torch.arange(2 * A)]
    diff_in, diff_out = torch.chunk(diff_pick, 2, 0)

    # ENC:198-200 + ALEncoder ENC:732-797 (bipartite lane -> actor)
    l = pre + ".al_encoder"
    la, lav = batch["lane_actor_index"], batch["lane_actor_vectors"]
    near = torch.norm(lav, p=2, dim=-1) < radius
    l_src, l_dst, lav = la[0][near], la[1][near], lav[near]
    if want_intermediates:
        inter["al_edge_list"] = torch.stack((l_src, l_dst))    # (lane, actor)
    xn = _ln(P, l + ".norm1", out)
    r_e = rot[l_dst]
    lane = multiple_input_embedding(P, l + ".lane_embed", [rotate2(lane_feat[l_src], r_e), rotate2(lav, r_e)])
    agg = attention_aggregate(_lin(P, l + ".lin_q", xn), _lin(P, l + ".lin_k", lane), _lin(P, l + ".lin_v", lane),
                              l_dst, N, attn_keep=drop.attn(1, l_src, l_dst, 8, xn) if drop is not None else None)
    out = out + proj_drop(_lin(P, l + ".out_proj", gated_update(P, l, agg, xn)), drop, 1)
    out = out + ff_block(P, l, _ln(P, l + ".norm2", out), drop, 1)
    return out, diff_in, diff_out, inter


def local_encoder_ood(P, cfg, batch, rot, noise, enc_sched, eval_iter=10):
    """LocalEncoderSDESepPara2.forward_ood, ENC:204-370: no fake agents, `eval_iter` stochastic recurrences from a
    zero initial state, per-actor std of the kept latent state, AL encoder on the mean."""
    pre = "encoder"
    H, ref_time, radius = cfg["historical_steps"], cfg["ref_time"], cfg["local_radius"]
    x, pos, pad = batch["x"], batch["positions"], batch["padding_mask"]
    N = x.shape[0]
    lane_len = (1 - batch["lane_paddings"]).sum(-1)
    lp = batch["lane_positions"]
    ar = torch.arange(lp.size(0))
    lane_feat = lp[ar, (lane_len - 1).long()] - lp[ar, 0]
    nus_mask = batch["source"][batch["batch"]] == 0                        # ENC:211-212
    valid = ~pad[:, :ref_time + 1]
    src, dst = batch["edge_index"]
    srcs, dsts, attrs = [], [], []
    for t in range(H):                                                      # ENC:226-236
        keep = valid[src, t] & valid[dst, t]
        s_t, d_t = src[keep], dst[keep]
        attr = pos[s_t, t] - pos[d_t, t]
        near = torch.norm(attr, p=2, dim=-1) < radius
        srcs.append(s_t[near] + t * N)
        dsts.append(d_t[near] + t * N)
        attrs.append(attr[near])
    e_src, e_dst, e_attr = torch.cat(srcs), torch.cat(dsts), torch.cat(attrs)
    a = pre + ".aa_encoder"
    xt = x.transpose(0, 1).reshape(H * N, 2)
    rot_rep = rot.repeat(H, 1, 1)
    center = single_input_embedding(P, a + ".center_embed", rotate2(xt, rot_rep))
    center = torch.where(batch["bos_mask"].t().reshape(H * N).unsqueeze(-1), P[a + ".bos_token"].repeat_interleave(N, 0), center)
    cn = _ln(P, a + ".norm1", center)
    r_e = rot_rep[e_dst]
    nbr = multiple_input_embedding(P, a + ".nbr_embed", [rotate2(xt[e_src], r_e), rotate2(e_attr, r_e)])
    agg = attention_aggregate(_lin(P, a + ".lin_q", cn), _lin(P, a + ".lin_k", nbr), _lin(P, a + ".lin_v", nbr), e_dst, H * N)
    center = center + _lin(P, a + ".out_proj", gated_update(P, a, agg, cn))
    center = center + ff_block(P, a, _ln(P, a + ".norm2", center))
    aa_out = center.view(H, N, D)
    eos = ref_time - torch.argmax(batch["bos_mask"].float(), dim=1)
    lf = pre + ".lsde_func"
    outs = []
    for j in range(eval_iter):                                              # ENC:255-309
        h = torch.zeros(N, D)
        latent = []
        for idx in range(H):
            t = H - 1 - idx
            s_t, c_t = float(enc_sched.sin_t0[idx]), float(enc_sched.cos_t0[idx])
            f = drift(P, lf + ".f_func", h, s_t, c_t)
            g = torch.empty(N, D)
            g[nus_mask] = diffusion(P, lf + ".g_nus", h[nus_mask], s_t, c_t).repeat(1, D)
            g[~nus_mask] = diffusion(P, lf + ".g_argo", h[~nus_mask], s_t, c_t).repeat(1, D)
            dW = noise.encoder(j * H + idx, (N, D)) * float(enc_sched.sqrt_h[idx])
            h_ode = h + f * torch.tensor(enc_sched.dt[idx]) + g * dW
            h = gru_unit(P, pre + ".gru_unit", h_ode, aa_out[t], valid[:, t])
            latent.append(h)
        outs.append(torch.stack(latent)[eos, torch.arange(N)])
    outs = torch.stack(outs)
    actors_std = outs.std(0).mean(-1)                                       # ENC:312
    out = outs.mean(0)
    l = pre + ".al_encoder"
    la, lav = batch["lane_actor_index"], batch["lane_actor_vectors"]
    near = torch.norm(lav, p=2, dim=-1) < radius
    l_src, l_dst, lav = la[0][near], la[1][near], lav[near]
    xn = _ln(P, l + ".norm1", out)
    r_e = rot[l_dst]
    lane = multiple_input_embedding(P, l + ".lane_embed", [rotate2(lane_feat[l_src], r_e), rotate2(lav, r_e)])
    agg = attention_aggregate(_lin(P, l + ".lin_q", xn), _lin(P, l + ".lin_k", lane), _lin(P, l + ".lin_v", lane), l_dst, N)
    out = out + _lin(P, l + ".out_proj", gated_update(P, l, agg, xn))
    out = out + ff_block(P, l, _ln(P, l + ".norm2", out))
    return out, actors_std


def global_interactor(P, cfg, batch, rot, local_embed, inter=None, drop=None):
    """GlobalInteractor.forward AGG:38-58 with GlobalInteractorLayer AGG:92-135.  `drop`: PhiloxDropout (train mode) or None."""
    pre = "aggregator"
    K = cfg["num_modes"]
    t_ref = cfg["historical_steps"] - 1
    valid = ~batch["padding_mask"][:, t_ref]
    src, dst = batch["edge_index"]
    keep = valid[src] & valid[dst]
    src, dst = src[keep], dst[keep]
    if inter is not None:
        inter["g_edge_list"] = torch.stack((src, dst))
    pos = batch["positions"][:, t_ref]
    rel_pos = rotate2(pos[src] - pos[dst], rot[dst])
    th = batch["rotate_angles"][src] - batch["rotate_angles"][dst]
    rel = multiple_input_embedding(P, pre + ".rel_embed", [rel_pos, torch.stack((torch.cos(th), torch.sin(th)), -1)])
    x = local_embed
    n = x.shape[0]
    for i in range(cfg["num_global_layers"]):
        g = f"{pre}.global_interactor_layers.{i}"
        xn = _ln(P, g + ".norm1", x)
        k_e = _lin(P, g + ".lin_k_node", xn)[src] + _lin(P, g + ".lin_k_edge", rel)
        v_e = _lin(P, g + ".lin_v_node", xn)[src] + _lin(P, g + ".lin_v_edge", rel)
        heads = cfg.get("num_heads", 8)
        agg = attention_aggregate(_lin(P, g + ".lin_q_node", xn), k_e, v_e, dst, n, heads=heads,
                                  attn_keep=drop.attn(2 + i, src, dst, heads, xn) if drop is not None else None)
        x = x + proj_drop(_lin(P, g + ".out_proj", gated_update(P, g, agg, xn)), drop, 2 + i)
        x = x + ff_block(P, g, _ln(P, g + ".norm2", x), drop, 2 + i)
    x = _ln(P, pre + ".norm", x)
    return _lin(P, pre + ".multihead_proj", x).view(n, K, D).transpose(0, 1)      # [K, N, 64]


def sde_decoder(P, cfg, batch, local_embed, global_embed, noise, dec_sched, want_intermediates=False):
    """SDEDecoder.forward DEC:77-105; the solve restates stock torchsde.sdeint
    over the float32 schedule table of App. D."""
    pre = "decoder"
    K, T = cfg["num_modes"], cfg["future_steps"]
    N = local_embed.shape[0]
    loc_exp = local_embed.expand(K, N, D)
    y = F.relu(_ln(P, pre + ".aggr_embed.1", _lin(P, pre + ".aggr_embed.0", torch.cat((global_embed, loc_exp), -1))))
    y = y.reshape(K * N, D)
    lf = pre + ".lsde_func"
    sol = []
    prev, o = y, 0
    for k in range(dec_sched.n_euler):
        s_t, c_t = float(dec_sched.sin_t0[k]), float(dec_sched.cos_t0[k])
        f = drift(P, lf + ".f_func", y, s_t, c_t)
        g = diffusion(P, lf + ".g_func", y, s_t, c_t).repeat(1, D)          # DEC:194
        dW = noise.decoder(k, (K * N, D)) * float(dec_sched.sqrt_h[k])
        prev = y
        y = y + f * torch.tensor(dec_sched.dt[k]) + g * dW
        while o < dec_sched.n_out and dec_sched.out_step[o] == k + 1:
            sol.append(torch.tensor(dec_sched.out_w0[o]) * prev + torch.tensor(dec_sched.out_w1[o]) * y)
            o += 1
    sol = torch.stack(sol).permute(1, 0, 2)                                     # [K*N, T, 64]
    pi = _lin(P, pre + ".pi.3", F.relu(_ln(P, pre + ".pi.1", _lin(P, pre + ".pi.0", torch.cat((loc_exp, global_embed), -1)))))
    pi = pi.squeeze(-1).t()
    loc = _lin(P, pre + ".decoder.3", F.relu(_ln(P, pre + ".decoder.1", _lin(P, pre + ".decoder.0", sol))))
    if pre + ".scale.0.weight" in P:                                            # DEC:96-99 (`uncertain: True`)
        sc = _lin(P, pre + ".scale.3", F.relu(_ln(P, pre + ".scale.1", _lin(P, pre + ".scale.0", sol))))
        sc = F.elu(sc, alpha=1.0) + 1.0 + cfg["min_scale"]
        loc = torch.cat((loc.view(K, N, T, 2), sc.view(K, N, T, 2)), -1)
    else:                                                                       # DEC:100-101: no scale head, loc [K, N, T, 2]
        loc = loc.view(K, N, T, 2)
    out = {"loc": loc, "pi": pi, "reg_mask": ~batch["padding_mask"][:, -T:]}
    if want_intermediates:
        out["y0"] = sol.new_tensor([])  # placeholder so keys are stable
        out["sol"] = sol
    return out


# ----------------------------------------------------------------------------- noise + driver
class InjectedNoise:
    """Standard normals handed over as tensors (golden fixtures / parity against the HIP path)."""

    def __init__(self, z_fake, z_enc, z_dec):
        self.z_fake, self.z_enc, self.z_dec = z_fake, z_enc, z_dec

    def fake_agent(self, shape):
        return self.z_fake.reshape(shape)

    def encoder(self, idx, shape):
        return self.z_enc[idx].reshape(shape)

    def decoder(self, k, shape):
        return self.z_dec[k].reshape(shape)


class PhiloxNoise:
    """The host twin of the in-kernel Philox stream (trajsde_amd/philox.py); rows are global ids."""

    def __init__(self, seed, enc_row_ids=None, dec_row_ids=None, fake_row_ids=None):
        self.seed, self.enc_rows, self.dec_rows, self.fake_rows = seed, enc_row_ids, dec_row_ids, fake_row_ids

    def _rows(self, given, n):
        import numpy as np
        return np.arange(n, dtype=np.uint32) if given is None else np.asarray(given, dtype=np.uint32)

    def fake_agent(self, shape):
        from trajsde_amd import philox
        A, H, two = shape
        z = philox.normals(self.seed, philox.STREAM_FAKE_AGENT, 0, self._rows(self.fake_rows, A), 64)
        return torch.from_numpy(z[:, :H * two].copy()).view(A, H, two)

    def encoder(self, idx, shape):
        from trajsde_amd import philox
        return torch.from_numpy(philox.normals(self.seed, philox.STREAM_ENCODER, idx, self._rows(self.enc_rows, shape[0]), shape[1]))

    def decoder(self, k, shape):
        from trajsde_amd import philox
        return torch.from_numpy(philox.normals(self.seed, philox.STREAM_DECODER, k, self._rows(self.dec_rows, shape[0]), shape[1]))


def flat_cfg(cfg):
    """Pick the handful of numbers the arithmetic needs out of the reference-style YAML dict."""
    e, a, d = cfg["encoder"]["kwargs"], cfg["aggregator"]["kwargs"], cfg["decoder"]["kwargs"]
    return dict(historical_steps=e["historical_steps"], ref_time=e["ref_time"], local_radius=e["local_radius"],
                max_past_t=e["max_past_t"], minimum_step=e["minimum_step"], num_modes=d["num_modes"],
                future_steps=d["future_steps"], max_fut_t=d["max_fut_t"], min_stepsize=d["min_stepsize"],
                min_scale=d["min_scale"], num_global_layers=a["num_layers"])


@torch.no_grad()
def forward(P: Dict[str, torch.Tensor], cfg: dict, batch, noise, want_intermediates: bool = False,
            schedules: Optional[tuple] = None, ood: bool = False, drop=None):
    """PredictionModelSDENet.forward, MODEL:74-102 (fp32, CPU).  `drop`: PhiloxDropout for train mode, None for eval."""
    from trajsde_amd.schedule import decoder_schedule, encoder_schedule
    c = flat_cfg(cfg) if "encoder" in cfg else cfg
    if schedules is None:
        schedules = (encoder_schedule(c["historical_steps"], c["max_past_t"], c["minimum_step"]),
                     decoder_schedule(c["future_steps"], c["max_fut_t"], c["min_stepsize"]))
    enc_sched, dec_sched = schedules
    rot, y_rot = rotate_inputs(batch)
    if ood:                                                                   # MODEL:89-90, 97-98
        local, stds = local_encoder_ood(P, c, batch, rot, noise, enc_sched)
        glob = global_interactor(P, c, batch, rot, local)
        out = sde_decoder(P, c, batch, local, glob, noise, dec_sched, want_intermediates)
        out.update(stds=stds, rotate_mat=rot, y=y_rot)
        if want_intermediates:
            out.update(local_embed=local, global_embed=glob)
        return out
    local, diff_in, diff_out, inter = local_encoder(P, c, batch, rot, noise, enc_sched, want_intermediates, drop)
    glob = global_interactor(P, c, batch, rot, local, inter if want_intermediates else None, drop)
    out = sde_decoder(P, c, batch, local, glob, noise, dec_sched, want_intermediates)
    out.update(diff_in=diff_in, diff_out=diff_out, label_in=torch.zeros_like(diff_in),
               label_out=torch.ones_like(diff_out), rotate_mat=rot, y=y_rot)
    if want_intermediates:
        out.update(local_embed=local, global_embed=glob, **inter)
    return out
