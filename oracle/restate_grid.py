"""CPU restatement of the vanilla-HiVT variant of the path: `PredictionModel.forward`
(models/model_base_mix.py:74-92) = LocalEncoder (models/encoders/enc_hivt_nusargo_grid.py: AAEncoder over 21 snapshots,
TemporalEncoder, ALEncoder) -> GlobalInteractor -> MLPDecoder (models/decoders/dec_hivt_nusargo_grid.py).
TEST INFRASTRUCTURE ONLY, same rules as restate.py; pinned by tests/golden_grid/*.npz (oracle/make_golden_grid.py).
Abbreviations: GENC = enc_hivt_nusargo_grid.py, GDEC = dec_hivt_nusargo_grid.py.
"""
import torch
import torch.nn.functional as F

from restate import (D, _lin, _ln, attention_aggregate, ff_block, gated_update, global_interactor, multiple_input_embedding,
                     proj_drop, rotate2, rotate_inputs, single_input_embedding)


def flat_cfg(cfg):
    e, a, d = cfg["encoder"]["kwargs"], cfg["aggregator"]["kwargs"], cfg["decoder"]["kwargs"]
    return dict(historical_steps=e["historical_steps"], local_radius=e["local_radius"], num_heads=e["num_heads"],
                num_temporal_layers=e["num_temporal_layers"], input_diff=e["input_diff"], num_modes=d["num_modes"],
                future_steps=d["future_steps"], min_scale=d["min_scale"], num_global_layers=a["num_layers"])


def temporal_masks(drop, layer, n_actors, heads, like):
    """the four train-mode dropout masks of TemporalEncoder layer `layer` in THIS file's layouts -- attention weights [N, h, S, S]
    (nn.MultiheadAttention's dropout on the softmax output), dropout1 [S, N, 64], the FFN's dropout [S, N, 256], dropout2 [S, N, 64]
    (GENC:262-283) -- from the host twin of csrc/dropout.hpp: block 16 + layer, node sites on the token row n * S + s"""
    from trajsde_amd import philox
    S, block = 22, philox.TEMPORAL_BLOCK0 + layer
    att = torch.from_numpy(philox.dropout_temporal_attn_mask(drop.seed, block, n_actors, heads, drop.p, S)).to(like.dtype)

    def rows(kind, width):
        m = philox.dropout_feature_mask(drop.seed, block, kind, n_actors * S, width, drop.p)           # row n * S + s
        return torch.from_numpy(m).to(like.dtype).view(n_actors, S, width).transpose(0, 1)              # -> [S, N, width]
    return att, rows(philox.DK_PROJ, D), rows(philox.DK_HIDDEN, 4 * D), rows(philox.DK_OUT, D)


def temporal_encoder(P, pre, x, padding_mask, heads, layers, drop=None):
    """TemporalEncoder.forward GENC:241-249 over nn.TransformerEncoder of pre-norm TemporalEncoderLayers (GENC:258-292),
    causal mask GENC:251-255; nn.MultiheadAttention semantics: q scaled by dh^-0.5, softmax over keys j <= i.
    `drop`: a restate.PhiloxDropout (train mode: the layers' four dropout sites) or None."""
    S, N, _ = x.shape                                                        # [21, N, 64]
    x = torch.where(padding_mask.t().unsqueeze(-1), P[pre + ".padding_token"], x)
    x = torch.cat((x, P[pre + ".cls_token"].expand(-1, N, -1)), 0) + P[pre + ".pos_embed"]
    S += 1
    dh = D // heads
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    for i in range(layers):
        l = f"{pre}.transformer_encoder.layers.{i}"
        xn = _ln(P, l + ".norm1", x)
        qkv = F.linear(xn, P[l + ".self_attn.in_proj_weight"], P[l + ".self_attn.in_proj_bias"])
        q, k, v = (t.reshape(S, N, heads, dh).permute(1, 2, 0, 3) for t in qkv.chunk(3, -1))    # [N, h, S, dh]
        att = (q * dh ** -0.5) @ k.transpose(-1, -2)
        att = att.masked_fill(~causal, float("-inf")).softmax(-1)
        m_att = m_proj = m_hid = m_out = None
        if drop is not None:
            m_att, m_proj, m_hid, m_out = temporal_masks(drop, i, N, heads, x)
            att = att * m_att
        o = (att @ v).permute(2, 0, 1, 3).reshape(S, N, D)
        sa = _lin(P, l + ".self_attn.out_proj", o)
        x = x + (sa * m_proj if drop is not None else sa)
        h = F.relu(_lin(P, l + ".linear1", _ln(P, l + ".norm2", x)))
        ff = _lin(P, l + ".linear2", h * m_hid if drop is not None else h)
        x = x + (ff * m_out if drop is not None else ff)
    return _ln(P, pre + ".transformer_encoder.norm", x)[-1]


def local_encoder_grid(P, cfg, batch, rot, drop=None, inter=None):
    """LocalEncoder.forward GENC:52-93.  `drop`: a restate.PhiloxDropout (train mode) or None; `inter`: dict that receives the edge
    lists in the reference's order (the fixture generator lays the injected masks out with them)"""
    pre = "encoder"
    H, radius, heads = cfg["historical_steps"], cfg["local_radius"], cfg["num_heads"]
    x, pos, pad = batch["x"], batch["positions"], batch["padding_mask"]
    N = x.shape[0]
    valid = ~pad[:, :H]
    src, dst = batch["edge_index"]
    srcs, dsts, attrs = [], [], []
    for t in range(H):                                                      # GENC:57-67
        keep = valid[src, t] & valid[dst, t]
        s_t, d_t = src[keep], dst[keep]
        attr = pos[s_t, t] - pos[d_t, t]
        near = torch.norm(attr, p=2, dim=-1) < radius
        srcs.append(s_t[near] + t * N)
        dsts.append(d_t[near] + t * N)
        attrs.append(attr[near])
    e_src, e_dst, e_attr = torch.cat(srcs), torch.cat(dsts), torch.cat(attrs)
    a = pre + ".aa_encoder"                                                 # AAEncoder.forward GENC:135-166
    xt = x.transpose(0, 1).reshape(H * N, 2)
    rot_rep = rot.repeat(H, 1, 1)
    center = single_input_embedding(P, a + ".center_embed", rotate2(xt, rot_rep))
    if cfg["input_diff"]:
        center = torch.where(batch["bos_mask"].t().reshape(H * N).unsqueeze(-1), P[a + ".bos_token"].repeat_interleave(N, 0), center)
    cn = _ln(P, a + ".norm1", center)
    r_e = rot_rep[e_dst]
    nbr = multiple_input_embedding(P, a + ".nbr_embed", [rotate2(xt[e_src], r_e), rotate2(e_attr, r_e)])
    agg = attention_aggregate(_lin(P, a + ".lin_q", cn), _lin(P, a + ".lin_k", nbr), _lin(P, a + ".lin_v", nbr), e_dst, H * N, heads,
                              attn_keep=drop.attn(0, e_src, e_dst, heads, cn) if drop is not None else None)
    center = center + proj_drop(_lin(P, a + ".out_proj", gated_update(P, a, agg, cn)), drop, 0)
    center = center + ff_block(P, a, _ln(P, a + ".norm2", center), drop, 0)
    out = temporal_encoder(P, pre + ".temporal_encoder", center.view(H, N, D), pad[:, :H], heads, cfg["num_temporal_layers"], drop)
    l = pre + ".al_encoder"                                                 # GENC:80-93 + ALEncoder
    la, lav = batch["lane_actor_index"], batch["lane_actor_vectors"]
    near = torch.norm(lav, p=2, dim=-1) < radius
    l_src, l_dst, lav = la[0][near], la[1][near], lav[near]
    lane_len = (1 - batch["lane_paddings"]).sum(-1)
    lp = batch["lane_positions"]
    ar = torch.arange(lp.size(0))
    lane_feat = lp[ar, (lane_len - 1).long()] - lp[ar, 0]
    xn = _ln(P, l + ".norm1", out)
    r_e = rot[l_dst]
    lane = multiple_input_embedding(P, l + ".lane_embed", [rotate2(lane_feat[l_src], r_e), rotate2(lav, r_e)])
    agg = attention_aggregate(_lin(P, l + ".lin_q", xn), _lin(P, l + ".lin_k", lane), _lin(P, l + ".lin_v", lane), l_dst, N, heads,
                              attn_keep=drop.attn(1, l_src, l_dst, heads, xn) if drop is not None else None)
    out = out + proj_drop(_lin(P, l + ".out_proj", gated_update(P, l, agg, xn)), drop, 1)
    if inter is not None:
        inter.update(aa_edge_list=(e_src, e_dst), al_edge_list=(l_src, l_dst))
    return out + ff_block(P, l, _ln(P, l + ".norm2", out), drop, 1)


def mlp_decoder(P, cfg, batch, local_embed, global_embed):
    """MLPDecoder.forward GDEC:47-63"""
    pre = "decoder"
    K, T = cfg["num_modes"], cfg["future_steps"]
    N = local_embed.shape[0]
    loc_exp = local_embed.expand(K, N, D)
    h = F.relu(_ln(P, pre + ".pi.1", _lin(P, pre + ".pi.0", torch.cat((loc_exp, global_embed), -1))))
    pi = _lin(P, pre + ".pi.6", F.relu(_ln(P, pre + ".pi.4", _lin(P, pre + ".pi.3", h)))).squeeze(-1).t()
    out = F.relu(_ln(P, pre + ".aggr_embed.1", _lin(P, pre + ".aggr_embed.0", torch.cat((global_embed, loc_exp), -1))))
    loc = _lin(P, pre + ".loc.3", F.relu(_ln(P, pre + ".loc.1", _lin(P, pre + ".loc.0", out)))).view(K, N, T, 2)
    if pre + ".scale.0.weight" not in P:                                        # GDEC:58-59 (`uncertain: False`): no scale head
        return {"loc": loc, "pi": pi, "reg_mask": ~batch["padding_mask"][:, -T:]}
    sc = _lin(P, pre + ".scale.3", F.relu(_ln(P, pre + ".scale.1", _lin(P, pre + ".scale.0", out))))
    sc = F.elu(sc, alpha=1.0).view(K, N, T, 2) + 1.0 + cfg["min_scale"]
    return {"loc": torch.cat((loc, sc), -1), "pi": pi, "reg_mask": ~batch["padding_mask"][:, -T:]}


@torch.no_grad()
def forward(P, cfg, batch, want_intermediates=False, drop=None):
    """PredictionModel.forward, models/model_base_mix.py:74-92 (eval mode; `drop`: a restate.PhiloxDropout for train mode)"""
    c = flat_cfg(cfg) if "encoder" in cfg else cfg
    rot, y_rot = rotate_inputs(batch)
    inter = {} if want_intermediates else None
    local = local_encoder_grid(P, c, batch, rot, drop, inter)
    glob = global_interactor(P, dict(c, historical_steps=c["historical_steps"]), batch, rot, local, inter, drop)
    out = mlp_decoder(P, c, batch, local, glob)
    out.update(rotate_mat=rot, y=y_rot)
    if want_intermediates:
        out.update(inter, local_embed=local, global_embed=glob)
    return out
