"""Shared helpers for the parity tests: golden fixture loading, model construction, oracle runs."""
import glob
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def our_cfg(num_modes, future_steps, max_fut_t, uncertain=True):
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model_specific"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps)
    cfg["aggregator"]["kwargs"]["num_modes"] = num_modes
    cfg["decoder"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps, max_fut_t=max_fut_t)
    if not uncertain:                                          # DEC:56: the decoder without its scale head
        cfg["decoder"]["kwargs"]["uncertain"] = False
    return cfg


def load_fixture(name):
    from trajsde_amd.data import TemporalData
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    batch = TemporalData(**{k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")})
    batch["num_nodes"] = batch["x"].shape[0]
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta.")}
    out = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")}
    mid = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mid.")}
    return batch, meta, out, mid


def build_model(meta_or_K, T=None, max_t=None, init_seed=0):
    from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet
    uncertain = True
    if isinstance(meta_or_K, dict):
        m = meta_or_K
        K, T, max_t, init_seed = int(m["num_modes"]), int(m["future_steps"]), float(m["max_fut_t"]), int(m["init_seed"])
        uncertain = bool(int(m.get("uncertain", 1)))
    else:
        K = meta_or_K
    cfg = our_cfg(K, T, max_t, uncertain)
    return PredictionModelSDENet(**cfg, init_seed=init_seed).eval(), cfg


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def clone_batch(batch):
    from trajsde_amd.data import TemporalData
    return TemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.as_dict().items()})


def oracle_forward(model, cfg, batch, noise_seed, want_intermediates=True):
    import restate
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return restate.forward(P, cfg, clone_batch(batch).to("cpu"), restate.PhiloxNoise(int(noise_seed)),
                           want_intermediates=want_intermediates)


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def reference_l2(y, loc, reg_mask):
    """losses/L2.py:10-27 spelled out on tensors (mean reduction)"""
    l2 = torch.norm(y.unsqueeze(0) - loc, p=2, dim=-1)
    ade = l2.clone()
    ade[:, ~reg_mask] = 0
    best = torch.argmin(ade.mean(-1), dim=0)
    minl2 = l2[best, torch.arange(l2.size(1))]
    return minl2[reg_mask].mean(), best


def reference_laplace_nll(y, out_loc4, reg_mask, eps=1e-6):
    """losses/laplace_nll_loss.py:29-44 spelled out on tensors (mean reduction; the clamp is applied without a gradient)"""
    loc, scale = out_loc4.chunk(2, dim=-1)
    diff = torch.norm(y.unsqueeze(0) - loc, dim=-1)
    d_ = diff.clone()
    d_[:, ~reg_mask] = 0
    best = torch.argmin(d_.mean(-1), dim=0)
    ar = torch.arange(best.size(0))
    loc, scale = loc[best, ar], scale[best, ar]
    scale = scale.clone()
    with torch.no_grad():
        scale.clamp_(min=eps)
    nll = torch.log(2 * scale) + torch.abs(y - loc) / scale
    return nll[reg_mask].mean(), best


def oracle_full_grads(model, cfg, batch_cpu, seed, w_l2, w_diff, want_parts=False, drop=None, nll_eps=None):
    """end-to-end autograd over the oracle (float64): encoder -> aggregator -> decoder -> w_l2 L2 (or, with `nll_eps`, the
    Laplace NLL) + w_diff DiffBCE;
    `drop`: a restate.PhiloxDropout for train-mode dropout (the masks the HIP kernels cut from their Philox stream)"""
    import restate
    import torch.nn.functional as F
    from trajsde_amd.schedule import decoder_schedule, encoder_schedule
    c = restate.flat_cfg(cfg)
    es = encoder_schedule(c["historical_steps"], c["max_past_t"], c["minimum_step"])
    ds = decoder_schedule(c["future_steps"], c["max_fut_t"], c["min_stepsize"])
    dt = torch.float64
    P = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    names = [k for k in P if P[k].is_floating_point()]
    for k in names:
        P[k].requires_grad_(True)
    b = clone_batch(batch_cpu)
    for k in b.keys:
        if torch.is_tensor(b[k]) and b[k].is_floating_point():
            b[k] = b[k].to(dt)

    class Noise64(restate.PhiloxNoise):
        def fake_agent(self, shape):
            return super().fake_agent(shape).to(dt)

        def encoder(self, idx, shape):
            return super().encoder(idx, shape).to(dt)

        def decoder(self, k, shape):
            return super().decoder(k, shape).to(dt)

    torch.set_default_dtype(dt)
    try:
        rot, y_rot = restate.rotate_inputs(b)
        noise = Noise64(seed)
        with torch.enable_grad():
            local, diff_in, diff_out, _ = restate.local_encoder(P, c, b, rot, noise, es, False, drop)
            glob = restate.global_interactor(P, c, b, rot, local, None, drop)
            out = restate.sde_decoder(P, c, b, local, glob, noise, ds)
            if nll_eps is None:
                l2, _ = reference_l2(y_rot, out["loc"][..., :2], out["reg_mask"])
            else:                                                    # the regression loss is the Laplace NLL (both heads trained)
                l2, _ = reference_laplace_nll(y_rot, out["loc"], out["reg_mask"], nll_eps)
            bce = (F.binary_cross_entropy(diff_in, torch.zeros_like(diff_in)) +
                   F.binary_cross_entropy(diff_out, torch.ones_like(diff_out)))
            loss = w_l2 * l2 + w_diff * bce
            loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    if want_parts:
        return float(loss.detach()), {k: P[k].grad for k in names}, float(l2.detach()), float(bce.detach())
    return float(loss.detach()), {k: P[k].grad for k in names}


def perturb_parameters(model, seed):
    """leave the initial point the way oracle/make_golden_train.py does: every trainable parameter += 0.02 * N(0,1) drawn
    in parameters() order from one seeded CPU generator"""
    g = torch.Generator().manual_seed(int(seed))
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn(p.shape, generator=g).to(p.device))


def fixture_dropout(meta):
    """the PhiloxDropout a train-mode fixture was made with (None for eval-mode fixtures)"""
    import restate
    if "dropout_p" not in meta or float(meta["dropout_p"]) <= 0:
        return None
    return restate.PhiloxDropout(int(meta["dropout_seed"]), float(meta["dropout_p"]))


def load_train_fixture(name):
    from trajsde_amd.data import TemporalData
    z = np.load(os.path.join(ROOT, "tests", "golden_train", name + ".npz"))
    batch = TemporalData(**{k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")})
    batch["num_nodes"] = batch["x"].shape[0]
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta.")}
    losses = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss.")}
    weights = {k[7:]: float(z[k]) for k in z.files if k.startswith("weight.")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
    digests = {k[7:]: z[k] for k in z.files if k.startswith("digest.")}
    return batch, meta, losses, weights, grads, digests


def digest_signs(key, n):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return (torch.randint(0, 2, (n,), generator=g) * 2 - 1).double()


def check_grads_against_train_fixture(got, grads, digests, rel):
    """`got`: {name: tensor or None}.  Full tensors: max-abs error <= rel * largest entry; digests: norm, a seeded +-1
    projection and the leading entries.  The edge-embedding matrices get 2 * rel (ill-conditioned, see test_gpu_backward)."""
    bad = []
    for k, w in grads.items():
        g = got.get(k)
        scale = float(w.abs().max())
        if g is None:
            if scale > 0:
                bad.append((k, "missing", scale))
            continue
        r = 2 * rel if "_embed.module_list" in k else rel
        zero_by_symmetry = k.endswith("lin_k.bias") or k.endswith("lin_k_node.bias") or k.endswith("lin_k_edge.bias")
        err = float((g.detach().cpu().double() - w.double()).abs().max())
        if (err > 5e-5 or scale > 5e-5) if zero_by_symmetry else (err > r * scale + 1e-7):
            bad.append((k, err, scale))
    for k, d in digests.items():
        g = got.get(k)
        norm, proj, lead = float(d[0]), float(d[1]), torch.from_numpy(d[2:])
        if g is None:
            if norm > 0:
                bad.append((k, "missing", norm))
            continue
        gd = g.detach().cpu().double().reshape(-1)
        n = gd.numel()
        r = 2 * rel if "_embed.module_list" in k else rel
        zero_by_symmetry = k.endswith("lin_k.bias") or k.endswith("lin_k_node.bias") or k.endswith("lin_k_edge.bias")
        if zero_by_symmetry:
            if norm > 5e-4 or float(gd.norm()) > 5e-4:
                bad.append((k, float(gd.norm()), norm))
            continue
        tol = r * norm * n ** 0.5 + 1e-7
        if abs(float(gd.norm()) - norm) > tol or abs(float((gd * digest_signs(k, n)).sum()) - proj) > tol:
            bad.append((k, "digest", float(gd.norm()), norm))
        m = min(n, lead.numel())
        if float((gd[:m] - lead[:m]).abs().max()) > r * max(float(lead[:m].abs().max()), norm) + 1e-7:
            bad.append((k, "lead", float((gd[:m] - lead[:m]).abs().max())))
    return bad


def grid_cfg(K, T, heads, layers, dropout=0.0, uncertain=True):
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        cfg = yaml.safe_load(f)
    if not uncertain:
        cfg["decoder"]["kwargs"]["uncertain"] = False
    cfg["encoder"]["kwargs"]["dropout"] = dropout
    cfg["aggregator"]["kwargs"]["dropout"] = dropout
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["encoder"]["kwargs"].update(num_heads=heads, num_temporal_layers=layers)
    cfg["aggregator"]["kwargs"].update(num_modes=K, num_heads=heads)
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T)
    return cfg


def oracle_grid_full_grads(model, cfg, batch_cpu, d_local=None, drop=None):
    """float64 autograd over oracle/restate_grid.py: whole model under L2, or the encoder alone under sum(local * d_local);
    `drop`: a restate.PhiloxDropout for train mode (the masks the HIP kernels cut from their Philox stream)"""
    import restate
    import restate_grid
    c = restate_grid.flat_cfg(cfg)
    dt = torch.float64
    P = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    names = [k for k in P if P[k].is_floating_point() and not k.endswith("attn_mask")]
    for k in names:
        P[k].requires_grad_(True)
    b = clone_batch(batch_cpu)
    for k in b.keys:
        if torch.is_tensor(b[k]) and b[k].is_floating_point():
            b[k] = b[k].to(dt)
    torch.set_default_dtype(dt)
    try:
        rot, y_rot = restate.rotate_inputs(b)
        with torch.enable_grad():
            local = restate_grid.local_encoder_grid(P, c, b, rot, drop)
            if d_local is not None:
                loss = (local * d_local.cpu().to(dt)).sum()
            else:
                glob = restate.global_interactor(P, c, b, rot, local, None, drop)
                out = restate_grid.mlp_decoder(P, c, b, local, glob)
                loss, _ = reference_l2(y_rot, out["loc"][..., :2], out["reg_mask"])
            loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return float(loss.detach()), {k: P[k].grad for k in names}
