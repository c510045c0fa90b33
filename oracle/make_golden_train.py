"""Generate tests/golden_train/*.npz: the REFERENCE's own training loss and parameter gradients on small synthetic batches.

    python oracle/make_golden_train.py            # build container only (needs /root/reference)

The reference model (imported from /root/reference over oracle/shims, our state_dict loaded key for key) runs its
`forward` in TRAIN mode with injected noise, its own loss modules (`losses/L2.py`, `losses/diff_BCE.py`, weighted as
`training_step` does, `models/model_base_mix_sde.py:104-111`) and torch.autograd.  Its dropout (p = 0.1 at four sites of
each of the five attention blocks, ENC:521-533,592,611, ENC:711-723,771,794, AGG:78-90,116,132) is served from injected
masks: the masks the HIP kernels cut from their Philox stream (host twin trajsde_amd/philox.py), laid out in the
reference's own element order (`ref_loader.injected_dropout`) -- so both sides differentiate the same function.
(The vanilla-variant case stays in eval mode: that variant's HIP training step refuses dropout.)

A fixture holds data only: the batch, the init / noise / dropout seeds, the two loss values, and every parameter gradient.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import ref_loader as R                                        # noqa: E402
from make_golden import our_cfg, philox_noise, state_checksum  # noqa: E402
from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet  # noqa: E402
from trajsde_amd.schedule import decoder_schedule            # noqa: E402
from trajsde_amd.synth import synth                          # noqa: E402

CASES = {
    # name: (synth kwargs, num_modes, future_steps, max_fut_t, init_seed, noise_seed, full)
    # last field: store every gradient tensor (True) or a digest per tensor (False) -- keeps the fixtures small
    "train_mixed_k3_t5": (dict(S=3, n=9, L=5, F=5, box=70.0, seed=21, mixed_source=True, history_dropout=0.3), 3, 5, 0.5, 5, 201, True),
    "train_argo_k6_t30": (dict(S=2, n=7, L=6, F=30, box=90.0, seed=22, source=1, history_dropout=0.2), 6, 30, 3.0, 6, 202, False),
    # BASELINE configs[3]: the shipped training shape (K=10 modes, T=60 future steps = 61 Euler steps, mixed sources; CFG:9-22)
    "train_shipped_k10_t60": (dict(S=4, n=6, L=5, F=60, box=80.0, seed=24, mixed_source=True, history_dropout=0.3, nus_sparsity=False),
                              10, 60, 6.0, 8, 204, False),
    # the reference's losses/laplace_nll_loss.py in place of losses/L2.py (its scale head is trained under it)
    "train_nll_k3_t5": (dict(S=3, n=9, L=5, F=5, box=70.0, seed=25, mixed_source=True, history_dropout=0.3), 3, 5, 0.5, 10, 205, True),
}
NLL_CASES = {"train_nll_k3_t5"}


def digest_signs(key, n):
    """the +-1 vector a gradient digest is projected on: seeded by the parameter name, shared with the tests"""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return (torch.randint(0, 2, (n,), generator=g) * 2 - 1).double()


def make(name):
    skw, K, T, max_t, init_seed, noise_seed, full = CASES[name]
    batch = synth(**skw)
    ours = PredictionModelSDENet(**our_cfg(K, T, max_t), init_seed=init_seed)
    g = torch.Generator().manual_seed(1000 + init_seed)
    with torch.no_grad():                                     # leave the initial point: zero biases hide bias-gradient bugs
        for p in ours.parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    ref_cfg = R.load_reference_cfg(num_modes=K, future_steps=T, max_fut_t=max_t)
    if name in NLL_CASES:                                     # the regression loss of this case: the reference's Laplace NLL module
        ref_cfg["losses"][0], ref_cfg["losses_module"][0] = "losses/laplace_nll_loss.py", "LaplaceNLLLoss"
        ref_cfg["loss_args"][0] = {"eps": 1e-6, "reduction": "mean"}
    ref = R.build_reference_model(ref_cfg)
    ref.load_state_dict(sd)
    ref.train()                                               # dropout on, served from injected masks (module docstring)
    p_drop = float(ref_cfg["encoder"]["kwargs"]["dropout"])
    assert p_drop == float(ref_cfg["aggregator"]["kwargs"]["dropout"]) == float(our_cfg(K, T, max_t)["encoder"]["kwargs"]["dropout"])
    dropout_seed = 7000 + noise_seed
    # (a float64 run of the reference is not the same function: the solver's time bookkeeping is float32 arithmetic and
    # takes a different number of Euler steps in double, SURVEY App. D -- so the fixture is the reference's float32
    # autograd result, whose own rounding noise on the ill-conditioned encoder gradients is ~1e-3 for T=30)
    N, A = batch.num_nodes, batch["agent_index"].numel()
    sched = decoder_schedule(T, max_t)
    z_fake, z_enc, z_dec = philox_noise(noise_seed, A, N + A, K * N, sched.n_euler)
    replay = [torch.from_numpy(z_fake)] + [torch.from_numpy(z) for z in z_enc] + [torch.from_numpy(z) for z in z_dec]

    # the dropout masks in the reference's call order and element order; the edge orders come from the oracle, whose lists
    # are the reference's (bit-exact restatement, tests/test_oracle_golden.py)
    import restate
    from trajsde_amd import philox
    orc = restate.forward({k: v.clone() for k, v in sd.items()}, our_cfg(K, T, max_t), batch, restate.PhiloxNoise(noise_seed),
                          want_intermediates=True)
    drop = restate.PhiloxDropout(dropout_seed, p_drop)
    Rn = 21 * (N + A)
    masks = []
    for block, (edges, rows) in enumerate([(orc["aa_edge_list"], Rn), (orc["al_edge_list"], N)] + [(orc["g_edge_list"], N)] * 3):
        like = torch.empty(rows, 64)
        masks += [drop.attn(block, edges[0], edges[1], 8, like), drop.feat(block, philox.DK_PROJ, like),
                  drop.feat(block, philox.DK_HIDDEN, torch.empty(rows, 256)), drop.feat(block, philox.DK_OUT, like)]
    R._install_paths()
    from noise_source import SOURCE
    SOURCE.reset(seed=None, replay=replay)
    data = R.to_reference_data(batch)
    # A ReLU input within float32 rounding of zero is a kink: which side a float32 evaluation lands on is a matter of summation
    # order, and the two one-sided gradients differ by that unit's whole contribution.  Fixtures that are asserted tensor by
    # tensor at a tight bound must not sit on one -- pick another init_seed when this trips.
    nearest = [float("inf")]
    hooks = [m.register_forward_pre_hook(lambda _m, a: nearest.__setitem__(0, min(nearest[0], float(a[0].detach().abs().min()))))
             for m in ref.modules() if isinstance(m, torch.nn.ReLU)]
    with R.reference_cwd(), R.injected_randn_like(), R.injected_dropout(masks) as served, torch.enable_grad():
        out = ref(data)
        parts = [fn(data, out) for fn in ref.losses]                                  # MODEL:108-110
        loss = sum(w * l for w, l in zip(ref.loss_weights, parts))
        loss.backward()
    assert len(SOURCE.record) == 1 + 21 + sched.n_euler
    for h in hooks:
        h.remove()
    print(f"{name}: nearest ReLU input to zero {nearest[0]:.3e}")
    assert not full or nearest[0] > 1e-6, f"{name}: a ReLU input at {nearest[0]:.2e} -- a kink, choose another init_seed"
    fx = {f"in.{k}": v.numpy() for k, v in batch.as_dict().items() if torch.is_tensor(v)}
    fx["meta.num_modes"], fx["meta.future_steps"], fx["meta.max_fut_t"] = K, T, max_t
    assert len(served) == 20, served
    fx["meta.init_seed"], fx["meta.noise_seed"], fx["meta.perturb_seed"] = init_seed, noise_seed, 1000 + init_seed
    fx["meta.dropout_p"], fx["meta.dropout_seed"] = p_drop, dropout_seed
    fx["meta.state_checksum"] = state_checksum(sd)
    for nm, w, l in zip(ref.loss_names, ref.loss_weights, parts):
        fx[f"loss.{nm}"] = np.float64(float(l))
        fx[f"weight.{nm}"] = np.float64(float(w))
    fx["loss.total"] = np.float64(float(loss))
    n_grad = 0
    for k, p in ref.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach().double().reshape(-1)
        n_grad += 1
        if full:
            fx[f"grad.{k}"] = p.grad.detach().numpy().astype(np.float32)
        else:                                                 # digest: norm, a seeded +-1 projection, the leading entries
            fx[f"digest.{k}"] = np.array([float(gr.norm()), float((gr * digest_signs(k, gr.numel())).sum())] +
                                         gr[:30].tolist(), dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden_train", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **fx)
    print(f"{name}: N={N} K={K} T={T} losses={[(n, round(float(l), 6)) for n, l in zip(ref.loss_names, parts)]} "
          f"grads={n_grad} -> {os.path.getsize(path) / 1024:.0f} KiB")


GRID_KINK_MARGIN = 5e-6           # (of 16 seeds tried per case the nearest input was above 5e-6 for two or three)
GRID_CASES = {
    # name: (synth kwargs, num_modes, future_steps, num_heads, temporal layers, init_seed[, dropout key])
    "train_grid_k3_t12_h4": (dict(S=3, n=9, L=6, F=12, box=70.0, seed=23, mixed_source=True, history_dropout=0.3), 3, 12, 4, 2, 43),
    # model.train() with the YAML's dropout 0.1: the reference's 36 dropout calls served from the Philox host twin's masks
    "train_grid_drop_k3_t12_h4": (dict(S=3, n=9, L=6, F=12, box=70.0, seed=26, mixed_source=True, history_dropout=0.3), 3, 12, 4, 2, 37, 7311),
}


def torch1_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **_):
    """what nn.MultiheadAttention computed under the reference's pinned torch 1.x (env.yml) for need_weights=False too: softmax of the
    masked scaled scores, F.dropout on the WEIGHTS, weights @ v.  torch 2.x routes this call to a fused kernel whose dropout does not
    pass through F.dropout; the generator puts the explicit form back so that the injected masks reach it."""
    import math
    import torch.nn.functional as F
    w = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(q.size(-1)) if scale is None else scale)
    if attn_mask is not None:
        w = w.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else w + attn_mask
    w = torch.softmax(w, dim=-1)
    if dropout_p > 0.0:
        w = F.dropout(w, p=dropout_p)
    return w @ v


def grid_dropout_masks(orc, drop, N, heads, layers, global_layers):
    """the masks of the vanilla reference forward in its call order and element order: AAEncoder (4), every TemporalEncoder layer
    (attention weights [N, heads, 22, 22], dropout1 / FFN / dropout2 on [22, N, .]), ALEncoder (4), the global layers (4 each)"""
    import restate_grid
    from trajsde_amd import philox
    masks = []

    def block(b, edges, rows):
        like = torch.empty(rows, 64)
        return [drop.attn(b, edges[0], edges[1], heads, like), drop.feat(b, philox.DK_PROJ, like),
                drop.feat(b, philox.DK_HIDDEN, torch.empty(rows, 256)), drop.feat(b, philox.DK_OUT, like)]
    masks += block(0, orc["aa_edge_list"], 21 * N)
    for l in range(layers):
        masks += list(restate_grid.temporal_masks(drop, l, N, heads, torch.empty(1)))
    masks += block(1, orc["al_edge_list"], N)
    for i in range(global_layers):
        masks += block(2 + i, orc["g_edge_list"], N)
    return masks


def make_grid(name):
    """the vanilla HiVT variant (models/model_base_mix.py PredictionModel, L2 only): deterministic, so the reference's float32
    autograd is a clean yardstick; stored as per-tensor digests"""
    import yaml
    import make_golden_grid as G
    from trajsde_amd.models.model_base_mix import PredictionModel
    skw, K, T, heads, layers, init_seed = GRID_CASES[name][:6]
    dropout_seed = GRID_CASES[name][6] if len(GRID_CASES[name]) > 6 else None
    batch = synth(**skw)
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        ours_cfg = G.edit(yaml.safe_load(f), K, T, heads, layers)
    ours = PredictionModel(**ours_cfg, init_seed=init_seed)
    g = torch.Generator().manual_seed(1000 + init_seed)
    with torch.no_grad():
        for p in ours.parameters():
            if p.requires_grad:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    with open(os.path.join(R.REFERENCE_ROOT, G.REF_CFG)) as f:
        ref_cfg = G.edit(yaml.safe_load(f), K, T, heads, layers)
    ref = R.build_reference_model(ref_cfg)
    ref.load_state_dict(sd)
    data = R.to_reference_data(batch)
    stock = torch.nn.TransformerEncoder.forward
    torch.nn.TransformerEncoder.forward = G.torch1_transformer_encoder_forward
    import contextlib
    import torch.nn.functional as F
    stock_sdpa = F.scaled_dot_product_attention
    inject = contextlib.nullcontext([])
    p_drop = 0.0
    if dropout_seed is None:
        ref.eval()                                            # dropout off
    else:
        import restate
        import restate_grid
        ref.train()                                           # dropout on, served from injected masks
        p_drop = float(ref_cfg["encoder"]["kwargs"]["dropout"])
        assert p_drop == float(ref_cfg["aggregator"]["kwargs"]["dropout"]) > 0
        orc = restate_grid.forward({k: v.clone() for k, v in sd.items()}, ours_cfg, batch, want_intermediates=True)
        masks = grid_dropout_masks(orc, restate.PhiloxDropout(dropout_seed, p_drop), batch.num_nodes, heads, layers,
                                   int(ref_cfg["aggregator"]["kwargs"]["num_layers"]))
        inject = R.injected_dropout(masks)
        F.scaled_dot_product_attention = torch1_attention
    # the ReLU kink guard of make() (module ReLUs: the embeddings and the attention blocks' MLPs; the transformer layers' functional
    # relu is not hooked).  The digests are asserted at 2e-4 of small gradients, so the margin is an order wider than there.
    nearest = [float("inf")]
    hooks = [m.register_forward_pre_hook(lambda _m, a: nearest.__setitem__(0, min(nearest[0], float(a[0].detach().abs().min()))))
             for m in ref.modules() if isinstance(m, torch.nn.ReLU)]
    try:
        with R.reference_cwd(), inject as served, torch.enable_grad():
            out = ref(data)
            parts = [fn(data, out) for fn in ref.losses]
            loss = sum(w * l for w, l in zip(ref.loss_weights, parts))
            loss.backward()
    finally:
        torch.nn.TransformerEncoder.forward = stock
        F.scaled_dot_product_attention = stock_sdpa
        for h in hooks:
            h.remove()
    print(f"{name}: nearest ReLU input to zero {nearest[0]:.3e}")
    assert nearest[0] > GRID_KINK_MARGIN, f"{name}: a ReLU input at {nearest[0]:.2e} -- a kink, choose another init_seed"
    if dropout_seed is not None:
        assert len(served) == 8 + 4 * layers + 4 * int(ref_cfg["aggregator"]["kwargs"]["num_layers"]), served
    fx = {f"in.{k}": v.numpy() for k, v in batch.as_dict().items() if torch.is_tensor(v)}
    fx.update({"meta.num_modes": K, "meta.future_steps": T, "meta.num_heads": heads, "meta.num_temporal_layers": layers,
               "meta.init_seed": init_seed, "meta.perturb_seed": 1000 + init_seed, "meta.state_checksum": G.state_checksum(sd)})
    if dropout_seed is not None:
        fx.update({"meta.dropout_p": p_drop, "meta.dropout_seed": dropout_seed})
    for nm, w, l in zip(ref.loss_names, ref.loss_weights, parts):
        fx[f"loss.{nm}"] = np.float64(float(l))
        fx[f"weight.{nm}"] = np.float64(float(w))
    fx["loss.total"] = np.float64(float(loss))
    n_grad = 0
    for k, p in ref.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach().double().reshape(-1)
        n_grad += 1
        fx[f"digest.{k}"] = np.array([float(gr.norm()), float((gr * digest_signs(k, gr.numel())).sum())] + gr[:30].tolist(),
                                     dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden_train", name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: N={batch.num_nodes} K={K} T={T} losses={[(n, round(float(l), 6)) for n, l in zip(ref.loss_names, parts)]} "
          f"grads={n_grad} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    if not R.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    for name in (sys.argv[1:] or list(CASES) + list(GRID_CASES)):
        make_grid(name) if name in GRID_CASES else make(name)
