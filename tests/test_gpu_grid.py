"""GPU parity (-m gpu) of the vanilla HiVT variant (LocalEncoder with the temporal transformer, GlobalInteractor with
4 or 8 heads, MLPDecoder) through the C-ABI: against the golden vectors made from the reference's own model
(tests/golden_grid) and against the oracle restatement on seeded synthetic batches.  Tolerance 1e-4 as for the SDE path."""
import glob
import os

import numpy as np
import pytest
import torch
import yaml

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4
GRID = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(H.ROOT, "tests", "golden_grid", "*.npz")))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from trajsde_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _cfg(K, T, heads, layers, dropout=0.0):
    """the shipped YAML (dropout 0.1 as the reference's) with the sizes of a test; `dropout` 0 unless a test is about it"""
    with open(os.path.join(H.ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        cfg = yaml.safe_load(f)
    cfg["encoder"]["kwargs"]["dropout"] = dropout
    cfg["aggregator"]["kwargs"]["dropout"] = dropout
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["encoder"]["kwargs"].update(num_heads=heads, num_temporal_layers=layers)
    cfg["aggregator"]["kwargs"].update(num_modes=K, num_heads=heads)
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T)
    return cfg


@pytest.mark.parametrize("name", GRID)
def test_vanilla_forward_matches_reference_golden(name, dev):
    from trajsde_amd.data import TemporalData
    from trajsde_amd.models.model_base_mix import PredictionModel
    z = np.load(os.path.join(H.ROOT, "tests", "golden_grid", name + ".npz"))
    batch = TemporalData(**{k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")})
    K, T, heads, layers = (int(z["meta." + k]) for k in ("num_modes", "future_steps", "num_heads", "num_temporal_layers"))
    cfg = _cfg(K, T, heads, layers)
    plain = "meta.uncertain" in z.files and int(z["meta.uncertain"]) == 0
    if plain:                                                    # dec_hivt_nusargo_grid.py:31: the decoder without its scale head
        cfg["decoder"]["kwargs"]["uncertain"] = False
    model = PredictionModel(**cfg, init_seed=int(z["meta.init_seed"])).eval().to(dev)
    assert plain == (not any(k.startswith("decoder.scale") for k in model.state_dict()))
    data = batch.to(dev)
    with torch.no_grad():
        out = model(data)
    assert out["loc"].shape[-1] == (2 if plain else 4)
    for k in ("loc", "pi"):
        assert H.maxdiff(out[k].cpu(), torch.from_numpy(z["out." + k])) <= TOL, k
    assert torch.equal(out["reg_mask"].cpu(), torch.from_numpy(z["out.reg_mask"]))
    if plain:                                                    # (the reference returns the embeddings with `uncertain: True` only, :56-59)
        assert "local_embed" not in out and "global_embed" not in out
        with pytest.raises(NotImplementedError):
            model.train().training_step(batch.to(dev), 0)
    else:
        assert H.maxdiff(out["local_embed"].cpu(), torch.from_numpy(z["mid.local_embed"])) <= TOL
        assert H.maxdiff(out["global_embed"].cpu(), torch.from_numpy(z["mid.global_embed"])) <= TOL
    assert H.maxdiff(data.y.cpu(), torch.from_numpy(z["out.y_rot"])) <= 1e-5


@pytest.mark.parametrize("S,n,K,T,heads,layers,kw", [
    (4, 24, 6, 30, 4, 4, dict(mixed_source=True, history_dropout=0.4)),
    (2, 40, 10, 60, 4, 4, dict(source=1)),
    (3, 17, 2, 5, 8, 1, dict(nus_sparsity=True)),
    (1, 1, 3, 64, 4, 2, dict()),                    # single actor, the widest head the decoder kernel takes
])
def test_vanilla_forward_matches_oracle_on_synthetic(S, n, K, T, heads, layers, kw, dev):
    import restate_grid
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    batch = synth(S=S, n=n, L=9, F=T, box=100.0, seed=700 + n, **kw)
    cfg = _cfg(K, T, heads, layers)
    model = PredictionModel(**cfg, init_seed=9).eval()
    want = restate_grid.forward({k: v.detach().clone() for k, v in model.state_dict().items()}, cfg, H.clone_batch(batch), True)
    model = model.to(dev)
    with torch.no_grad():
        out = model(batch.to(dev))
    for k in ("loc", "pi", "local_embed", "global_embed"):
        assert H.maxdiff(out[k].cpu(), want[k]) <= TOL, k
    assert torch.equal(out["reg_mask"].cpu(), want["reg_mask"])


def test_vanilla_variant_refuses_what_it_does_not_build(dev):
    from trajsde_amd.models.model_base_mix import PredictionModel
    with pytest.raises(NotImplementedError):
        PredictionModel(**_cfg(3, 80, 4, 2), init_seed=1)          # 2T > 128 outputs per head


def test_ts_drop_augmentation_masks_history_steps_like_the_reference(dev):
    """models/model_base_mix.py:96-100: a dropped step loses its input and becomes padding; bos steps and the current step
    are never dropped.  ts_drop = 1 drops every droppable step, which makes the augmented batch predictable: the training
    step on it equals the training step of a model without ts_drop on the hand-masked batch."""
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    cfg = _cfg(3, 12, 4, 2)
    batch = synth(S=2, n=7, L=4, F=12, box=60.0, seed=4, history_dropout=0.3)
    plain = PredictionModel(**cfg, init_seed=3).to(dev).train()
    cfg_ts = _cfg(3, 12, 4, 2)
    cfg_ts["model_specific"]["kwargs"]["ts_drop"] = 1.0
    aug = PredictionModel(**cfg_ts, init_seed=3).to(dev).train()
    a = batch.to(dev)
    loss_aug = aug.training_step(a, 0)
    mask = torch.ones(batch["x"].shape[0], 21, dtype=torch.bool)
    mask[batch["bos_mask"]] = False
    mask[:, -1] = False
    assert torch.equal(a["padding_mask"][:, :21].cpu(), batch["padding_mask"][:, :21] | mask)      # in place, like the reference
    assert float(a["x"].cpu()[mask].abs().max()) == 0.0
    b = H.clone_batch(batch)
    b["x"][mask] = 0
    b["padding_mask"][:, :21] |= mask
    loss_plain = plain.training_step(b.to(dev), 0)
    assert torch.equal(loss_aug.detach(), loss_plain.detach())
    # a fractional rate drops about that fraction of the droppable steps
    cfg_ts["model_specific"]["kwargs"]["ts_drop"] = 0.3
    frac = PredictionModel(**cfg_ts, init_seed=3).to(dev).train()
    big = synth(S=4, n=60, L=4, F=12, box=90.0, seed=5).to(dev)
    before = big["padding_mask"][:, :21].clone()
    frac.apply_ts_drop(big, generator=torch.Generator(device=dev).manual_seed(1))
    dropped = (big["padding_mask"][:, :21] & ~before).float().mean().item() * 21 / 19          # 19 of 21 steps are droppable here
    assert 0.25 < dropped < 0.35


def _l2(y, loc, reg_mask):
    l2 = torch.norm(y.unsqueeze(0) - loc, p=2, dim=-1)
    ade = l2.clone()
    ade[:, ~reg_mask] = 0
    best = torch.argmin(ade.mean(-1), dim=0)
    return l2[best, torch.arange(l2.size(1))][reg_mask].mean(), best


@pytest.mark.parametrize("S,n,K,T", [(3, 14, 4, 30), (2, 9, 1, 64), (2, 21, 10, 60)])
def test_mlp_decoder_l2_backward_matches_autograd(S, n, K, T, dev):
    import restate_grid
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    batch = synth(S=S, n=n, L=5, F=T, box=80.0, seed=800 + n, mixed_source=True, history_dropout=0.3)
    cfg = _cfg(K, T, 4, 2)
    model = PredictionModel(**cfg, init_seed=3).to(dev)
    data = batch.to(dev)
    with torch.no_grad():
        out = model(data)                                           # rotates data.y
    local, glob = out["local_embed"], out["global_embed"]
    res = model.decoder._rt.mlp_decoder_l2_backward(data, local, glob, out)
    torch.cuda.synchronize()
    c = restate_grid.flat_cfg(cfg)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k in P if k.startswith("decoder.")]
    for k in names:
        P[k].requires_grad_(True)
    lo, gl = local.detach().cpu().clone().requires_grad_(True), glob.detach().cpu().clone().requires_grad_(True)
    with torch.enable_grad():
        o = restate_grid.mlp_decoder(P, c, batch, lo, gl)
        loss, best = _l2(data.y.cpu(), o["loc"][..., :2], o["reg_mask"])
        loss.backward()
    assert torch.equal(res["best_mode"].cpu().long(), best)
    assert abs(float(res["loss"]) - float(loss.detach())) <= 1e-5 * max(1.0, float(loss.detach()))
    got = res["grads"]
    for k in names:
        short = k[len("decoder."):]
        want = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        if short not in got:
            assert float(want.abs().max()) == 0.0, k                 # scale / pi heads
            continue
        scale = float(want.abs().max())
        assert float((got[short].cpu().double() - want.double()).abs().max()) <= 2e-4 * scale + 1e-7, k
    for a, b in ((res["d_local_embed"], lo.grad), (res["d_global_embed"], gl.grad)):
        assert float((a.cpu() - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7


_oracle_full = H.oracle_grid_full_grads


def _compare(named_grads, want, prefix=""):
    bad = []
    for n, g in named_grads:
        w = want[prefix + n]
        if g is None:
            assert w is None or float(w.abs().max()) == 0.0, n
            continue
        scale = float(w.abs().max())
        err = float((g.cpu().double() - w).abs().max())
        zero_by_symmetry = n.endswith("lin_k.bias") or n.endswith("lin_k_node.bias") or n.endswith("lin_k_edge.bias")
        if (err > 5e-5 or scale > 5e-5) if zero_by_symmetry else (err > 2e-4 * scale + 1e-7):
            bad.append((n, err, scale))
    assert not bad, bad


@pytest.mark.parametrize("S,n,heads,layers,kw", [
    (3, 12, 4, 2, dict(mixed_source=True, history_dropout=0.4)),
    (2, 9, 8, 1, dict(source=1, history_dropout=0.2)),
])
def test_vanilla_encoder_backward_matches_autograd(S, n, heads, layers, kw, dev):
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    batch = synth(S=S, n=n, L=6, F=5, box=60.0, seed=900 + n, **kw)
    cfg = _cfg(2, 5, heads, layers)
    model = PredictionModel(**cfg, init_seed=5).to(dev)
    data = batch.to(dev)
    rot, y_rot = runtime.rotate_inputs(data)
    data.y, data["rotate_mat"] = y_rot, rot
    with torch.no_grad():
        local = model.encoder(data=data)
    d_local = torch.randn(local.shape, generator=torch.Generator().manual_seed(2))
    res = model.encoder._rt.encoder_grid_backward(data, d_local.to(dev))
    torch.cuda.synchronize()
    _, want = _oracle_full(model, cfg, batch, d_local)
    _compare(res["grads"].items(), want, "encoder.")


def _cfg_drop(K, T, heads, layers, p=0.1):
    return _cfg(K, T, heads, layers, dropout=p)                   # the reference's YAML value (hivt_nuSArgo_trmenc_mlpdec.yml:33)


@pytest.mark.parametrize("S,n,heads,layers,kw", [
    (3, 12, 4, 2, dict(mixed_source=True, history_dropout=0.4)),
    (2, 9, 8, 1, dict(source=1, history_dropout=0.2)),
])
def test_vanilla_encoder_train_mode_dropout_forward_and_backward(S, n, heads, layers, kw, dev):
    """model.train() with the reference's dropout 0.1: the 4 sites of the AA and AL blocks and the 4 of every TemporalEncoder layer
    (nn.MultiheadAttention's dropout on the softmax output, dropout1, the FFN's dropout, dropout2: GENC:256-283).  Forward against
    the oracle fed the host twin's masks (1e-4), every encoder gradient against float64 autograd over it; eval mode is untouched
    by the key, train mode depends on it."""
    import restate
    import restate_grid
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    batch = synth(S=S, n=n, L=6, F=5, box=60.0, seed=900 + n, **kw)
    cfg = _cfg_drop(2, 5, heads, layers)
    model = PredictionModel(**cfg, init_seed=5).to(dev)
    H.perturb_parameters(model, 78)          # (77 puts an FFN unit of temporal layer 0 on its ReLU kink for the first case: 2e-3 on linear1)
    data = batch.to(dev)
    rot, y_rot = runtime.rotate_inputs(data)
    data.y, data["rotate_mat"] = y_rot, rot
    noise = runtime.NoiseSpec(seed=3, dropout_seed=4242)
    with torch.no_grad():
        model.eval()
        ev = model.encoder(data=data, noise=noise)
        assert torch.equal(ev, model.encoder(data=data))                    # eval: no dropout, whatever the key
        model.train()
        tr = model.encoder(data=data, noise=noise)
        assert torch.equal(tr, model.encoder(data=data, noise=noise))       # same key, same masks
        assert not torch.equal(tr, model.encoder(data=data, noise=runtime.NoiseSpec(seed=3, dropout_seed=4243)))
        assert float((tr - ev).abs().max()) > 1e-3
    drop = restate.PhiloxDropout(4242, 0.1)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    b = H.clone_batch(batch)
    r_cpu, _ = restate.rotate_inputs(b)
    with torch.no_grad():
        want = restate_grid.local_encoder_grid(P, restate_grid.flat_cfg(cfg), b, r_cpu, drop)
    assert float((tr.cpu() - want).abs().max()) <= 1e-4
    g = torch.Generator().manual_seed(5)
    d_local = torch.randn(tr.shape, generator=g).to(dev)
    res = model.encoder._rt.encoder_grid_backward(data, d_local, noise)
    torch.cuda.synchronize()
    _, want_g = _oracle_full(model, cfg, batch, d_local, drop)
    _compare(res["grads"].items(), want_g, "encoder.")


def test_vanilla_training_step_in_train_mode_with_the_reference_dropout(dev):
    """the whole vanilla training step under model.train() with dropout 0.1 in encoder and aggregator (36 dropout sites):
    loss and every gradient against float64 autograd over the oracle with the host twin's masks"""
    import restate
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    K, T = 3, 12
    batch = synth(S=3, n=11, L=6, F=T, box=70.0, seed=93, mixed_source=True, history_dropout=0.3)
    cfg = _cfg_drop(K, T, 4, 2)
    model = PredictionModel(**cfg, init_seed=8)
    H.perturb_parameters(model, 1234)
    model = model.to(dev).train()
    loss = model.training_step(batch.to(dev), 0, noise=runtime.NoiseSpec(seed=1, dropout_seed=99))
    loss.backward()
    torch.cuda.synchronize()
    want_loss, want = _oracle_full(model, cfg, batch, None, restate.PhiloxDropout(99, 0.1))
    assert abs(float(loss.detach()) - want_loss) <= 1e-5 * max(1.0, want_loss)
    reached = {id(p) for p in model.params_with_gradient()}
    _compare(((n, p.grad) for n, p in model.named_parameters() if id(p) in reached), want)
    with pytest.raises(Exception, match="NoiseSpec|dropout"):
        model.encoder._rt.encoder_grid_forward(batch.to(dev), None)          # train mode without a key is refused, never silently eval


def test_vanilla_training_step_repeated_is_bitwise_identical(dev):
    """the vanilla variant's whole training step (grid encoder with its temporal transformer, aggregator, MLP decoder; train mode, dropout
    0.1) six times on one batch of 24 scenes x 64 agents with the same keys: the same loss bits, the same gradient words, the same
    output -- the reproducibility check of tests/test_gpu_backward.py *_bitwise_identical for the entry points only this variant has"""
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    K, T = 6, 30
    batch = synth(S=24, n=64, L=24, F=T, box=120.0, seed=17, mixed_source=True, history_dropout=0.2).to(dev)
    y0 = batch.y.clone()
    model = PredictionModel(**_cfg_drop(K, T, 8, 4), init_seed=3)
    H.perturb_parameters(model, 77)
    model = model.to(dev).train()
    ref = None
    for call in range(6):
        model.zero_grad(set_to_none=True)
        batch.y = y0
        loss = model.training_step(batch, 0, noise=runtime.NoiseSpec(seed=5, dropout_seed=6))
        loss.backward()
        torch.cuda.synchronize()
        cur = (loss.detach().clone(), model.last_output["loc"].clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            assert len(cur[2]) > 100 and all(bool(torch.isfinite(g).all()) for g in cur[2].values())
            continue
        assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), call
        bad = [n for n in ref[2] if not torch.equal(cur[2][n], ref[2][n])]
        assert not bad, (call, bad[:6])


def test_vanilla_pipelined_training_loop_is_the_plain_loop(dev):
    """driver.train over the vanilla variant, next batch prepared on the side stream (this variant's graph stage: no fake agents, no
    noise) against the plain loop: the same losses and parameters bit for bit"""
    from trajsde_amd import driver, runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    K, T = 3, 12
    base = [synth(S=2 + (k % 2), n=8 + 3 * k, L=6, F=T, box=70.0, seed=40 + k, mixed_source=True, history_dropout=0.2) for k in range(3)]

    def per_epoch(epoch):
        for b in base:
            yield H.clone_batch(b).to(dev)
    prefetched, stock = [], runtime.prefetch_graph

    def counting(data, *args, **kw):
        prefetched.append(int(data["x"].shape[0]))
        return stock(data, *args, **kw)

    def run(pipelined):
        m = PredictionModel(**_cfg_drop(K, T, 4, 2), init_seed=5)
        m.lr, m.weight_decay, m.T_max = 1e-3, 1e-4, 4
        m = m.to(dev)
        m.pipeline_training = pipelined
        hist = driver.train(m, per_epoch, epochs=2, seed=3)
        torch.cuda.synchronize()
        return m, hist
    runtime.prefetch_graph = counting
    try:
        a, hist_a = run(True)
        assert len(prefetched) == 6
        b, hist_b = run(False)
        assert len(prefetched) == 6
    finally:
        runtime.prefetch_graph = stock
    assert len(hist_a) == 6 and hist_a == hist_b
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n


def test_vanilla_training_step_matches_end_to_end_autograd_and_trains(dev):
    from trajsde_amd import driver
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    K, T = 3, 12
    batch = synth(S=3, n=11, L=6, F=T, box=70.0, seed=91, mixed_source=True, history_dropout=0.3)
    cfg = _cfg(K, T, 4, 2)
    model = PredictionModel(**cfg, init_seed=7).to(dev).train()
    data = batch.to(dev)
    y0 = data.y.clone()
    loss = model.training_step(data, 0)
    loss.backward()
    torch.cuda.synchronize()
    want_loss, want = _oracle_full(model, cfg, batch)
    assert abs(float(loss.detach()) - want_loss) <= 1e-5 * max(1.0, want_loss)
    reached = {id(p) for p in model.params_with_gradient()}
    _compare(((n, p.grad) for n, p in model.named_parameters() if id(p) in reached), want)
    assert all(p.grad is None for n, p in model.named_parameters() if id(p) not in reached)
    model.zero_grad(set_to_none=True)
    model.lr, model.weight_decay, model.T_max = 2e-3, 1e-4, 10

    def fresh(epoch):
        for _ in range(6):
            data.y = y0
            yield data
    hist = driver.train(model, fresh, epochs=2)
    assert len(hist) == 12 and sum(hist[-3:]) < sum(hist[:3]), hist


@pytest.mark.parametrize("name", ["train_grid_k3_t12_h4", "train_grid_drop_k3_t12_h4"])
def test_vanilla_training_step_matches_the_reference_training_step(name, dev):
    """loss and parameter-gradient digests of the HIP training step of the vanilla variant against the REFERENCE's own model
    (models/model_base_mix.py), L2 module and torch.autograd (tests/golden_train/train_grid_*.npz): one fixture with dropout off,
    one made under model.train() with the YAML's dropout 0.1 and the reference's 28 dropout calls (2 temporal layers) served from the
    masks of the Philox host twin (oracle/make_golden_train.py make_grid)"""
    from trajsde_amd import runtime
    from trajsde_amd.models.model_base_mix import PredictionModel
    batch, meta, losses, weights, grads, digests = H.load_train_fixture(name)
    p = float(meta.get("dropout_p", 0.0))
    cfg = _cfg(int(meta["num_modes"]), int(meta["future_steps"]), int(meta["num_heads"]), int(meta["num_temporal_layers"]), dropout=p)
    model = PredictionModel(**cfg, init_seed=int(meta["init_seed"]))
    H.perturb_parameters(model, int(meta["perturb_seed"]))
    model = model.to(dev).train()
    noise = runtime.NoiseSpec(seed=0, dropout_seed=int(meta["dropout_seed"])) if p > 0 else None
    loss = model.training_step(batch.to(dev), 0, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - losses["total"]) <= 1e-5 * max(1.0, abs(losses["total"]))
    bad = H.check_grads_against_train_fixture({n: p.grad for n, p in model.named_parameters()}, grads, digests, rel=2e-4)
    assert not bad, bad[:8]
