"""The small launches of a training step folded into single calls (ABI 10): the six weight images of a step packed by one call
(trajsde_pack_weights_many), the stage gradient buffers gathered into the flat gradient tensor by one launch
(trajsde_grad_gather_add), AdamW over the flat parameter tensor in one launch (trajsde_adamw_step).  Each against what it replaces:
the per-stage packing call, torch's index_select + addcmul_, torch.optim.AdamW's single-tensor form (MODEL:204-207)."""

import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from trajsde_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _stage_entries(model):
    from trajsde_amd import _lib
    e, a, d = model.encoder._rt, model.aggregator._rt, model.decoder._rt
    return [(e, _lib.STAGE_ENCODER), (e, _lib.STAGE_ENCODER_BWD), (a, _lib.STAGE_AGGREGATOR), (a, _lib.STAGE_AGGREGATOR_BWD),
            (d, _lib.STAGE_DECODER), (d, _lib.STAGE_DECODER_BWD), (d, _lib.STAGE_DECODER_NLL_BWD)]


def test_one_packing_call_leaves_the_images_of_the_per_stage_calls(dev):
    """seven images (forward and backward of the three stages, both decoder losses) through PackSet.refresh(): bit-identical to
    StageRuntime.blob() stage by stage; after an in-place parameter update + touch() the SAME blobs are re-packed (no upload: the
    table stays on the device) and equal the per-stage images of the updated parameters; untouched parameters: no launch at all"""
    from trajsde_amd import runtime
    model, _ = H.build_model(6, 20, 2.0, init_seed=3)
    model = model.to(dev)
    twin, _ = H.build_model(6, 20, 2.0, init_seed=3)
    twin = twin.to(dev)
    ps = runtime.PackSet(_stage_entries(model))
    ps.refresh()
    torch.cuda.synchronize()
    first = [b.clone() for b in ps._blobs]
    for (rt, sid), (rt2, _), blob in zip(ps.entries, _stage_entries(twin), ps._blobs):
        assert rt.blob(sid) is blob                                   # the stage's cache now holds the set's blob
        ref = rt2.blob(sid)
        assert ref.numel() == blob.numel() and torch.equal(ref, blob), sid
    ptrs = [b.data_ptr() for b in ps._blobs]
    ps.refresh()                                                          # nothing changed: nothing packed
    assert [b.data_ptr() for b in ps._blobs] == ptrs and ps._fresh == 0
    with torch.no_grad():
        for m in (model, twin):
            g = torch.Generator(device="cpu").manual_seed(5)
            for p in m.parameters():
                p.add_((0.01 * torch.randn(p.shape, generator=g)).to(dev))
            for s in m.modules():
                if hasattr(s, "touch"):
                    s.touch()
    ps.refresh()
    torch.cuda.synchronize()
    assert [b.data_ptr() for b in ps._blobs] == ptrs
    changed = 0
    for (rt, sid), (rt2, _), blob, old in zip(ps.entries, _stage_entries(twin), ps._blobs, first):
        assert torch.equal(rt2.blob(sid), blob), sid
        changed += int(not torch.equal(blob, old))
    assert changed == len(first)


def test_training_step_packs_once_and_matches_the_per_stage_route(dev):
    """model.training_step through the PackSet against the same step with the set disabled (every blob() packs its own stage):
    loss and every gradient bit-identical; the step leaves six current images behind"""
    from trajsde_amd import runtime
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import synth
    batch = synth(S=2, n=10, L=5, F=6, box=50.0, seed=21, mixed_source=True).to(dev)
    y0 = batch.y.clone()
    out = []
    for use_set in (True, False):
        m, _ = H.build_model(3, 6, 0.5, init_seed=9)
        m = m.to(dev).train()
        if not use_set:
            class _Off:
                def refresh(self):
                    pass
            m.__dict__["_step_pack_set"] = lambda: _Off()
        batch.y = y0.clone()
        loss = m.training_step(batch, 0, noise=NoiseSpec(seed=40))
        loss.backward()
        if use_set:
            ps = m.__dict__["_pack_set_obj"]
            assert isinstance(ps, runtime.PackSet) and len(ps.entries) == 6
            for (rt, sid), blob in zip(ps.entries, ps._blobs):
                assert rt._blobs[sid][0] is blob
        out.append((float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert out[0][0] == out[1][0]
    assert out[0][1].keys() == out[1][1].keys() and len(out[0][1]) > 100
    for n in out[0][1]:
        assert torch.equal(out[0][1][n], out[1][1][n]), n


@pytest.mark.parametrize("form", ["foreach", "single"])
@pytest.mark.parametrize("n", [1, 7, 1000, 530_001])
def test_adamw_launch_is_torch_adamw_element_by_element(n, form, dev):
    """five steps of driver.FlatAdamW against torch.optim.AdamW from the same parameters, gradients and schedule: parameters and both
    moments bit-identical after every step (the kernel performs torch's operations in torch's order with torch's roundings).
    form "foreach": torch's multi-tensor implementation -- what `AdamW(model.parameters())` of MODEL:205 runs on a GPU -- which divides
    sqrt(exp_avg_sq) by sqrt(1 - beta2^step); form "single": `foreach=False`, which multiplies by the reciprocal.  state_dict
    interchangeable"""
    from trajsde_amd import driver
    g = torch.Generator(device="cpu").manual_seed(n)
    p0 = torch.randn(n, generator=g)
    pa = torch.nn.Parameter(p0.clone().to(dev))
    pb = torch.nn.Parameter(p0.clone().to(dev))
    a = torch.optim.AdamW([pa], lr=3e-3, weight_decay=1e-2, foreach=form == "foreach")
    b = driver.FlatAdamW([pb], lr=3e-3, weight_decay=1e-2, form=form)
    sa = torch.optim.lr_scheduler.CosineAnnealingLR(a, T_max=4, eta_min=0.0)
    sb = torch.optim.lr_scheduler.CosineAnnealingLR(b, T_max=4, eta_min=0.0)
    for step in range(5):
        grad = (torch.randn(n, generator=g) * (10.0 ** (step - 2))).to(dev)
        if n > 6:
            grad[3] = 0.0                                      # a dead element: 0 / (0 + eps)
        pa.grad, pb.grad = grad.clone(), grad.clone()
        a.step()
        b.step()
        sa.step()
        sb.step()
        assert torch.equal(pa.detach(), pb.detach()), (step, float((pa.detach() - pb.detach()).abs().max()))
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(a.state[pa][key], b.state[pb][key]), (step, key)
        assert float(a.state[pa]["step"]) == float(b.state[pb]["step"]) == step + 1
    # either optimizer resumes the other's checkpoint
    c = torch.optim.AdamW([torch.nn.Parameter(pb.detach().clone())], lr=3e-3, weight_decay=1e-2, foreach=form == "foreach")
    c.load_state_dict(b.state_dict())
    d = driver.FlatAdamW([torch.nn.Parameter(pa.detach().clone())], lr=3e-3, weight_decay=1e-2, form=form)
    d.load_state_dict(a.state_dict())
    grad = torch.randn(n, generator=g).to(dev)
    for opt in (c, d):
        opt.param_groups[0]["params"][0].grad = grad.clone()
        opt.step()
    assert torch.equal(c.param_groups[0]["params"][0].detach(), d.param_groups[0]["params"][0].detach())


def test_adamw_launch_leaves_unsupported_forms_to_torch(dev):
    """amsgrad / a CPU tensor: torch's own step runs (same results as torch.optim.AdamW, state layout torch's)"""
    from trajsde_amd import driver
    for kwargs, device in (({"amsgrad": True}, dev), ({}, torch.device("cpu"))):
        p0 = torch.randn(33)
        pa, pb = torch.nn.Parameter(p0.clone().to(device)), torch.nn.Parameter(p0.clone().to(device))
        a = torch.optim.AdamW([pa], lr=1e-2, **kwargs)                   # torch's own choice of implementation (`foreach=None`) ...
        b = driver.FlatAdamW([pb], lr=1e-2, **kwargs)                    # ... which is what the default form leaves to it
        assert b.param_groups[0]["foreach"] is None and driver.FlatAdamW([torch.nn.Parameter(p0.clone())], form="single").param_groups[0]["foreach"] is False
        for _ in range(2):
            g = torch.randn(33).to(device)
            pa.grad, pb.grad = g.clone(), g.clone()
            a.step()
            b.step()
        assert torch.equal(pa.detach(), pb.detach())
        assert set(a.state[pa]) == set(b.state[pb])


def test_gather_launch_adds_what_index_select_and_addcmul_add(dev):
    """three (block of the flat tensor, stage buffer, gather index) items in one launch against flat[block].addcmul_(src[index], s)"""
    from trajsde_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(1)
    flat = torch.randn(5000, generator=g).to(dev)
    ref = flat.clone()
    scale = torch.tensor(0.37, device=dev)
    srcs = [torch.randn(k, generator=g).to(dev) for k in (900, 2500, 64)]
    idx = [torch.randperm(s.numel(), generator=g)[:k].to(dev) for s, k in zip(srcs, (700, 2500, 1))]
    firsts, mults = (10, 1000, 4999), (1.0, 0.25, 3.0)
    items = (_lib.GatherItem * 3)()
    for it, f, s, i, m in zip(items, firsts, srcs, idx, mults):
        it.dst, it.src, it.index, it.n, it.mult = flat.data_ptr() + 4 * f, s.data_ptr(), i.data_ptr(), i.numel(), m
        ref[f:f + i.numel()].addcmul_(s.index_select(0, i), scale * m)
    _lib.check(_lib.lib().trajsde_grad_gather_add(items, 3, scale.data_ptr(), torch.cuda.current_stream().cuda_stream), "gather")
    torch.cuda.synchronize()
    assert torch.equal(flat, ref)
    before = flat.clone()
    _lib.check(_lib.lib().trajsde_grad_gather_add(items, 3, None, torch.cuda.current_stream().cuda_stream), "gather")
    for f, s, i, m in zip(firsts, srcs, idx, mults):
        before[f:f + i.numel()].addcmul_(s.index_select(0, i), torch.tensor(m, device=dev))
    assert torch.equal(flat, before)
    with pytest.raises(_lib.TrajsdeError):
        _lib.check(_lib.lib().trajsde_grad_gather_add(items, 9, None, 0), "gather")
