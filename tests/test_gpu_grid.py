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


def _cfg(K, T, heads, layers):
    with open(os.path.join(H.ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        cfg = yaml.safe_load(f)
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
    model = PredictionModel(**_cfg(K, T, heads, layers), init_seed=int(z["meta.init_seed"])).eval().to(dev)
    data = batch.to(dev)
    with torch.no_grad():
        out = model(data)
    for k in ("loc", "pi"):
        assert H.maxdiff(out[k].cpu(), torch.from_numpy(z["out." + k])) <= TOL, k
    assert torch.equal(out["reg_mask"].cpu(), torch.from_numpy(z["out.reg_mask"]))
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


def test_vanilla_variant_is_inference_only_and_loud(dev):
    from trajsde_amd.models.model_base_mix import PredictionModel
    from trajsde_amd.synth import synth
    model = PredictionModel(**_cfg(3, 12, 4, 2), init_seed=1).to(dev)
    batch = synth(S=1, n=5, L=3, F=12, box=50.0, seed=1).to(dev)
    with pytest.raises(NotImplementedError):
        model.training_step(batch, 0)
    with pytest.raises(NotImplementedError):
        PredictionModel(**_cfg(3, 80, 4, 2), init_seed=1)          # 2T > 128 outputs per head


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
