"""Initial parameter distributions against the reference's freshly constructed models (tests/golden_init/init_moments.npz,
made by oracle/make_golden_init.py from the reference's own constructors over the shims: UTIL:94-159 init_weights,
ODEU:211-215, ENC:49 then ENC:64, DEC:69-70, AGG:26-36).  Every parameter family is held to the reference's law:
constants exactly; random tensors by their standard deviation and by the shape of their tail (absmax / std is sqrt(3) for
a uniform law, well above 2.5 for a normal one at these sizes), pooled over the same number of seeds on both sides."""
import os

import numpy as np
import pytest
import torch

import helpers as H

FIX = os.path.join(H.ROOT, "tests", "golden_init", "init_moments.npz")


def _pooled(build, seeds):
    per = {}
    for s in seeds:
        for k, v in build(int(s)).state_dict().items():
            if v.is_floating_point() and torch.isfinite(v).all():
                per.setdefault(k, []).append(v.double().reshape(-1))
    return {k: torch.stack(v).reshape(-1) for k, v in per.items()}


def _check(prefix, ours):
    z = np.load(FIX)
    names = [k for k in z.files if k.startswith(prefix)]
    assert len(names) > 200
    assert {k[len(prefix):] for k in names} == set(ours), "state_dict keys differ from the reference's"
    bad = []
    for k in names:
        numel, mean, std, amax, first, const = z[k]
        o = ours[k[len(prefix):]]
        assert o.numel() == int(numel) * len(z["meta.seeds"]), k
        if const:
            if not bool((o == first).all()):
                bad.append((k, "constant", first))
            continue
        n = o.numel()
        ostd, oamax = float(o.std(unbiased=False)), float(o.abs().max())
        # std of a sample std: ~ std / sqrt(2n) (normal) or less (uniform); 5 sigma + 1 %
        if abs(ostd - std) > std * (5.0 / (2 * n) ** 0.5 + 0.01):
            bad.append((k, "std", ostd, std))
        if abs(float(o.mean()) - mean) > 5.0 * std / n ** 0.5 + 1e-12:
            bad.append((k, "mean", float(o.mean()), mean))
        ref_uniform = amax / std < 2.0                    # bounded law: absmax / std -> sqrt(3); a normal sample of these sizes: > 2.5
        if ref_uniform:
            bound = std * 3.0 ** 0.5 * (1.0 + 5.0 / (2 * n) ** 0.5 + 0.01)
            if oamax > bound or oamax / ostd >= 2.0:
                bad.append((k, "bound (uniform law expected)", oamax, amax))
        elif n >= 256 and oamax / ostd < 2.5:
            bad.append((k, "tail (normal law expected)", oamax / ostd, amax / std))
    assert not bad, bad[:10]


def test_sde_model_initialisers_follow_the_reference():
    from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet
    z = np.load(FIX)
    ours = _pooled(lambda s: PredictionModelSDENet(**H.our_cfg(6, 20, 2.0), init_seed=s), z["meta.seeds"])
    _check("sde.", ours)
    # the case VERDICT r1 flagged: GRU_Unit's N(0, 0.1) is overwritten by xavier-uniform (ENC:49 then ENC:64)
    w = ours["encoder.gru_unit.update_gate.2.weight"]
    assert float(w.abs().max()) <= (6.0 / 128) ** 0.5 + 1e-6


def test_vanilla_model_initialisers_follow_the_reference():
    from trajsde_amd.models.model_base_mix import PredictionModel
    z = np.load(FIX)
    with open(os.path.join(H.ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        import yaml
        cfg = yaml.safe_load(f)
    ours = _pooled(lambda s: PredictionModel(**cfg, init_seed=s), z["meta.seeds"])
    _check("grid.", ours)
