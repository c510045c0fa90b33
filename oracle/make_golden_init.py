"""Generate tests/golden_init/init_moments.npz: per-parameter moments of the REFERENCE's freshly constructed models
(models/model_base_mix_sde.py PredictionModelSDENet and the vanilla models/model_base_mix.py PredictionModel, imported from
/root/reference over oracle/shims), pooled over several torch seeds.  Build-container only.

    python oracle/make_golden_init.py

A fixture holds numbers only.  Per parameter name: [numel, mean, std, absmax, first element, is_constant] where the
moments are pooled over SEEDS constructions.  The CPU test (tests/test_init_golden.py) holds the build's own initialisers
(trajsde_amd/models/params.py) to the same families: constants exactly, random tensors by std and by the bound / tail shape
(absmax / std = sqrt(3) for a uniform law, > 2.5 for a normal one at these sizes).  What this pins: the
encoder's `self.apply(init_weights)` (ENC:64) runs AFTER `GRU_Unit` was built (ENC:49), so the N(0, 0.1) of
`init_network_weights` (ODEU:211-215) is overwritten by xavier-uniform (UTIL:94-98).
"""
import copy
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import ref_loader as R                                        # noqa: E402

SEEDS = (0, 1, 2, 3, 4, 5)
GRID_CFG = "configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml"


def moments(tensors):
    x = torch.stack([t.detach().double().reshape(-1) for t in tensors])          # [seeds, numel]
    flat = x.reshape(-1)
    const = bool((flat == flat[0]).all())
    return np.array([x.shape[1], float(flat.mean()), float(flat.std(unbiased=False)), float(flat.abs().max()), float(flat[0]),
                     1.0 if const else 0.0])


def collect(build):
    per_name = {}
    for s in SEEDS:
        model = build(s)
        for k, v in model.state_dict().items():
            if v.is_floating_point() and torch.isfinite(v).all():
                per_name.setdefault(k, []).append(v)
    return {k: moments(v) for k, v in per_name.items()}


def main():
    if not R.reference_available():
        sys.exit("reference tree not found; fixtures can only be generated in the build container")
    fx = {}
    cfg = R.load_reference_cfg(num_modes=6, future_steps=20, max_fut_t=2.0)
    for k, m in collect(lambda s: R.build_reference_model(copy.deepcopy(cfg), seed=s)).items():
        fx["sde." + k] = m
    with open(os.path.join(R.REFERENCE_ROOT, GRID_CFG)) as f:
        gcfg = yaml.safe_load(f)

    def build_grid(seed):
        R._install_paths()
        from importlib.machinery import SourceFileLoader
        with R.reference_cwd():
            torch.manual_seed(seed)
            ms = gcfg["model_specific"]
            mod = SourceFileLoader(ms["module_name"], ms["file_path"]).load_module(ms["module_name"])
            return getattr(mod, ms["module_name"])(**copy.deepcopy(gcfg))

    for k, m in collect(build_grid).items():
        fx["grid." + k] = m
    fx["meta.seeds"] = np.array(SEEDS)
    path = os.path.join(ROOT, "tests", "golden_init", "init_moments.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **fx)
    print(f"{len(fx) - 1} parameters -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
