"""Shared helpers for the parity tests: golden fixture loading, model construction, oracle runs."""
import glob
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def our_cfg(num_modes, future_steps, max_fut_t):
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model_specific"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps)
    cfg["aggregator"]["kwargs"]["num_modes"] = num_modes
    cfg["decoder"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps, max_fut_t=max_fut_t)
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
    if isinstance(meta_or_K, dict):
        m = meta_or_K
        K, T, max_t, init_seed = int(m["num_modes"]), int(m["future_steps"]), float(m["max_fut_t"]), int(m["init_seed"])
    else:
        K = meta_or_K
    cfg = our_cfg(K, T, max_t)
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
