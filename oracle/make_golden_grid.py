"""Generate tests/golden_grid/*.npz: the REFERENCE's vanilla HiVT model (models/model_base_mix.py PredictionModel with
configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml, imported from /root/reference over oracle/shims) run on small synthetic
batches.  Deterministic model, so a fixture is: the batch, the seed of the build's weight init (+ state checksum) and the
reference's outputs / stage boundaries.  Build-container only; also prints the restatement's deviation.

    python oracle/make_golden_grid.py
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
import restate_grid                                           # noqa: E402
from trajsde_amd.models.model_base_mix import PredictionModel  # noqa: E402
from trajsde_amd.synth import synth                          # noqa: E402

REF_CFG = "configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml"
CASES = {
    # name: (synth kwargs, num_modes, future_steps, num_heads, temporal layers, init_seed)
    "grid_k10_t60_h4": (dict(S=2, n=7, L=5, F=60, box=90.0, seed=21, mixed_source=True, history_dropout=0.4), 10, 60, 4, 4, 0),
    "grid_k3_t12_h4": (dict(S=3, n=9, L=6, F=12, box=70.0, seed=22, source=0, nus_sparsity=True), 3, 12, 4, 2, 1),
    "grid_k6_t30_h8": (dict(S=2, n=11, L=4, F=30, box=110.0, seed=23, source=1, history_dropout=0.3), 6, 30, 8, 3, 2),
    # `uncertain: False` (dec_hivt_nusargo_grid.py:31, :58-59): no scale head, loc [K, N, T, 2]
    "grid_plain_k3_t12_h4": (dict(S=2, n=8, L=5, F=12, box=80.0, seed=24, mixed_source=True, history_dropout=0.3), 3, 12, 4, 2, 3, False),
}


def torch1_transformer_encoder_forward(self, src, mask=None, src_key_padding_mask=None, **_):
    """nn.TransformerEncoder.forward as in the torch 1.x the reference targets: current torch passes `is_causal=` to
    every layer, which the reference's own TemporalEncoderLayer.forward (GENC:270-277) does not accept."""
    output = src
    for mod in self.layers:
        output = mod(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
    return output if self.norm is None else self.norm(output)


def edit(cfg, K, T, heads, layers, uncertain=True):
    cfg = copy.deepcopy(cfg)
    if not uncertain:
        cfg["decoder"]["kwargs"]["uncertain"] = False
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["encoder"]["kwargs"].update(num_heads=heads, num_temporal_layers=layers)
    cfg["aggregator"]["kwargs"].update(num_modes=K, num_heads=heads)
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T)
    return cfg


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values() if torch.isfinite(v).all()))


def make(name):
    skw, K, T, heads, layers, init_seed = CASES[name][:6]
    uncertain = CASES[name][6] if len(CASES[name]) > 6 else True
    batch = synth(**skw)
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        ours_cfg = edit(yaml.safe_load(f), K, T, heads, layers, uncertain)
    ours = PredictionModel(**ours_cfg, init_seed=init_seed)
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    with open(os.path.join(R.REFERENCE_ROOT, REF_CFG)) as f:
        ref_cfg = edit(yaml.safe_load(f), K, T, heads, layers, uncertain)
    ref = R.build_reference_model(ref_cfg)
    ref.load_state_dict(sd)                                   # key-for-key, buffers included
    caps = {}
    hooks = [ref.encoder.register_forward_hook(lambda m, a, o: caps.__setitem__("local_embed", o)),
             ref.aggregator.register_forward_hook(lambda m, a, o: caps.__setitem__("global_embed", o)),
             ref.encoder.temporal_encoder.register_forward_hook(lambda m, a, o: caps.__setitem__("temporal_out", o))]
    data = R.to_reference_data(batch)
    stock = torch.nn.TransformerEncoder.forward
    torch.nn.TransformerEncoder.forward = torch1_transformer_encoder_forward
    try:
        with R.reference_cwd(), torch.no_grad():
            out = ref(data)
    finally:
        torch.nn.TransformerEncoder.forward = stock
    for h in hooks:
        h.remove()
    P = {k: v for k, v in sd.items()}
    mine = restate_grid.forward(P, ours_cfg, batch, want_intermediates=True)
    dev = {k: float((mine[k] - out[k]).abs().max()) for k in ("loc", "pi")}
    dev["local_embed"] = float((mine["local_embed"] - caps["local_embed"]).abs().max())
    fx = {f"in.{k}": v.numpy() for k, v in batch.as_dict().items() if torch.is_tensor(v)}
    fx.update({"meta.num_modes": K, "meta.future_steps": T, "meta.num_heads": heads, "meta.num_temporal_layers": layers,
               "meta.init_seed": init_seed, "meta.state_checksum": state_checksum(sd)})
    if not uncertain:
        fx["meta.uncertain"] = 0
        assert out["loc"].shape[-1] == 2 and not any(k.startswith("decoder.scale") for k in sd)
    for k in ("loc", "pi", "reg_mask"):
        fx[f"out.{k}"] = out[k].numpy()
    fx["out.y_rot"] = data.y.numpy()
    for k in ("local_embed", "global_embed", "temporal_out"):
        fx[f"mid.{k}"] = caps[k].numpy()
    path = os.path.join(ROOT, "tests", "golden_grid", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **fx)
    print(f"{name}: N={batch.num_nodes} K={K} T={T} heads={heads} -> {os.path.getsize(path) / 1024:.0f} KiB; restatement deviation {dev}")


if __name__ == "__main__":
    if not R.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    for name in (sys.argv[1:] or list(CASES)):
        make(name)
