"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference over
oracle/shims) on small synthetic batches with injected noise.  Build-container only.

    python oracle/make_golden.py            # rewrites every fixture

A fixture holds data only: the batch tensors, the seed of the build's deterministic weight init
(+ a checksum of the resulting state_dict), the Philox seed of the injected normals, and the
reference's outputs and intermediates.  No reference source travels.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import ref_loader as R                                        # noqa: E402
from trajsde_amd import philox                                # noqa: E402
from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet  # noqa: E402
from trajsde_amd.schedule import decoder_schedule            # noqa: E402
from trajsde_amd.synth import synth                          # noqa: E402

CASES = {
    # name: (synth kwargs, num_modes, future_steps, max_fut_t, init_seed, noise_seed)
    "mixed_k6_t20": (dict(S=2, n=6, L=5, F=20, box=100.0, seed=11, mixed_source=True, history_dropout=0.5), 6, 20, 2.0, 0, 101),
    "nus_k1_t5": (dict(S=1, n=12, L=8, F=5, box=60.0, seed=12, nus_sparsity=True, source=0), 1, 5, 0.5, 1, 102),
    "argo_k6_t30": (dict(S=3, n=10, L=8, F=30, box=120.0, seed=13, source=1, history_dropout=0.3), 6, 30, 3.0, 2, 103),
    "shipped_k10_t60": (dict(S=2, n=8, L=6, F=60, box=90.0, seed=14, mixed_source=True, history_dropout=0.3), 10, 60, 6.0, 3, 104),
    # `uncertain: False` (DEC:56, DEC:100-101): the decoder without its scale head, loc [K, N, T, 2]
    "plain_k3_t12": (dict(S=2, n=9, L=5, F=12, box=80.0, seed=16, mixed_source=True, history_dropout=0.3), 3, 12, 1.2, 6, 106, False),
}


def our_cfg(num_modes, future_steps, max_fut_t, uncertain=True):
    import yaml
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model_specific"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps)
    cfg["aggregator"]["kwargs"]["num_modes"] = num_modes
    cfg["decoder"]["kwargs"].update(num_modes=num_modes, future_steps=future_steps, max_fut_t=max_fut_t)
    if not uncertain:
        cfg["decoder"]["kwargs"]["uncertain"] = False
    return cfg


def state_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def philox_noise(seed, A, Nt, KN, n_euler, H=21):
    z_fake = philox.normals(seed, philox.STREAM_FAKE_AGENT, 0, np.arange(A), 64)[:, :H * 2].reshape(A, H, 2)
    z_enc = np.stack([philox.normals(seed, philox.STREAM_ENCODER, i, np.arange(Nt), 64) for i in range(H)])
    z_dec = np.stack([philox.normals(seed, philox.STREAM_DECODER, k, np.arange(KN), 64) for k in range(n_euler)])
    return z_fake, z_enc, z_dec


OOD_CASES = {
    "ood_k3_t5": (dict(S=2, n=7, L=5, F=5, box=70.0, seed=15, mixed_source=True, history_dropout=0.4), 3, 5, 0.5, 4, 105),
}


def make_ood(name):
    """MODEL:89-98 with ood=True: encoder.forward_ood (10 stochastic recurrences) -> stds."""
    skw, K, T, max_t, init_seed, noise_seed = OOD_CASES[name]
    batch = synth(**skw)
    ours = PredictionModelSDENet(**our_cfg(K, T, max_t), init_seed=init_seed)
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    ref = R.build_reference_model(R.load_reference_cfg(num_modes=K, future_steps=T, max_fut_t=max_t))
    ref.load_state_dict(sd)
    ref.ood = True
    N = batch.num_nodes
    sched = decoder_schedule(T, max_t)
    z_enc = [philox.normals(noise_seed, philox.STREAM_ENCODER, s, np.arange(N), 64) for s in range(10 * 21)]
    z_dec = [philox.normals(noise_seed, philox.STREAM_DECODER, k, np.arange(K * N), 64) for k in range(sched.n_euler)]
    caps = {}
    hook = ref.aggregator.register_forward_hook(lambda m, a, o: caps.__setitem__("global_embed", o))
    out, data, rec = R.run_reference_forward(ref, batch, replay=[torch.from_numpy(z) for z in z_enc + z_dec])
    hook.remove()
    assert len(rec) == 210 + sched.n_euler
    fx = {f"in.{k}": v.numpy() for k, v in batch.as_dict().items() if torch.is_tensor(v)}
    fx["meta.num_modes"], fx["meta.future_steps"], fx["meta.max_fut_t"] = K, T, max_t
    fx["meta.init_seed"], fx["meta.noise_seed"] = init_seed, noise_seed
    fx["meta.state_checksum"] = state_checksum(sd)
    fx["meta.n_euler"] = sched.n_euler
    for k in ("loc", "pi", "reg_mask", "stds"):
        fx[f"out.{k}"] = out[k].numpy()
    fx["mid.global_embed"] = caps["global_embed"].numpy()
    path = os.path.join(ROOT, "tests", "golden_ood", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **fx)
    print(f"{name}: N={N} K={K} T={T} -> {os.path.getsize(path) / 1024:.0f} KiB")


def make(name):
    skw, K, T, max_t, init_seed, noise_seed = CASES[name][:6]
    uncertain = CASES[name][6] if len(CASES[name]) > 6 else True
    batch = synth(**skw)
    ours = PredictionModelSDENet(**our_cfg(K, T, max_t, uncertain), init_seed=init_seed)
    sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    rcfg = R.load_reference_cfg(num_modes=K, future_steps=T, max_fut_t=max_t)
    rcfg["decoder"]["kwargs"]["uncertain"] = uncertain
    ref = R.build_reference_model(rcfg)
    ref.load_state_dict(sd)                                   # key-for-key: the state_dict contract of App. C

    N, A = batch.num_nodes, batch["agent_index"].numel()
    sched = decoder_schedule(T, max_t)
    z_fake, z_enc, z_dec = philox_noise(noise_seed, A, N + A, K * N, sched.n_euler)
    replay = [torch.from_numpy(z_fake)] + [torch.from_numpy(z) for z in z_enc] + [torch.from_numpy(z) for z in z_dec]

    caps = {"gru": []}
    hooks = [ref.encoder.register_forward_hook(lambda m, a, o: caps.__setitem__("local_embed", o[0])),
             ref.aggregator.register_forward_hook(lambda m, a, o: caps.__setitem__("global_embed", o)),
             ref.encoder.aa_encoder.register_forward_hook(lambda m, a, o: caps.__setitem__("aa_out", o)),
             ref.encoder.gru_unit.register_forward_hook(lambda m, a, o: caps["gru"].append(o))]
    out, data, rec = R.run_reference_forward(ref, batch, replay=replay)
    for h in hooks:
        h.remove()
    assert len(rec) == 1 + 21 + sched.n_euler, (len(rec), sched.n_euler)

    fx = {f"in.{k}": v.numpy() for k, v in batch.as_dict().items() if torch.is_tensor(v)}
    fx["meta.num_modes"], fx["meta.future_steps"], fx["meta.max_fut_t"] = K, T, max_t
    fx["meta.init_seed"], fx["meta.noise_seed"] = init_seed, noise_seed
    fx["meta.state_checksum"] = state_checksum(sd)
    fx["meta.n_euler"] = sched.n_euler
    if not uncertain:
        fx["meta.uncertain"] = 0
        assert out["loc"].shape[-1] == 2 and not any(k.startswith("decoder.scale") for k in sd)
    for k in ("loc", "pi", "reg_mask", "diff_in", "diff_out", "label_in", "label_out"):
        fx[f"out.{k}"] = out[k].numpy()
    fx["out.y_rot"] = data.y.numpy()
    fx["out.rotate_mat"] = data["rotate_mat"].numpy()
    fx["mid.local_embed"] = caps["local_embed"].numpy()
    fx["mid.global_embed"] = caps["global_embed"].numpy()
    fx["mid.aa_out"] = caps["aa_out"].view(21, N + A, 64).numpy()
    fx["mid.latent_ys"] = torch.stack(caps["gru"])[:, :N].numpy()
    for t in range(21):                                       # the encoder's side effect on the batch (ENC:107-110)
        fx[f"mid.edge_index_{t}"] = data[f"edge_index_{t}"].numpy()
        fx[f"mid.edge_attr_{t}"] = data[f"edge_attr_{t}"].numpy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: N={N} A={A} K={K} T={T} euler={sched.n_euler} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    if not R.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    for name in (sys.argv[1:] or list(CASES) + list(OOD_CASES)):
        make_ood(name) if name in OOD_CASES else make(name)
