"""Generate tests/golden_data/mixds.npz: raw per-scene tensors in the two stored layouts (nuScenes 5+12 slots,
Argoverse 20+30 slots) and what the REFERENCE's `nuArgoDataset.get` (dataset/nuScenes_Argoverse/
nuScenes_Argoverse.py:140-232, run where it lies over oracle/shims) returns for them -- split `val`, split
`train` with seeded flips, and `is_gtabs: false`.  Build-container only; the fixture holds data only.

    python oracle/make_golden_dataset.py
"""
import json
import os
import pickle
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import ref_loader as R                                        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_data", "mixds.npz")
FLIP_SEEDS = (0, 1, 2, 3)


def raw_scene(rng, source, n, lanes, seq_id):
    """A stored scene with the keys the two preprocessors write (nuScenes_hivt.py:258-286,
    Argoverse_abs.py:252-278); values are random, shapes and dtypes are the real ones."""
    past, fut = (5, 12) if source == 0 else (20, 30)
    f32 = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    pad = torch.from_numpy(rng.random((n, past + fut)) < 0.25)
    pad[:, past - 1] = False
    bos = torch.zeros(n, past, dtype=torch.bool)
    bos[:, 0] = ~pad[:, 0]
    bos[:, 1:] = pad[:, :past - 1] & ~pad[:, 1:past]
    src, dst = np.nonzero(~np.eye(n, dtype=bool))
    n_la = int(rng.integers(1, lanes * n))
    sc = dict(
        x=f32(n, past, 2), positions=f32(n, past + fut, 2) * 20, y=f32(n, fut, 2),
        edge_index=torch.from_numpy(np.stack([src, dst])).long(), num_nodes=n,
        padding_mask=pad, bos_mask=bos, rotate_angles=f32(n),
        lane_positions=f32(lanes, 6, 2) * 30, lane_vectors=f32(lanes, 2),
        lane_paddings=torch.from_numpy((rng.random((lanes, 6)) < 0.2).astype(np.float32)), lane_lengths=f32(lanes).abs(),
        lane_actor_index=torch.from_numpy(np.stack([rng.integers(0, lanes, n_la), rng.integers(0, n, n_la)])).long(),
        lane_actor_vectors=f32(n_la, 2),
        goal_idcs=torch.from_numpy(rng.integers(0, lanes, n)).long(), has_goal=torch.from_numpy(rng.random(n) < 0.5),
        seq_id=seq_id, av_index=0, agent_index=int(rng.integers(0, n)),
        origin=f32(1, 2) * 100, theta=f32(1)[0],
    )
    if source == 0:
        sc.update(category=torch.from_numpy(rng.integers(0, 11, n)).long(),
                  lane_rotate_angles=f32(lanes), lane_edge_index=torch.zeros(2, 0, dtype=torch.long),
                  lane_edge_type=torch.zeros(0, dtype=torch.long))
    else:
        sc.update(is_intersections=torch.from_numpy(rng.integers(0, 2, lanes)).to(torch.uint8),
                  turn_directions=torch.from_numpy(rng.integers(0, 3, lanes)).to(torch.uint8),
                  traffic_controls=torch.from_numpy(rng.integers(0, 2, lanes)).to(torch.uint8), city="PIT")
    return sc


def flat(prefix, sc, out):
    for k, v in sc.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            out[f"{prefix}/{k}"] = v.numpy()
        elif isinstance(v, str):
            out[f"{prefix}/{k}"] = np.array(v)
        else:
            out[f"{prefix}/{k}"] = np.array(v)


def main():
    R._install_paths()
    rng = np.random.default_rng(2024)
    nus = [raw_scene(rng, 0, n, l, f"tok{i:02d}_smp{i:02d}") for i, (n, l) in enumerate([(4, 3), (7, 5), (3, 2)])]
    argo = [raw_scene(rng, 1, n, l, 1000 + i) for i, (n, l) in enumerate([(5, 4), (2, 3), (6, 6)])]

    tmp = tempfile.mkdtemp(prefix="mixds_")
    nu_dir, argo_dir = os.path.join(tmp, "nu"), os.path.join(tmp, "argo")
    with R.reference_cwd():
        from importlib.machinery import SourceFileLoader
        from models.utils.util import TemporalData as RefTemporalData
        mod = SourceFileLoader("nuArgoDataset", "dataset/nuScenes_Argoverse/nuScenes_Argoverse.py").load_module("nuArgoDataset")

    # the reference calls torch.load(path) on pickled scene objects (MIXDS:141)
    _load = torch.load
    torch.load = lambda p, *a, **k: _load(p, *a, **{**k, "weights_only": False})

    tokens = {}
    for split_dir in ("train", "val"):
        os.makedirs(os.path.join(nu_dir, split_dir), exist_ok=True)
        tokens[split_dir] = [s["seq_id"] for s in nus]
        for s in nus:
            torch.save(RefTemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in s.items()}),
                       os.path.join(nu_dir, split_dir, s["seq_id"] + ".pt"))
    os.environ["TRAJSDE_FAKE_NUSCENES_SPLITS"] = json.dumps(tokens)
    os.makedirs(os.path.join(argo_dir, "train"), exist_ok=True)
    raw_names = [f"{s['seq_id']}.csv" for s in argo]
    proc_names = [f"{s['seq_id']}.pt" for s in argo]
    proc_paths = [os.path.join(argo_dir, "train", p) for p in proc_names]
    for s, p in zip(argo, proc_paths):
        torch.save(RefTemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in s.items()}), p)
    with open(os.path.join(argo_dir, "raw_processed_fns_train.pt"), "wb") as f:     # MIXDS:63-66
        pickle.dump((raw_names, proc_names, proc_paths), f)

    out = {}
    for i, s in enumerate(nus):
        flat(f"raw/nus/{i}", s, out)
    for i, s in enumerate(argo):
        flat(f"raw/argo/{i}", s, out)

    def dump(tag, ds):
        for i in range(ds.len()):
            d = ds.get(i)
            flat(f"{tag}/{i}", {k: d[k] for k in d.keys}, out)

    base = dict(nus=True, Argo=True, type="grid", is_gtabs=True, random_flip=False)
    dump("val", mod.nuArgoDataset("val", None, None, nu_dir, argo_dir, spec_args=base))
    dump("val_rel", mod.nuArgoDataset("val", None, None, nu_dir, argo_dir, spec_args={**base, "is_gtabs": False}))
    dump("val_argo_only", mod.nuArgoDataset("val", None, None, nu_dir, argo_dir, spec_args={**base, "nus": False}))
    train = mod.nuArgoDataset("train", None, None, nu_dir, argo_dir, spec_args={**base, "random_flip": True})
    for seed in FLIP_SEEDS:
        random.seed(seed)
        dump(f"train_seed{seed}", train)
    out["meta/flip_seeds"] = np.array(FLIP_SEEDS)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
