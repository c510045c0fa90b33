"""Generate tests/golden_metrics/metrics.npz: the REFERENCE's own metric classes (metrics/ade_t.py, fde_t.py, mr_t.py, over
the torchmetrics stand-in) evaluated on seeded random predictions.  Build container only.

    python oracle/make_golden_metrics.py

Two update() calls per metric object (accumulation), both `dataset` switches (best mode by ADE for 'nuScenes', by FDE at
the per-source end index for 'Argoverse'), mixed sources ordered by source as the reference assumes (its
repeat_interleave over counts, ade_t.py:55-56), agents without any valid step, agents whose end index is masked.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import ref_loader as R                                        # noqa: E402


def batch(g, K, A, T, n_src0):
    target = torch.randn(A, T, 2, generator=g).cumsum(1)
    pred = target.unsqueeze(0) + torch.randn(K, A, T, 2, generator=g) * torch.linspace(0.2, 3.0, T).view(1, 1, T, 1)
    mask = torch.rand(A, T, generator=g) > 0.25
    mask[3] = False                                           # an agent with no valid step at all
    mask[5, -1] = False                                       # end index masked
    mask[A - 2, 29] = False
    source = torch.cat([torch.zeros(n_src0, dtype=torch.long), torch.ones(A - n_src0, dtype=torch.long)])
    return pred, target, mask, source


def main():
    if not R.reference_available():
        sys.exit("reference tree not found")
    R._install_paths()
    from importlib.machinery import SourceFileLoader
    K, A, T = 6, 24, 60
    g = torch.Generator().manual_seed(77)
    b1, b2 = batch(g, K, A, T, 10), batch(g, K, A, T, 15)
    fx = {}
    for i, b in enumerate((b1, b2)):
        for nm, v in zip(("pred", "target", "mask", "source"), b):
            fx[f"in{i}.{nm}"] = v.numpy()
    with R.reference_cwd():
        for ds in ("nuScenes", "Argoverse"):
            for cls, path in (("ADE_T", "metrics/ade_t.py"), ("FDE_T", "metrics/fde_t.py"), ("MR_T", "metrics/mr_t.py")):
                M = getattr(SourceFileLoader(cls, path).load_module(cls), cls)
                m = M(dataset=ds, end_idcs=[59, 29], sources=[0, 1])
                m.update(*[t.clone() for t in b1])
                fx[f"out.{ds}.{cls}.after1"] = np.float64(float(m.compute()))
                m.update(*[t.clone() for t in b2])
                fx[f"out.{ds}.{cls}.after2"] = np.float64(float(m.compute()))
                print(ds, cls, fx[f"out.{ds}.{cls}.after1"], fx[f"out.{ds}.{cls}.after2"])
    path = os.path.join(ROOT, "tests", "golden_metrics", "metrics.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **fx)
    print(f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
