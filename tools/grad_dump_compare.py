"""Cross-process comparison of the gradients of one full-size training step, element by element: which parameters differ between
two fresh processes (optionally with the allocator pre-filled, TRAJSDE_TEST_POISON), in which elements and by how much.

    python tools/grad_dump_compare.py [tag ...]      tags: plain | zero | nan | <seed>   (first run is the reference)"""
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, path):
    env = dict(os.environ, TRAJSDE_TEST_DUMP=path)
    if tag != "plain":
        env["TRAJSDE_TEST_POISON"] = tag
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "grad_digest_child.py"), "config2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(path)


def main():
    tags = sys.argv[1:] or ["plain", "plain", "zero", "zero"]
    with tempfile.TemporaryDirectory() as d:
        grads = [run(t, os.path.join(d, f"g{i}.pt")) for i, t in enumerate(tags)]
    ref = grads[0]
    for i in range(1, len(grads)):
        print(f"== run {i} ({tags[i]}) against run 0 ({tags[0]})")
        for n in ref:
            a, b = ref[n], grads[i][n]
            if torch.equal(a, b):
                continue
            diff = (a != b)
            idx = diff.nonzero()
            rel = ((a - b).abs() / a.abs().clamp_min(1e-30))[diff]
            rows = sorted(set(int(x[0]) for x in idx)) if a.dim() == 2 else []
            cols = sorted(set(int(x[1]) for x in idx)) if a.dim() == 2 else sorted(set(int(x[0]) for x in idx))
            print(f"  {n} {tuple(a.shape)}: {int(diff.sum())} elements differ, max rel {float(rel.max()):.2e}, median rel {float(rel.median()):.2e}; "
                  f"rows {rows[:10]}{'...' if len(rows) > 10 else ''} ({len(rows)}) cols {cols[:10]}{'...' if len(cols) > 10 else ''} ({len(cols)})")


if __name__ == "__main__":
    main()
