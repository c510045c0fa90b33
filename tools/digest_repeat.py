"""Run tests/grad_digest_child.py several times (fresh processes, optional environment per run) and report which gradient digests differ
between runs: the bit-reproducibility check of a full-size training step across processes.

    python tools/digest_repeat.py [n_plain] [n_tight]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "grad_digest_child.py"), "config2"], env=dict(os.environ, **extra),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    n_plain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_tight = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    runs = [("plain", run({})) for _ in range(n_plain)]
    runs += [("tight", run({"TRAJSDE_REDUCE_CAP": "600", "TRAJSDE_VPART_ARENA": "300000"})) for _ in range(n_tight)]
    for tag in sys.argv[3:]:                              # workspaces pre-filled: "zero", "nan" or a seed for random bits
        runs.append(("poison " + tag, run({"TRAJSDE_TEST_POISON": tag})))
    ref = runs[0][1]
    for i, (tag, r) in enumerate(runs[1:], 1):
        bad = [k for k in ref["digests"] if repr(ref["digests"][k]) != repr(r["digests"][k])]
        print(f"run {i} ({tag}): loss equal {ref['loss'] == r['loss']}, {len(bad)} of {len(ref['digests'])} digests differ")
        for k in bad[:12]:
            a, b = ref["digests"][k], r["digests"][k]
            print(f"     {k}: norm {a[0]!r} / {b[0]!r}   proj {a[1]!r} / {b[1]!r}")


if __name__ == "__main__":
    main()
