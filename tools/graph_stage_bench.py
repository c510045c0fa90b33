"""Time the graph stage alone (trajsde_graph_prepare_async + trajsde_graph_compact through runtime.GraphContext) on the metric
workload, with HIP events, and print the library's per-kernel table for it.

    python tools/graph_stage_bench.py [--workload metric256] [--iters 50] [--exact]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from trajsde_amd import _lib, runtime
    from trajsde_amd.runtime import GraphContext, NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="metric256")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--exact", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    batch = synth(**CONFIGS[a.workload]["synth"]).to(dev)
    rot, y_rot = runtime.rotate_inputs(batch)
    batch["rotate_mat"] = rot

    def once(i):
        if GraphContext.KEY in batch:
            del batch[GraphContext.KEY]
        return GraphContext.get(batch, 50.0, 21, NoiseSpec(seed=i), exact=a.exact)
    for i in range(5):
        gc = once(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        gc = once(10 + i)
    e1.record()
    torch.cuda.synchronize()
    print(f"graph stage ({'exact' if a.exact else 'sync-free'}): {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per batch, counts {gc.true_counts()}")
    _lib.lib().trajsde_profile_mode(2)
    for i in range(3):
        once(1000 + i)
    torch.cuda.synchronize()
    _lib.lib().trajsde_profile_mode(0)
    tab = _lib.profile_report()
    for tag, (n, ms, dom) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
        print(f"   {tag:24s} {ms / 3 * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
