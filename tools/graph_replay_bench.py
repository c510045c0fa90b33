"""Eager launches vs HIP-graph replay of the inference forward at the metric workload (32 scenes x 256 agents, K=6, 20 steps).

    python tools/graph_replay_bench.py [--streams 3] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yaml
    from trajsde_amd import driver, runtime
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="metric256")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    spec = CONFIGS[a.config]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = spec["num_modes"], spec["future_steps"]
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=spec["max_fut_t"])
    dev = torch.device("cuda:0")
    model = driver.build_model(cfg, None, dev, init_seed=0).eval()
    cpu = synth(**spec["synth"])
    out = {"config": a.config, "scenes": spec["synth"]["S"], "steps": a.steps}
    for ns in sorted({1, a.streams}):
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        batches, y0 = [], []
        for st in streams:
            with torch.cuda.stream(st):
                b = cpu.to(dev)
                batches.append(b)
                y0.append(b.y.clone())
        torch.cuda.synchronize()

        def eager(i):
            k = i % ns
            with torch.cuda.stream(streams[k]), torch.no_grad():
                batches[k].y = y0[k]
                model(batches[k], noise=NoiseSpec(seed=1000 + i))

        graphs = []
        for k in range(ns):
            with torch.cuda.stream(streams[k]):
                batches[k].y = y0[k]
                graphs.append(runtime.GraphedForward(model, batches[k]))
        torch.cuda.synchronize()

        def replay(i):
            k = i % ns
            with torch.cuda.stream(streams[k]):
                graphs[k](seed=1000 + i)

        for name, fn in (("eager", eager), ("graph", replay)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            best = None
            for w in range(5):
                t0 = time.perf_counter()
                for i in range(a.steps):
                    fn(10 + i)
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                if best is None or el < best[0]:
                    best = (el, t_enq)
            out[f"{name}_streams{ns}"] = {"ms_per_forward": 1e3 * best[0] / a.steps, "host_enqueue_ms_per_forward": 1e3 * best[1] / a.steps,
                                          "scenes_per_s": spec["synth"]["S"] * a.steps / best[0]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
