"""One-off stress measurement in the shape of BASELINE.json configs[4] (1024 agents per scene, K=20, 50 SDE steps):
forward only, fp32 state, one GPU.  Prints edge counts, peak memory and ms per forward.

    python tools/stress_forward.py [--scenes 4] [--agents 1024] [--modes 20] [--steps 50] [--iters 3]
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
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--modes", type=int, default=20)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--box", type=float, default=400.0)
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = a.modes, a.steps
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=T / 10.0)
    dev = torch.device("cuda:0")
    model = driver.build_model(cfg, None, dev, init_seed=0)
    model.encoder.capture_intermediates = False
    batch = synth(S=a.scenes, n=a.agents, L=64, F=T, box=a.box, seed=9, mixed_source=True).to(dev)
    y0 = batch.y.clone()
    times = []
    with torch.no_grad():
        for i in range(a.iters + 1):
            batch.y = y0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model(batch, noise=NoiseSpec(seed=1 + i))
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
    gc = batch["_trajsde_graph"].graph
    print(json.dumps({"scenes": a.scenes, "agents_per_scene": a.agents, "num_modes": K, "future_steps": T,
                      "sample_paths": K * a.scenes * a.agents, "E_aa": gc.E_aa, "E_global": gc.E_g, "E_lane": gc.E_la,
                      "ms_per_forward": min(times[1:]), "first_ms": times[0], "scenes_per_s": a.scenes / min(times[1:]) * 1e3,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "finite": bool(torch.isfinite(out["loc"]).all())}))


if __name__ == "__main__":
    main()
