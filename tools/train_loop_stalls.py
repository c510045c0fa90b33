"""What the HOST side of a free-running training loop stalls on: per-step enqueue times of N pipelined steps (one synchronisation every
50), Python garbage-collection passes and their durations, allocator events.

    python tools/train_loop_stalls.py [--config config4] [--steps 400]

Finding of round 5 (HISTORY.md section 7 "Round 5"): no collection worth the name; the host runs ahead until a few thousand launches are in flight
(2 200 ... 6 300 by run) and is then held for 11-16 ms at a time -- back-pressure, which costs nothing while the GPU has queued work."""
import argparse
import gc
import os
import statistics
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from trajsde_amd import driver, runtime
    from trajsde_amd.data import TemporalData
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--steps", type=int, default=400)
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
    model = driver.build_model(cfg, None, dev, init_seed=0).train()
    flat = driver.FlatTraining(model)
    batch = synth(**spec["synth"]).to(dev)
    base = {k: v for k, v in batch.as_dict().items() if not k.startswith("_")}
    side = runtime.side_stream(dev)
    passes, t_gc = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            passes.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))
    gc.callbacks.append(on_gc)

    def fresh(i):
        with torch.cuda.stream(side):
            b = TemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in base.items()})
            model.prefetch_graph(b, NoiseSpec(seed=100 + i))
        return b
    nxt = fresh(0)
    enq, allocs = [], []
    t_all = time.perf_counter()
    for i in range(a.steps):
        t0 = time.perf_counter()
        flat.zero()
        model.training_step(nxt, i, noise=NoiseSpec(seed=100 + i)).backward()
        flat.step()
        nxt = fresh(i + 1)
        enq.append((time.perf_counter() - t0) * 1e3)
        allocs.append(torch.cuda.memory_stats()["num_device_alloc"])
        if i % 50 == 49:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / a.steps * 1e3
    late = [(round(v, 2), i) for i, v in enumerate(enq) if i > 0 and v > 3 * statistics.median(enq)]
    print(f"{a.config}: {a.steps} steps, {wall:.2f} ms a step; enqueue median {statistics.median(enq):.2f} ms, first step {enq[0]:.1f} ms")
    print("steps whose enqueue took more than 3 x the median (ms, step):", late[:12])
    if len(late) > 2:
        gaps = [b[1] - a_[1] for a_, b in zip(late, late[1:])]
        print("steps between them:", gaps[:12])
    print("garbage-collection passes by generation (count, longest ms):",
          {g: (sum(1 for p in passes if p[0] == g), round(max([p[1] for p in passes if p[0] == g] or [0.0]), 2)) for g in (0, 1, 2)})
    print("device allocations after the first step:", allocs[-1] - allocs[0])


if __name__ == "__main__":
    main()
