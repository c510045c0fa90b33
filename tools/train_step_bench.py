"""Time `training_step + backward + AdamW` at a BASELINE.json synthetic configuration (default config2: 64 scenes x 128
agents, K=6, 20 steps) and print the library's per-kernel HIP-event table for one step.

    python tools/train_step_bench.py [--config config2] [--steps 5] [--warmup 2]
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
    from trajsde_amd import _lib, driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config2")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--pipeline", action="store_true",
                    help="as driver.train runs it: every step on a fresh copy of the batch, the NEXT step's copy rotated and run "
                         "through the graph stage on the side stream right after this step was enqueued (runtime.prefetch_graph)")
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
    flat = driver.FlatTraining(model)                  # what driver.train uses: AdamW over one flat parameter tensor
    batch = synth(**spec["synth"]).to(dev)
    y0 = batch.y.clone()

    from trajsde_amd import runtime
    from trajsde_amd.data import TemporalData
    side = runtime.side_stream(dev)
    base = {k: v for k, v in batch.as_dict().items() if not k.startswith("_")}
    base["y"] = y0
    nxt = [None]

    def fresh(i):                                      # under the side stream: device copies of the batch, then rotation + graph stage
        with torch.cuda.stream(side):
            b = TemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in base.items()})
            model.prefetch_graph(b, NoiseSpec(seed=100 + i))
        return b

    def step(i):
        flat.zero()
        if a.pipeline:
            cur = nxt[0] if nxt[0] is not None else fresh(i)
        else:
            cur = batch
            batch.y = y0                               # forward rotates y in place (MODEL:83-84)
        loss = model.training_step(cur, i, noise=NoiseSpec(seed=100 + i))
        loss.backward()
        flat.step()
        if a.pipeline:
            nxt[0] = fresh(i + 1)
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    enq = (time.perf_counter() - t0) / a.steps * 1e3            # host time to enqueue a step (the graph stage's one sync included)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.reset_peak_memory_stats()
    _lib.lib().trajsde_profile_mode(2)
    nxt[0] = None
    was_pipelined, a.pipeline = bool(a.pipeline), False
    step(999)
    torch.cuda.synchronize()
    table = _lib.profile_report()
    _lib.lib().trajsde_profile_mode(0)
    rows = sorted(((v[1], k, v[0]) for k, v in table.items()), reverse=True)
    print(json.dumps({"config": a.config, "pipelined": was_pipelined, "scenes": spec["synth"]["S"], "ms_per_train_step": ms, "host_enqueue_ms": enq,
                      "scenes_per_s": spec["synth"]["S"] / ms * 1e3, "loss": float(loss),
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))
    # roofline of the HBM-bound weight-gradient launch: the three embedding problems of every edge list (AA, AL, global)
    gc = batch["_trajsde_graph"]
    cnt = gc.true_counts()
    edges = cnt["E_aa"] + cnt["E_la"] + cnt["E_g"]
    if "k_wgrad[edge-embed]" in table:
        n, ms_w = table["k_wgrad[edge-embed]"][0], table["k_wgrad[edge-embed]"][1]
        by = edges * (512 + 272)                       # (round 4: the two branch problems share one read of their delta rows)
        print(json.dumps({"roofline_k_wgrad_edge_embed": {"bound": "hbm", "algorithmic_bytes_per_step": by, "ms_per_step": ms_w, "launches": n,
                                                          "achieved_GBps": by / (ms_w * 1e-3) * 1e-9, "peak_GBps": 8000.0,
                                                          "frac": by / (ms_w * 1e-3) * 1e-9 / 8000.0,
                                                          "rows": {"E_aa": cnt["E_aa"], "E_la": cnt["E_la"], "E_g": cnt["E_g"]},
                                                          "bytes_per_row": "(d e_pre, s) 2 x 256 B + (d s_pre 256 B + geometry 16 B) read once for both branches"}}))
    tot = sum(r[0] for r in rows)
    print(f"kernel time of one step (HIP events, serial): {tot:.2f} ms")
    for ms_k, name, n in rows[:int(os.environ.get("TSB_ROWS", "28"))]:
        print(f"  {ms_k:8.3f} ms  x{n:<4d} {name}")


if __name__ == "__main__":
    main()
