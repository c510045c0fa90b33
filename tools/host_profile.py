"""Where the HOST time of one forward goes (cProfile over a few forwards of BASELINE config 2).

    python tools/host_profile.py [--steps 30]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yaml
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    spec = CONFIGS["config2"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = spec["num_modes"], spec["future_steps"]
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=spec["max_fut_t"])
    dev = torch.device("cuda:0")
    model = driver.build_model(cfg, None, dev, init_seed=0).eval()
    batch = synth(**spec["synth"]).to(dev)
    y0 = batch.y.clone()

    def step(i):
        batch.y = y0
        with torch.no_grad():
            return model(batch, noise=NoiseSpec(seed=i))

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    # host-only cost: enqueue without waiting (the one sync inside graph prep stays)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(10 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / a.steps:.3f} ms/forward, drained after {1e3 * (t2 - t0) / a.steps:.3f} ms/forward")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        step(100 + i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
