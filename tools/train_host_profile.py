"""Where the HOST time of one training step goes (cProfile over a few steps of BASELINE config 2).

    python tools/train_host_profile.py [--steps 8]
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
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--config", default="config2")
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

    def step(i):
        flat.zero()
        batch.y = y0
        loss = model.training_step(batch, i, noise=NoiseSpec(seed=100 + i))
        loss.backward()
        flat.step()

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for i in range(a.steps):
        step(10 + i)
    pr.disable()
    torch.cuda.synchronize()
    print(f"wall {1e3 * (time.perf_counter() - t0) / a.steps:.2f} ms/step (profiled)")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
    st.print_callers("method .to. of")


if __name__ == "__main__":
    main()
