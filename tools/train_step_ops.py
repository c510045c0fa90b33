"""Which torch operators one training step issues (torch.profiler, CPU side: operator name, calls, where from).

    python tools/train_step_ops.py [--config config2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yaml
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
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
        return loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step(10)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=40,
                                                      max_src_column_width=90))


if __name__ == "__main__":
    main()
