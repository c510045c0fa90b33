"""child process of test_gpu_backward.py::test_full_size_training_step_agrees_between_kernel_forms: one training step at a BASELINE
configuration with the library's run-time switches taken from the environment; prints one JSON line {loss, digests: {param: [norm, projection]}}"""
import json
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import yaml
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    spec = CONFIGS[sys.argv[1]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = spec["num_modes"], spec["future_steps"]
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=spec["max_fut_t"])
    dev = torch.device("cuda:0")
    model = driver.build_model(cfg, None, dev, init_seed=0).train()
    batch = synth(**spec["synth"]).to(dev)
    poison = os.environ.get("TRAJSDE_TEST_POISON")             # what the workspaces hold before the kernels write them
    if poison:
        junk = torch.empty(3 << 30, device=dev, dtype=torch.float32)          # 12 GB: the step's allocations are carved from this block
        if poison == "nan":
            junk.fill_(float("nan"))
        elif poison == "zero":
            junk.zero_()
        else:
            junk.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1, generator=torch.Generator(device=dev).manual_seed(int(poison)))
        del junk
    loss = model.training_step(batch, 0, noise=NoiseSpec(seed=100, dropout_seed=101))
    loss.backward()
    torch.cuda.synchronize()
    dump = os.environ.get("TRAJSDE_TEST_DUMP")                 # every gradient, word for word, for a cross-process comparison
    if dump:
        torch.save({n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}, dump)
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().reshape(-1).cpu()
        gen = torch.Generator().manual_seed(zlib.crc32(n.encode()))
        signs = (torch.randint(0, 2, (g.numel(),), generator=gen) * 2 - 1).double()
        out[n] = [float(g.norm()), float((g * signs).sum()), bool(torch.isfinite(g).all())]
    print(json.dumps({"loss": float(loss.detach()), "digests": out}))


if __name__ == "__main__":
    main()
