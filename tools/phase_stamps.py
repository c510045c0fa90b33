"""Per-phase cycles of a stamped kernel (diagnostic build of the library, csrc/stamps.hpp):

    tools/build_variant.sh stamps "-DTSDE_STAMPS"
    TRAJSDE_LIB=$PWD/trajsde_amd/variants/stamps.so python tools/phase_stamps.py recur        # the encoder's cooperative recurrence
    TRAJSDE_LIB=$PWD/trajsde_amd/variants/stamps.so python tools/phase_stamps.py sde_step     # trajsde_sde_step at 786 432 rows

Prints cycles per unit (s_memtime of the stamped waves / units those waves report) and the share of every phase, plus the in-kernel
clock (cycles per 100 MHz tick).  A stamped build runs slower than the shipped one: read the shares."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TABLES = {
    "recur": (16, "tile-iterations", ["P1 first layers f | g", "  barrier", "P2 second layers, head dot", "  barrier",
                                      "P3 drift out, EM update", "  barrier", "P4 GRU gates first layers", "  barrier",
                                      "P5 gates, r*h", "  barrier", "P6 candidate first layer", "  barrier",
                                      "P7 candidate out, blend", "  barrier", "top: x_t loads, noise, biases", "-"]),
    "gattn": (8, "16-edge tiles", ["issue the loads of the tiles ahead", "wait for this tile's rel / k_node rows", "splits + stage writes",
                                   "P1: fragment reads + 12 matrix instr.", "softmax", "P2: splits + 16 matrix instr.", "loop overhead",
                                   "per-target epilogue (W_ve, store)"]),
    "gh3": (8, "16-edge tiles", ["loop overhead", "wait for this tile's rows (sc: rel rows + indices)", "h3: P1 | sc: parking + requests of the tiles ahead",
                                 "h3: v rows, softmax, split | sc: P1", "h3: P2 | sc: softmax", "h3: loop exit | sc: P2",
                                 "per-target prologue + epilogue (sc: and the unit's staging)", "-"]),
    "tail": (8, "16-edge tiles", ["request the loads (geometry, target rows, scalars)", "first layers of the two branches", "two products, LayerNorm, ReLU",
                                  "third product, LayerNorm", "d emb: two adjoint products on the target rows", "LayerNorm backwards, W2^T adjoint, mask",
                                  "three slabs out (whole rows)", "-"]),
    "gmf": (8, "16-edge tiles", ["wait for the rel rows, stage", "first product (16 matrix instr.)", "wait for key rows, node logits",
                                 "softmax scalars", "rescale", "second product (16 matrix instr.)", "wait for value rows, node sums",
                                 "per-target prologue / epilogue"]),
    "sde_step": (8, "tile-steps", ["wait for the state rows", "noise: Philox + Box-Muller (16 normals / lane)",
                                   "first layers: split + 48 matrix instr.", "32 tanh / lane", "drift: layer 2 + tanh + layer 3",
                                   "diffusion: layer 2 + tanh + head + sigmoid", "update + store", "-"]),
}


def read(lib, name, n, reset):
    fn = getattr(lib, f"trajsde_debug_stamps_{name}")
    fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
    buf = (C.c_ulonglong * (n + 4))()
    assert fn(buf, 1 if reset else 0) == n
    return list(buf)


def forward_runner():
    import yaml
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    spec = CONFIGS[os.environ.get("WORKLOAD", "metric256")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = spec["num_modes"], spec["future_steps"]
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=spec["max_fut_t"])
    dev = torch.device("cuda:0")
    model = driver.build_model(cfg, None, dev, init_seed=0)
    batch = synth(**spec["synth"]).to(dev)
    y0 = batch.y.clone()

    def run(i):
        with torch.no_grad():
            batch.y = y0
            model(batch, noise=NoiseSpec(seed=i))
    return run


def train_runner():
    """one training step (forward + backward) at 64 x 128 agents per call: the backward kernels' tables (tail)"""
    import yaml
    from trajsde_amd import driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    spec = CONFIGS[os.environ.get("WORKLOAD", "config2")]
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
    y0 = batch.y.clone()

    def run(i):
        flat.zero()
        batch.y = y0
        model.training_step(batch, i, noise=NoiseSpec(seed=100 + i)).backward()
    run.keep = (model, flat)
    return run


def sde_step_runner():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sde_step_bench
    return sde_step_bench.make_runner(int(os.environ.get("ROWS", "786432")))


def main(name):
    from trajsde_amd import _lib
    lib = _lib.lib()
    n, unit, labels = TABLES[name]
    run = sde_step_runner() if name == "sde_step" else (train_runner() if name == "tail" else forward_runner())
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    read(lib, name, n, True)
    for i in range(10):
        run(10 + i)
    torch.cuda.synchronize()
    t = read(lib, name, n, True)
    waves, units, ticks = t[n], t[n + 1], t[n + 2]
    tot = sum(t[:n])
    res = {"kernel": name, "stamped_waves": waves, "units": units, "cycles_per_unit": tot / max(units, 1),
           "in_kernel_clock_ghz": tot / ticks * 0.1 if ticks else None, "phases": {}}
    print(f"# {name}: {waves} stamped waves, {units} {unit}, {tot / max(units, 1):.0f} cycles per unit, in-kernel clock "
          f"{res['in_kernel_clock_ghz']:.3f} GHz")
    for i in range(n):
        if labels[i] == "-":
            continue
        res["phases"][labels[i].strip() + f" [{i}]"] = {"cycles_per_unit": t[i] / max(units, 1), "share": t[i] / tot}
        print(f"  {labels[i]:36s} {t[i] / max(units, 1):8.0f}  {100 * t[i] / tot:5.1f} %")
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "recur")
