"""Where an iteration of the fused edge attention spends its cycles (diagnostic build of the library with in-kernel stamps).

    tools/build_variant.sh stamps "-DTSDE_EDGE_STAMPS"
    TRAJSDE_LIB=$PWD/trajsde_amd/variants/stamps.so python tools/edge_phase_stamps.py

Runs a few one-stream forwards of the metric workload and prints, per phase of k_edge_attn2's loop body, the cycles per iteration
(s_memtime, summed over two waves of every workgroup -- one of each half -- and divided by the stamped iterations) next to the
phase's instruction counts (static, from the listing: tools/isa_loop_histogram.py).  The stamps drain the LDS queue at every
mark, so the build is slower than the shipped kernel; the shares are what it is for."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PHASES = ["loads + target changes (flush, q row)", "in2 A (rstd, operand, 8 mfma, relu)", "W_A product (split + 48 mfma)",
          "in2 B", "W_B product", "LayerNorm + ReLU of the sum", "W_2 product", "last LayerNorm", "lin_k | lin_v product (96 mfma)",
          "logits + online softmax"]


def main():
    import yaml
    from trajsde_amd import _lib, driver
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    lib = _lib.lib()
    fn = lib.trajsde_debug_edge_stamps
    fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
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
    buf = (C.c_ulonglong * 16)()
    with torch.no_grad():
        for i in range(3):
            batch.y = y0
            model(batch, noise=NoiseSpec(seed=i))
        torch.cuda.synchronize()
        assert fn(buf, 1) == 0
        n = 10
        for i in range(n):
            batch.y = y0
            model(batch, noise=NoiseSpec(seed=10 + i))
        torch.cuda.synchronize()
        assert fn(buf, 1) == 0
    iters = buf[10]
    tot = sum(buf[i] for i in range(10))
    res = {"stamped_wave_iterations": int(iters), "cycles_per_iteration": tot / iters,
           "phases": {PHASES[i]: {"cycles_per_iteration": buf[i] / iters, "share": buf[i] / tot} for i in range(10)}}
    ghz = tot / buf[11] * 0.1 if buf[11] else float("nan")               # s_memtime ticks per 100 MHz s_memrealtime tick
    res["in_kernel_clock_ghz"] = ghz
    print(f"# {iters} stamped wave-iterations, {tot / iters:.0f} cycles per iteration (s_memtime ticks); in-kernel clock {ghz:.3f} GHz")
    if buf[14]:
        res["loop_us"] = {"mean": buf[11] / buf[14] / 100.0, "max": buf[12] / 100.0, "min": buf[13] / 100.0}
        print(f"# loop time per stamped wave: mean {buf[11] / buf[14] / 100.0:.1f} us, fastest {buf[13] / 100.0:.1f} us, slowest {buf[12] / 100.0:.1f} us")
    for i in range(10):
        print(f"  {PHASES[i]:44s} {buf[i] / iters:8.0f}  {100 * buf[i] / tot:5.1f} %")
    try:                                                      # per-workgroup table of the last launch (diagnostic build only)
        fw = lib.trajsde_debug_edge_wg
        fw.argtypes, fw.restype = [C.POINTER(C.c_ulonglong)], C.c_int
        wg = (C.c_ulonglong * 2048)()
        assert fw(wg) == 0
        import numpy as np
        a = np.array(list(wg), dtype=np.float64).reshape(512, 4)
        a = a[a[:, 0] > 0]
        t = a[:, 0] / 100.0
        print(f"# per workgroup (wave 0, last launch): n {len(t)}, loop us min {t.min():.1f} p10 {np.percentile(t, 10):.1f} median {np.median(t):.1f} "
              f"p90 {np.percentile(t, 90):.1f} max {t.max():.1f}")
        for x in sorted(set(a[:, 2].astype(int) & 15)):
            m = (a[:, 2].astype(int) & 15) == x
            print(f"#   XCC {x}: n {int(m.sum())}, mean {t[m].mean():.1f} us, min {t[m].min():.1f}, max {t[m].max():.1f}, phase-0 cycles/iter {a[m, 1].mean() / (iters / buf[14]):.0f}")
        order = np.argsort(t)
        print("#   corr(loop time, phase-0 cycles) = %.3f" % np.corrcoef(t, a[:, 1])[0, 1])
        print("#   slowest 8 blocks:", [(int(i), round(float(t[i]), 1)) for i in order[-8:]], " fastest 8:", [(int(i), round(float(t[i]), 1)) for i in order[:8]])
    except Exception as e:
        print("# no per-workgroup table:", repr(e))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
