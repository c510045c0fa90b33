"""BASELINE.json configs[4]: the stress shape -- 1024 agents per scene, K=20 modes, 50 SDE steps (51 Euler steps), hidden
state stored bf16 between kernels -- forward only, on one GPU (the 8-GPU run is 8 of these, scenes sharded, no collective).

    python tools/stress_forward.py [--scenes 8] [--agents 1024] [--modes 20] [--steps 50] [--iters 5] [--storage bf16|fp32|both]

Prints one JSON line: per storage type ms per forward, scenes/s, peak memory, the per-kernel device times (HIP events inside
the library) and the HBM view of the two kernels the storage type matters for -- the global attention (streams the
relative-pose rows of every edge once per layer) and the step-granular SDE step at this configuration's K*N rows (the
north star's "HBM roofline in the SDE step": 512 B per path-step with an fp32 state, 256 B with a bf16 one).  With
`--check` the bf16 forward is compared with the fp32 one on the same Philox seed (what the storage type costs in accuracy).
Under rocprofv3 (--pmc FETCH_SIZE / WRITE_SIZE, --kernel-trace --stats) this is the command of the profiles/r02_stress_* files.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS = 8000.0


def main():
    import yaml
    from trajsde_amd import _lib, driver, runtime
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.schedule import decoder_schedule
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    spec = CONFIGS["config5"]
    ap.add_argument("--scenes", type=int, default=spec["synth"]["S"])
    ap.add_argument("--agents", type=int, default=spec["synth"]["n"])
    ap.add_argument("--modes", type=int, default=spec["num_modes"])
    ap.add_argument("--steps", type=int, default=spec["future_steps"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--box", type=float, default=spec["synth"]["box"])
    ap.add_argument("--storage", default="both", choices=["bf16", "fp32", "both"])
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T = a.modes, a.steps
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=T / 10.0)
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    model = driver.build_model(cfg, None, dev, init_seed=0)
    batch = synth(S=a.scenes, n=a.agents, L=spec["synth"]["L"], F=T, box=a.box, seed=spec["synth"]["seed"], mixed_source=True).to(dev)
    y0 = batch.y.clone()
    N = int(batch["x"].shape[0])
    rows = K * N
    tab = np.ascontiguousarray(decoder_schedule(T, T / 10.0).step_table())
    res, outs = {"scenes": a.scenes, "agents_per_scene": a.agents, "num_modes": K, "future_steps": T, "euler_steps": int(tab.shape[0]),
                 "sample_paths": rows}, {}
    for storage in (["fp32", "bf16"] if a.storage == "both" else [a.storage]):
        runtime.set_state_storage(storage)
        torch.cuda.reset_peak_memory_stats()
        times = []
        with torch.no_grad():
            for i in range(a.iters + 1):
                batch.y = y0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = model(batch, noise=NoiseSpec(seed=11))
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
            outs[storage] = {k: out[k].float().cpu() for k in ("loc", "pi")}
            lib.trajsde_profile_mode(2)
            for i in range(3):
                batch.y = y0
                model(batch, noise=NoiseSpec(seed=11))
            torch.cuda.synchronize()
            lib.trajsde_profile_mode(0)
        prof = _lib.profile_report()
        gc = batch["_trajsde_graph"].graph
        elem = 2 if storage == "bf16" else 4
        kern = {tag: round(ms / 3, 4) for tag, (n, ms, dom) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]}
        ga_ms = prof.get("k_global_attn<8>", (0, 0.0, False))[1] / 9                     # 3 layers x 3 forwards
        # global attention: algorithmic HBM bytes per layer = one rel row per edge (64 elements) + the node rows q, k_node, v_node
        # in and agg out (fp32); the gathered k_node / v_node rows of the senders are L2 / Infinity-Cache traffic
        ga_bytes = gc.E_g * 64 * elem + 4 * N * 256
        # step-granular SDE step at this configuration's K*N rows
        ya = torch.randn(rows, 64, device=dev).to(torch.bfloat16 if storage == "bf16" else torch.float32)
        yb = torch.empty_like(ya)
        nz = _lib.Noise(C.c_uint64(7), None, None)
        cur = torch.cuda.current_stream().cuda_stream
        dblob = model.decoder._rt.blob()

        def steps(n):
            for k in range(n):
                e = tab[k % tab.shape[0]].ctypes.data_as(C.POINTER(C.c_float))
                src, dst = (ya, yb) if k % 2 == 0 else (yb, ya)
                _lib.check(lib.trajsde_sde_step(rows, dblob.data_ptr(), src.data_ptr(), dst.data_ptr(), e, k, C.byref(nz), cur))
        steps(10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        steps(60)
        e1.record()
        torch.cuda.synchronize()
        sms = e0.elapsed_time(e1) / 60
        res[storage] = {"ms_per_forward": float(np.median(times[1:])), "first_ms": times[0],
                        "scenes_per_s": a.scenes / float(np.median(times[1:])) * 1e3, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
                        "E_aa": gc.E_aa, "E_global": gc.E_g, "E_lane": gc.E_la, "kernel_ms_per_forward": kern,
                        "global_attention_hbm": {"ms_per_layer": ga_ms, "algorithmic_bytes_per_layer": int(ga_bytes),
                                                 "achieved_GBps": ga_bytes / (ga_ms * 1e-3) / 1e9 if ga_ms else None,
                                                 "frac_of_8TBps": ga_bytes / (ga_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ga_ms else None},
                        "sde_step_hbm": {"rows": rows, "bytes_per_path_step": 128 * elem, "avg_launch_ms": sms,
                                         "achieved_GBps": rows * 128 * elem / (sms * 1e-3) / 1e9,
                                         "frac_of_8TBps": rows * 128 * elem / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "TFLOPps": rows * 41.8e3 / (sms * 1e-3) / 1e12},
                        "finite": bool(torch.isfinite(out["loc"]).all()),
                        # accuracy class of this storage type against the CPU oracle (what tests/test_gpu_parity.py pins):
                        # fp32 storage is the 1e-4 north-star path; bf16 storage rounds every intra-stage row to 8 mantissa bits
                        "parity_level": {"fp32": {"max_abs_loc_vs_oracle_bound": 1e-4, "class": "north-star parity (fp32 state)"},
                                         "bf16": {"max_abs_loc_vs_oracle_bound": 5e-2, "class": "bf16 state storage: ~200x looser than the "
                                                  "fp32-state path (measured 2.1e-2 on 3.8 m trajectories at this shape); inference only"}}[storage]}
        del ya, yb
    runtime.set_state_storage("fp32")
    _lib.check_range()
    if a.check and len(outs) == 2:
        res["bf16_vs_fp32"] = {"max_abs_loc_xy": float((outs["bf16"]["loc"][..., :2] - outs["fp32"]["loc"][..., :2]).abs().max()),
                               "max_abs_scale": float((outs["bf16"]["loc"][..., 2:] - outs["fp32"]["loc"][..., 2:]).abs().max()),
                               "max_abs_pi": float((outs["bf16"]["pi"] - outs["fp32"]["pi"]).abs().max()),
                               "loc_xy_scale": float(outs["fp32"]["loc"][..., :2].abs().max())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
