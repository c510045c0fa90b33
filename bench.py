"""bench.py -- scenes/sec of PredictionModelSDENet.forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]           # N=1: plain process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one forward pass (graph preparation + encoder + global interactor + SDE decoder, inference,
fp32, fresh Philox seed) over one synthetic batch of BASELINE config 2: 64 scenes x 128 agents, K=6,
20 SDE steps (SURVEY.md 8(d) generator), inputs resident in HBM before the timed region.  Scenes
shard over ranks with no data-path collective: every rank runs the same-sized batch
(weak scaling); the only collectives are the timing barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel (HIP events recorded
inside the library on the launch stream during the timed region) and the CPU baseline (the oracle --
oracle/restate.py, the bit-exact restatement of the reference -- timed on this box's host cores on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trajsde_amd import _lib  # noqa: E402
from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet  # noqa: E402
from trajsde_amd.runtime import NoiseSpec  # noqa: E402
from trajsde_amd.synth import CONFIGS, synth  # noqa: E402

WORKLOAD = "config2"                 # "Synthetic batch of 64 scenes x 128 agents, K=6, 20 steps, 1xMI355X inference-only"
FLOP_PER_EDGE = 41.7e3               # neighbour embed 25.1k + k,v 16.4k + dot 0.256k per (t, edge)
# The edge kernel evaluates its fp32 GEMMs as split-precision products on the 16-bit matrix cores (csrc/tile.hpp):
# three v_mfma_f32_16x16x32_f16 per fp32 product (fp16x3, the default build) or six ..._bf16 (bf16x6), fp32-accurate.
# Its roofline is therefore the dense fp16/bf16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s) divided by the number
# of products, expressed in algorithmic (fp32) FLOP/s.
MFMA_16BIT_PEAK_TFLOPS = 2500.0
SPLIT_PRODUCTS = _lib.lib().trajsde_split_products()
SPLIT_NAME = {3: "fp16x3", 6: "bf16x6"}[SPLIT_PRODUCTS]
PEAK_FP32_EQUIV_TFLOPS = MFMA_16BIT_PEAK_TFLOPS / SPLIT_PRODUCTS
CPU_SAMPLE_SCENES = 16


# static instruction stream of the pair edge kernel per 16-edge tile (hipcc -S of csrc/attn.hip, k_edge_kv2<768>, counted
# over the tile loop): VALU incl. the operand splits and LayerNorms, and the 16x16x32 matrix instructions
EDGE_TILE_VALU = {3: 552, 6: 1060}
EDGE_TILE_MFMA = {3: 120, 6: 240}


def issue_view(e_aa, avg_s):
    """What actually bounds the dominant kernel: a wave64 VALU instruction occupies its SIMD's VALU for 4 cycles, a
    16x16x32 matrix instruction the matrix pipe for 16; with 2-3 waves per SIMD the two barely overlap, so the sum of both
    per tile against the measured SIMD-cycles per tile (at the 2.4 GHz peak clock, 1024 SIMDs) is the fraction of the time
    the SIMDs spend issuing this kernel's own arithmetic."""
    tiles = e_aa / 16.0
    valu, mfma = EDGE_TILE_VALU[SPLIT_PRODUCTS] * 4, EDGE_TILE_MFMA[SPLIT_PRODUCTS] * 16
    measured = avg_s * 2.4e9 * 1024 / tiles
    return {"valu_cycles_per_tile": valu, "mfma_cycles_per_tile": mfma, "measured_simd_cycles_per_tile": measured,
            "frac": (valu + mfma) / measured, "clock_GHz": 2.4, "simds": 1024}


def build_cfg(spec):
    import yaml
    with open(os.path.join(ROOT, "trajsde_amd/configs/mi355x_sde_encoder_decoder.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T, mt = spec["num_modes"], spec["future_steps"], spec["max_fut_t"]
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["aggregator"]["kwargs"]["num_modes"] = K
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=mt)
    return cfg


def cpu_baseline(model, cfg, spec, gpu_loc_fn):
    """Oracle timed on the host cores: CPU_SAMPLE_SCENES scenes of the same generator, 1 warm-up + 3 timed
    forwards; also the 'minADE match' leg: GPU vs oracle on that very sample with the same Philox seed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate
    from trajsde_amd.metrics import ADE_T, FDE_T
    skw = dict(spec["synth"], S=CPU_SAMPLE_SCENES)
    batch = synth(**skw)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    seed = 1234
    # the oracle is many small torch ops: more threads is not monotonically faster, so a few thread counts are timed
    # (1 warm-up + 2 timed forwards each) and the best one is the reported baseline
    all_threads = torch.get_num_threads()
    table = {}
    out = None
    for nt in sorted({1, 8, 32, all_threads}):
        if nt > all_threads:
            continue
        torch.set_num_threads(nt)
        times = []
        for i in range(3):
            t0 = time.perf_counter()
            out = restate.forward(P, cfg, batch, restate.PhiloxNoise(seed))
            times.append(time.perf_counter() - t0)
        table[nt] = float(np.mean(times[1:]))
    torch.set_num_threads(all_threads)
    best_nt = min(table, key=table.get)
    med = table[best_nt]
    o_gpu, data_gpu = gpu_loc_fn(batch, seed)
    idx = batch["agent_index"]
    res = {}
    for name, src in (("gpu", (o_gpu["loc"].cpu(), o_gpu["reg_mask"].cpu(), data_gpu.y.cpu())), ("cpu", (out["loc"], out["reg_mask"], out["y"]))):
        loc, mask, y = src
        last = loc.shape[2] - 1
        ade, fde = ADE_T("nuScenes", [last, last]), FDE_T("nuScenes", [last, last])
        args = (loc[:, idx, :, :2], y[idx], mask[idx], batch["source"])
        ade.update(*args)
        fde.update(*args)
        res[name] = (float(ade.compute()), float(fde.compute()))
    match = {"minADE_gpu": res["gpu"][0], "minADE_cpu_oracle": res["cpu"][0], "minFDE_gpu": res["gpu"][1],
             "minFDE_cpu_oracle": res["cpu"][1], "max_abs_loc_diff": float((o_gpu["loc"].cpu() - out["loc"]).abs().max()),
             "tolerance": 1e-4}
    base = {"value": CPU_SAMPLE_SCENES / med, "unit": "scenes/s", "cores": best_nt, "kind": "port",
            "sample": f"{CPU_SAMPLE_SCENES} of the 64 scenes of {WORKLOAD} (same generator and seed), oracle/restate.py, "
                      f"1 warm-up + 2 timed forwards per thread count, best = {best_nt} threads at {med:.3f} s/forward, "
                      f"torch {torch.__version__} fp32, host has {os.cpu_count()} logical cores",
            "scenes_per_s_by_threads": {str(k): CPU_SAMPLE_SCENES / v for k, v in table.items()}}
    return base, match


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TRAJSDE_BENCH_STREAMS", "3")),
                    help="HIP streams the K steps are dealt over (each step is still one complete forward of one batch)")
    ap.add_argument("--kernel-table", action="store_true", help="extra untimed pass timing every kernel (stderr)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary training-step figure")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        os.dup2(2, 1)       # only rank 0 owns stdout (the JSON line); anything other ranks' libraries print goes to stderr
    if world > 1 or os.environ.get("TRAJSDE_BENCH_FORCE_DIST") == "1":      # the latter: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)
    lib = _lib.lib()

    spec = CONFIGS[WORKLOAD]
    cfg = build_cfg(spec)
    model = PredictionModelSDENet(**cfg, init_seed=0).eval().to(dev)        # random-init weights of the named architecture
    skw = dict(spec["synth"])
    skw["seed"] = skw["seed"] + 1000 * rank                                   # every rank owns different scenes
    batch_cpu = synth(**skw)
    scenes = skw["S"]
    # Steps are dealt round-robin over `--streams` HIP streams so that the host-side part of step i+1 (graph
    # preparation incl. its one stream sync, launches) and its latency-bound kernels overlap the tail of step i.
    # Every step is a complete forward of one batch; each stream owns its own copy of the inputs.
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    batches, y0s = [], []
    for st in streams:
        with torch.cuda.stream(st):
            b = batch_cpu.to(dev)
            batches.append(b)
            y0s.append(b.y.clone())
    batch = batches[0]

    def step(i, single_stream=False):
        k = 0 if single_stream else i % n_streams
        b = batches[k]
        with torch.cuda.stream(streams[k]):
            b.y = y0s[k]                                                      # forward rotates y in place (MODEL:83-84)
            return model(b, noise=NoiseSpec(seed=10_000 * (rank + 1) + i))

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        sync_all()
        lib.trajsde_profile_mode(1)                                           # events around the dominant kernel only
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(args.warmup + i)
        sync_all()
        elapsed = time.perf_counter() - t0
        lib.trajsde_profile_mode(0)
    prof = _lib.profile_report()
    e_aa = batch[ "_trajsde_graph"].graph.E_aa
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_launch, total_ms, _ = prof.get("k_edge_kv[aa]", (0, 0.0, True))
        avg_s = (total_ms / max(n_launch, 1)) * 1e-3
        achieved = (FLOP_PER_EDGE * e_aa / avg_s) * 1e-12 if avg_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_final_traffic.json")
        if os.path.isfile(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("k_edge_kv[aa]", {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "scenes/sec (K=6, 20 SDE steps, ~256 agents) at 1/2/4/8 MI355X; minADE match",
            "value": world * scenes * args.steps / elapsed, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic batch of 64 scenes x 128 agents, K=6, 20 SDE steps, "
                                   "inference-only forward (graph prep + encoder + global interactor + SDE decoder), "
                                   "synth(S=64,n=128,L=64,F=20,box=200,seed=2,mixed_source)",
                       "scenes_per_gpu": scenes, "agents_per_scene": skw["n"], "num_modes": spec["num_modes"],
                       "future_steps": spec["future_steps"], "aa_edges_per_step": int(e_aa), "parallelism": f"scene-shard x{world}",
                       "streams_per_gpu": n_streams},
            "roofline": {"kernel": f"k_edge_kv[aa] (agent-agent edge embedding + k,v + logits; {SPLIT_NAME} split-precision MFMA 16x16x32, fp32-accurate)",
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_EQUIV_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_EQUIV_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": avg_s * 1e3, "launches": n_launch, "flop_per_edge": FLOP_PER_EDGE,
                         "peak_note": "algorithmic fp32 FLOP/s; the kernel is bound by VALU issue and LDS fragment reads, not by the matrix cores "
                                      "(ISA per 16-edge tile: ~550 VALU + 120 MFMA + ~76 LDS instructions with fp16x3; ~1060 + 240 + ~190 "
                                      "with bf16x6, where SQ counters showed the issue port saturated and the MFMA pipe ~45 % busy); "
                                      f"peak = 2500 TFLOP/s dense 16-bit MFMA / {SPLIT_PRODUCTS} products per fp32 product "
                                      "(the same kernel on exact fp32 MFMA, TRAJSDE_EDGE_FP32=1, ran at 110-113 TFLOP/s = 0.70-0.72 of the "
                                      "157.3 TFLOP/s fp32 matrix peak; bf16x6 ran at 142-188 TFLOP/s)",
                         "split_precision": SPLIT_NAME,
                         "issue_view": issue_view(e_aa, avg_s),
                         "mfma_16bit_tflops": achieved * SPLIT_PRODUCTS * (40960.0 / 41700.0)},
        }
        if not args.no_cpu_baseline and world == 1:
            def gpu_fn(b_cpu, seed):
                b = b_cpu.to(dev)
                with torch.no_grad():
                    o = model(b, noise=NoiseSpec(seed=seed))
                return o, b
            base, match = cpu_baseline(model, cfg, spec, gpu_fn)
            line["cpu_baseline"] = base
            line["minade_match"] = match
        # the same kernel measured alone (one stream, nothing overlapping it): the figure to hold against the rocprof
        # kernel-trace summary of a --streams 1 run; with several streams the events above also see co-running kernels
        lib.trajsde_profile_mode(1)
        with torch.no_grad():
            for i in range(5):
                step(800 + i, single_stream=True)
        torch.cuda.synchronize()
        lib.trajsde_profile_mode(0)
        iso_n, iso_ms, _ = _lib.profile_report().get("k_edge_kv[aa]", (0, 0.0, True))
        if iso_n:
            iso = FLOP_PER_EDGE * e_aa / (iso_ms / iso_n * 1e-3) * 1e-12
            line["roofline_isolated"] = {"kernel": "k_edge_kv[aa]", "streams": 1, "achieved": iso, "peak": PEAK_FP32_EQUIV_TFLOPS,
                                         "unit": "TFLOP/s", "frac": iso / PEAK_FP32_EQUIV_TFLOPS, "avg_launch_ms": iso_ms / iso_n,
                                         "launches": iso_n, "issue_view": issue_view(e_aa, iso_ms / iso_n * 1e-3)}
        # the step-granular decoder SDE step (state round-trips HBM every Euler step: SURVEY 8(d)'s 512 B / path-step
        # variant) in both views: algorithmic HBM GB/s -- the figure the north star names -- and the FLOP/s that binds it
        try:
            import ctypes as C
            from trajsde_amd.schedule import decoder_schedule
            rows = spec["num_modes"] * int(batch["x"].shape[0])
            tab = np.ascontiguousarray(decoder_schedule(spec["future_steps"], spec["max_fut_t"]).step_table())
            dblob = model.decoder._rt.blob()
            ya, yb = torch.randn(rows, 64, device=dev), torch.empty(rows, 64, device=dev)
            nz = _lib.Noise(C.c_uint64(7), None, None)
            cur = torch.cuda.current_stream().cuda_stream

            def sde_steps(n):
                for k in range(n):
                    e = tab[k % tab.shape[0]].ctypes.data_as(C.POINTER(C.c_float))
                    src, dst = (ya, yb) if k % 2 == 0 else (yb, ya)
                    _lib.check(lib.trajsde_sde_step(rows, dblob.data_ptr(), src.data_ptr(), dst.data_ptr(), e, k, C.byref(nz), cur))
            sde_steps(20)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sde_steps(100)
            e1.record()
            torch.cuda.synchronize()
            sms = e0.elapsed_time(e1) / 100
            gbs, tfl = rows * 512 / (sms * 1e-3) / 1e9, rows * 41.8e3 / (sms * 1e-3) / 1e12
            fp32_path = os.environ.get("TRAJSDE_DECODE_FP32", "0") not in ("", "0")
            fpeak = 157.3 if fp32_path else PEAK_FP32_EQUIV_TFLOPS
            line["roofline_sde_step"] = {"kernel": "k_sde_step (one Euler-Maruyama step per launch, state in HBM; "
                                                   + ("exact fp32 MFMA)" if fp32_path else f"{SPLIT_NAME} split-precision MFMA, fp32-accurate)"),
                                         "rows": rows, "avg_launch_ms": sms, "bound": "mfma",
                                         "hbm_view": {"achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                                      "bytes_per_path_step": 512},
                                         "flop_view": {"achieved": tfl, "peak": fpeak, "unit": "TFLOP/s", "frac": tfl / fpeak,
                                                       "flop_per_path_step": 41.8e3},
                                         "note": "82 FLOP/B against a ridge of ~20 FLOP/B (fp32 matrix peak) .. ~52-104 (split precision): compute-bound at "
                                                 "fp32 accuracy, so the HBM fraction is low by construction; the fused decoder never writes "
                                                 "the state at all"}
        except Exception as e:
            line["roofline_sde_step"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_train_step:
            # secondary figure (not `value`): the training step of the same workload -- forward + L2/DiffBCE + the three
            # stage backward calls + AdamW, reported next to the inference metric
            try:
                from trajsde_amd.driver import FlatGrads
                model.train()
                (opt,), _ = model.configure_optimizers()
                flat = FlatGrads(model.params_with_gradient())
                tb = batches[0]

                def tstep(i):
                    flat.zero()
                    tb.y = y0s[0]                                               # forward rotates y in place (MODEL:83-84)
                    model.training_step(tb, i, noise=NoiseSpec(seed=5000 + i)).backward()
                    opt.step()
                tstep(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(3):
                    tstep(1 + i)
                torch.cuda.synchronize()
                tms = (time.perf_counter() - t0) / 3 * 1e3
                line["train_step"] = {"ms_per_step": tms, "scenes_per_s": spec["synth"]["S"] / tms * 1e3, "steps": 3,
                                      "what": "training_step + backward (HIP backward kernels of all three stages) + AdamW on one "
                                              "batch of the same workload, fp32 gradients, 1 stream; weights are updated, so this "
                                              "runs after every inference measurement",
                                      "loss_L2": float(model.last_losses["L2"]), "loss_DiffBCE": float(model.last_losses["DiffBCE"])}
                model.eval()
            except Exception as e:                                              # never let the secondary figure cost the main line
                line["train_step"] = {"error": repr(e)[:300]}
        if args.kernel_table:
            lib.trajsde_profile_mode(2)
            with torch.no_grad():
                for i in range(3):
                    step(900 + i, single_stream=True)
            torch.cuda.synchronize()
            lib.trajsde_profile_mode(0)
            tab = _lib.profile_report()
            tot = sum(v[1] for v in tab.values())
            print(f"# per-kernel device time over 3 forwards (HIP events), total {tot / 3:.3f} ms/forward", file=sys.stderr)
            for tag, (n, ms, dom) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                print(f"#   {tag:24s} launches/fwd {n / 3:6.1f}  ms/fwd {ms / 3:8.3f}  {100 * ms / tot:5.1f}%", file=sys.stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:                                                                  # RCCL writes its version banner through C stdio,
            import ctypes                                                     # buffered until exit when stdout is a pipe: push it
            ctypes.CDLL(None).fflush(None)                                    # out now so that the JSON line really is the last one
        except Exception:
            pass
        print(json.dumps(line), flush=True)                                   # the one JSON line, last on stdout


if __name__ == "__main__":
    main()
