"""bench.py -- scenes/sec of PredictionModelSDENet.forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
        N = 1: one plain process.  N > 1 with WORLD_SIZE unset: this process touches no GPU and starts N rank
        processes itself (children, one per GPU, RCCL over xGMI), like `train.py --gpus N` lets Lightning do
        (train.py:35,54); rank 0's JSON line is the output.
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      # a launcher owns the ranks

A step is one forward pass (graph preparation + encoder + global interactor + SDE decoder, inference, fresh Philox
seed) over one synthetic batch of the workload BASELINE.json's metric is quoted on -- 256-agent scenes, K=6, 20 SDE
steps -- batched 32 scenes per GPU (synth.CONFIGS["metric256"]), inputs resident in HBM before the timed region.
Scenes shard over ranks with no data-path collective: every rank runs a same-sized batch of its own
scenes (weak scaling); the collectives are the timing barrier, the max-over-ranks of the elapsed time and one
all-reduce that proves RCCL saw N ranks.

The K timed steps are bracketed by barrier + synchronize on both sides; that window is repeated `--windows` times and
the MEDIAN window is reported (one host hiccup must not move the headline), every window's time is in the line.

Rank 0 prints ONE JSON line: the throughput, the roofline of the dominant kernel (HIP events recorded inside the
library on the launch stream during the timed region), the same workload on one stream, BASELINE configs[1]
(64 x 128) as a secondary figure, the step-granular SDE step and the CPU baseline (the oracle -- oracle/restate.py,
the bit-exact restatement of the reference -- timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = "metric256"               # BASELINE.json metric: 256-agent scenes, K=6, 20 SDE steps; 32 scenes per GPU
SECONDARY = "config2"                # BASELINE configs[1]: 64 scenes x 128 agents
FLOP_PER_EDGE = 41.7e3               # neighbour embed 25.1k + k,v 16.4k + dot 0.256k per (t, edge)
# The edge kernel evaluates its fp32 GEMMs as split-precision products on the 16-bit matrix cores (csrc/tile.hpp):
# three v_mfma_f32_16x16x32_f16 per fp32 product (fp16x3, the default build) or six ..._bf16 (bf16x6), fp32-accurate.
# Its roofline is therefore the dense fp16/bf16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s) divided by the number
# of products, expressed in algorithmic (fp32) FLOP/s.
MFMA_16BIT_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
CPU_SAMPLE_SCENES = 1
DTYPE = {3: "f32 (fp16x3 split: 22-bit operands on v_mfma_f32_16x16x32_f16, f32 accumulate)",
         6: "f32 (bf16x6 split: 24-bit operands on v_mfma_f32_16x16x32_bf16, f32 accumulate)"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes as CHILDREN of this process, which never
    initialises the GPU itself (no re-exec of a process that has).  Rank 0 inherits stdout, so its JSON line is ours."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


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
    """Oracle timed on the host cores (BASELINE.md section 3 protocol): CPU_SAMPLE_SCENES scene(s) of the workload's
    generator, per thread count 1 warm-up + the median of 3 timed forwards; also the 'minADE match' leg: GPU vs oracle
    on that very sample with the same Philox seed."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate
    from trajsde_amd.metrics import ADE_T, FDE_T
    from trajsde_amd.synth import synth
    skw = dict(spec["synth"], S=CPU_SAMPLE_SCENES)
    batch = synth(**skw)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    seed = 1234
    # the oracle is many small torch ops: more threads is not monotonically faster, so a few thread counts are timed and the
    # best one is the reported baseline
    all_threads = torch.get_num_threads()
    table, out = {}, None
    # (torch's own default -- all physical cores, 128 on the pool's hosts -- is NOT timed: it measured 3-4x slower than 16
    #  threads on this many-small-ops workload and cost ~30 s of every bench run for a number nobody uses)
    for nt in sorted({1, 8, 16, 32}):
        if nt > max(all_threads, 1):
            continue
        torch.set_num_threads(nt)
        times = []
        for i in range(4):                                  # one warm-up + three timed forwards per thread count (~8 s in all)
            t0 = time.perf_counter()
            out = restate.forward(P, cfg, batch, restate.PhiloxNoise(seed))
            times.append(time.perf_counter() - t0)
        table[nt] = float(np.median(times[1:]))
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
        a = (loc[:, idx, :, :2], y[idx], mask[idx], batch["source"])
        ade.update(*a)
        fde.update(*a)
        res[name] = (float(ade.compute()), float(fde.compute()))
    match = {"minADE_gpu": res["gpu"][0], "minADE_cpu_oracle": res["cpu"][0], "minFDE_gpu": res["gpu"][1],
             "minFDE_cpu_oracle": res["cpu"][1], "max_abs_loc_diff": float((o_gpu["loc"].cpu() - out["loc"]).abs().max()),
             "tolerance": 1e-4}
    base = {"value": CPU_SAMPLE_SCENES / med, "unit": "scenes/s", "cores": best_nt, "kind": "port",
            "sample": f"{CPU_SAMPLE_SCENES} scene x {skw['n']} agents of {WORKLOAD} (same generator and seed, K=6, 20 steps), "
                      f"oracle/restate.py (bit-exact restatement of the reference), 1 warm-up + median of 3 timed forwards per "
                      f"thread count, best = {best_nt} threads at {med:.3f} s/forward, torch {torch.__version__} fp32, "
                      f"host: {cpu_model()}, {os.cpu_count()} logical cores",
            "scenes_per_s_by_threads": {str(k): CPU_SAMPLE_SCENES / v for k, v in table.items()}}
    return base, match


def strict24_child(args) -> int:
    """The headline workload on the strict-precision library (variants/libtrajsde_strict24.so: bf16x6 split, 24-bit operands, same
    C-ABI), in a process of its own because a process holds ONE build of the library.  Prints one JSON object."""
    import numpy as np
    import torch
    from trajsde_amd import _lib
    from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth
    lib = _lib.lib()
    assert lib.trajsde_split_products() == 6, "the strict-precision child must run the bf16x6 library"
    dev = torch.device("cuda", 0)
    spec = CONFIGS[args.workload]
    model = PredictionModelSDENet(**build_cfg(spec), init_seed=0).eval().to(dev)
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    cpu = synth(**spec["synth"])
    batches, y0s = [], []
    for st in streams:
        with torch.cuda.stream(st):
            b = cpu.to(dev)
            batches.append(b)
            y0s.append(b.y.clone())
    torch.cuda.synchronize()

    def step(i):
        k = i % n_streams
        with torch.cuda.stream(streams[k]), torch.no_grad():
            batches[k].y = y0s[k]
            model(batches[k], noise=NoiseSpec(seed=10_000 + i))
    for i in range(2 * n_streams):
        step(i)
    wins = []
    for w in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(100 + w * args.steps + i)
        torch.cuda.synchronize()
        wins.append(time.perf_counter() - t0)
    el = float(np.median(wins))
    _lib.check_range()
    sample = synth(**dict(spec["synth"], S=CPU_SAMPLE_SCENES)).to(dev)
    with torch.no_grad():
        loc = model(sample, noise=NoiseSpec(seed=1234))["loc"].cpu().numpy()
    ref = np.load(args.strict24_child)
    print(json.dumps({"value": spec["synth"]["S"] * args.steps / el, "ms_per_step": 1e3 * el / args.steps, "unit": "scenes/s",
                      "max_abs_loc_diff": float(np.abs(loc - ref).max()), "windows_ms": [1e3 * w for w in wins],
                      "steps": args.steps, "streams_per_gpu": n_streams, "split_products": 6}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed K-step windows; the median one is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TRAJSDE_BENCH_STREAMS", "3")),
                    help="HIP streams the K steps are dealt over (each step is still one complete forward of one batch)")
    ap.add_argument("--workload", default=WORKLOAD, help="synth.CONFIGS key of the headline workload")
    ap.add_argument("--kernel-table", action="store_true", help="extra untimed pass timing every kernel (stderr)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary training-step figure")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1-stream / 64x128 / SDE-step side figures")
    ap.add_argument("--strict24-child", default=None, metavar="NPY",
                    help="(internal) run as the child that times the 24-bit-operand library (TRAJSDE_LIB set by the parent) and "
                         "compares its trajectories with the parent's, saved in NPY")
    args = ap.parse_args()
    if args.strict24_child:
        sys.exit(strict24_child(args))

    if (args.gpus > 1 or os.environ.get("TRAJSDE_BENCH_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                 # nothing above this line has touched the GPU or loaded the HIP library

    import numpy as np
    import torch
    from trajsde_amd import _lib
    from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet
    from trajsde_amd import runtime as runtime_mod
    from trajsde_amd.runtime import NoiseSpec
    from trajsde_amd.synth import CONFIGS, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:
        os.dup2(2, 1)       # only rank 0 owns stdout (the JSON line); anything other ranks' libraries print goes to stderr
    # N ranks share one host: each is confined to its share of the cores and torch's thread pool capped to it, before any GPU call
    from trajsde_amd.shard import pin_rank_to_cores
    host_share = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    rccl_ranks = None
    if world > 1 or os.environ.get("TRAJSDE_BENCH_FORCE_DIST") == "1":      # the latter: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        one = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(one)                                                  # a real collective over RCCL: every rank adds 1
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)
    lib = _lib.lib()
    split_products = lib.trajsde_split_products()
    split_name = {3: "fp16x3", 6: "bf16x6"}[split_products]
    peak_fp32_equiv = MFMA_16BIT_PEAK_TFLOPS / split_products

    spec = CONFIGS[args.workload]
    cfg = build_cfg(spec)
    model = PredictionModelSDENet(**cfg, init_seed=0).eval().to(dev)        # random-init weights of the named architecture
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

    class Workload:
        """one synthetic batch of `name` resident in HBM, one copy per stream (each stream owns its inputs)"""

        def __init__(self, name):
            self.spec = CONFIGS[name]
            skw = dict(self.spec["synth"])
            skw["seed"] = skw["seed"] + 1000 * rank                           # every rank owns different scenes
            self.skw, self.scenes = skw, skw["S"]
            cpu = synth(**skw)
            self.batches, self.y0s = [], []
            for st in streams:
                with torch.cuda.stream(st):
                    b = cpu.to(dev)
                    self.batches.append(b)
                    self.y0s.append(b.y.clone())
            torch.cuda.synchronize()

        # Steps are dealt round-robin over the streams so that the latency-bound kernels of step i+1 overlap the tail of
        # step i.  The forward is sync-free (runtime.sync_free: list lengths stay on the device), so the host only
        # enqueues.  Every step is a complete forward of one batch.
        def step(self, i, single_stream=False):
            k = 0 if single_stream else i % n_streams
            b = self.batches[k]
            with torch.cuda.stream(streams[k]):
                b.y = self.y0s[k]                                             # forward rotates y in place (MODEL:83-84)
                return model(b, noise=NoiseSpec(seed=10_000 * (rank + 1) + i))

        def e_aa(self):
            return self.batches[0]["_trajsde_graph"].true_counts()["E_aa"]       # (reads the device-side count: outside timed regions)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_window(wl, steps, first, single_stream=False):
        """EXACTLY `steps` forwards between barrier + synchronize on both sides; max over ranks"""
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(first + i, single_stream)
        sync_all()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    wl = Workload(args.workload)
    with torch.no_grad():
        # W untimed steps -- and never fewer than two per stream: the first step of a stream sizes its buffers, the second is the
        # first to run while the previous one's buffers are still alive; leaving that to the timed region put a 100-200 ms
        # allocator stall into its first window on some boxes
        warm_steps = max(args.warmup, 2 * n_streams)
        for i in range(warm_steps):
            wl.step(i)
        sync_all()
        lib.trajsde_profile_mode(1)                                           # events around the dominant kernel only
        windows = [timed_window(wl, args.steps, args.warmup + w * args.steps) for w in range(max(1, args.windows))]
        lib.trajsde_profile_mode(0)
    prof = _lib.profile_report()
    e_aa = wl.e_aa()
    elapsed = float(np.median(windows))
    _lib.check_range()                                                        # no saturated fp16x3 operand in the timed forwards
    # The same workload on ONE stream, on every rank: the dominant kernel's HIP events then bracket that kernel alone -- the
    # figure a `--streams 1` rocprofv3 kernel trace reproduces (profiles/) and the one `roofline` reports.  With several
    # streams the events also span whatever co-runs on the chip (`roofline_corun`).
    with torch.no_grad():
        for i in range(2):
            wl.step(700 + i, single_stream=True)
        lib.trajsde_profile_mode(1)
        w1 = [timed_window(wl, args.steps, 800 + w * args.steps, single_stream=True) for w in range(3)]
        lib.trajsde_profile_mode(0)
    iso_n, iso_ms, _ = _lib.profile_report().get("k_edge_kv[aa]", (0, 0.0, True))
    el1 = float(np.median(w1))

    def edge_roofline(n_launch, total_ms, e_aa_, streams_):
        avg_s = (total_ms / max(n_launch, 1)) * 1e-3
        ach = (FLOP_PER_EDGE * e_aa_ / avg_s) * 1e-12 if avg_s > 0 else 0.0
        return {"kernel": f"k_edge_attn2, profile tag k_edge_kv[aa] (fused agent-agent edge attention: embedding + k,v + online segment softmax + "
                          f"aggregation; {split_name} split-precision MFMA 16x16x32, fp32-accurate)",
                "bound": "mfma", "achieved": ach, "peak": peak_fp32_equiv, "unit": "TFLOP/s", "frac": ach / peak_fp32_equiv,
                "avg_launch_ms": avg_s * 1e3, "launches": n_launch, "flop_per_edge": FLOP_PER_EDGE, "edges_per_launch": int(e_aa_),
                "streams": streams_, "mfma_16bit_tflops": ach * split_products * (40960.0 / 41700.0)}

    if rank == 0:
        n_launch, total_ms, _ = prof.get("k_edge_kv[aa]", (0, 0.0, True))
        roof = edge_roofline(iso_n, iso_ms, e_aa, 1)                          # the kernel alone (one stream): what profiles/ reproduces
        roof_corun = edge_roofline(n_launch, total_ms, e_aa, n_streams)       # inside the timed multi-stream region
        roof["traffic"] = None                                                # HBM bytes per launch are a PMC quantity: not measurable in this run
        for tname in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.isfile(tpath):
                with open(tpath) as f:
                    roof["traffic_profiled"] = dict(json.load(f).get("k_edge_kv[aa]", {}), source=f"profiles/{tname} (rocprofv3 --pmc "
                                                    "FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command; NOT measured in this run)")
                break
        roof["peak_note"] = (f"algorithmic fp32 FLOP/s; peak = 2500 TFLOP/s dense 16-bit MFMA / {split_products} products per fp32 product; "
                             "HIP events of the one-stream pass of this run (the kernel alone on the chip); `roofline_corun` is the same "
                             "kernel inside the timed multi-stream region, where the events also span co-running kernels")
        line = {
            "metric": "scenes/sec (K=6, 20 SDE steps, ~256 agents) at 1/2/4/8 MI355X; minADE match",
            "value": world * wl.scenes * args.steps / elapsed, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[split_products], "data": "synthetic",
            "config": {"workload": f"BASELINE metric workload: {wl.scenes} scenes x {wl.skw['n']} agents per GPU, K={spec['num_modes']}, "
                                   f"{spec['future_steps']} SDE steps, inference-only forward (graph prep + encoder + global interactor + "
                                   f"SDE decoder), synth({', '.join(f'{k}={v}' for k, v in spec['synth'].items())})",
                       "scenes_per_gpu": wl.scenes, "agents_per_scene": wl.skw["n"], "num_modes": spec["num_modes"],
                       "future_steps": spec["future_steps"], "aa_edges_per_step": int(e_aa), "parallelism": f"scene-shard x{world}",
                       "streams_per_gpu": n_streams, "rccl_ranks_seen": rccl_ranks, "sync_free_forward": bool(runtime_mod.sync_free()),
                       "host_cores_rank0": host_share},
            "timing": {"windows_ms": [1e3 * w for w in windows], "reported": "median window", "steps_per_window": args.steps,
                       "untimed_warmup_steps_run": warm_steps},
            "roofline": roof,
            "roofline_corun": roof_corun,
            "streams1": {"value": world * wl.scenes * args.steps / el1, "ms_per_step": 1e3 * el1 / args.steps,
                         "windows_ms": [1e3 * w for w in w1], "what": "the same steps on one HIP stream per GPU"},
        }
        if world > 1:                                                         # world-1-only legs: say so instead of leaving the keys out
            for k in ("cpu_baseline", "minade_match", "graph_replay", "config2_64x128", "config3_argo_t30", "roofline_sde_step", "strict24", "train_step"):
                line[k] = "n/a (N>1): measured by the N=1 run"                   # (config4_train / train_scaling ARE measured under N>1)
        if not args.no_cpu_baseline and world == 1:
            def gpu_fn(b_cpu, seed):
                b = b_cpu.to(dev)
                with torch.no_grad():
                    o = model(b, noise=NoiseSpec(seed=seed))
                return o, b
            base, match = cpu_baseline(model, cfg, spec, gpu_fn)
            line["cpu_baseline"] = base
            line["minade_match"] = match
    if not args.no_secondary and world == 1:
        with torch.no_grad():
            # the same steps as HIP-graph replays (runtime.GraphedForward: the whole forward incl. the graph stage captured per
            # stream, the Philox key read from device memory so every replay draws fresh noise): the GPU side is unchanged --
            # the forward is GPU-bound -- what it removes is the host's ~0.4 ms of launches per forward
            try:
                gfs = []
                for k, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        wl.batches[k].y = wl.y0s[k]
                        gfs.append(runtime_mod.GraphedForward(model, wl.batches[k]))
                torch.cuda.synchronize()

                def graph_window(first):
                    sync_all()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        k = (first + i) % n_streams
                        with torch.cuda.stream(streams[k]):
                            gfs[k](seed=10_000 * (rank + 1) + first + i)
                    t_enq = time.perf_counter() - t0
                    sync_all()
                    return time.perf_counter() - t0, t_enq
                graph_window(0)
                gw = [graph_window(900 + w * args.steps) for w in range(3)]
                elg = float(np.median([g[0] for g in gw]))
                line["graph_replay"] = {"value": wl.scenes * args.steps / elg, "ms_per_step": 1e3 * elg / args.steps,
                                        "host_enqueue_ms_per_step": 1e3 * float(np.median([g[1] for g in gw])) / args.steps,
                                        "streams_per_gpu": n_streams, "windows_ms": [1e3 * g[0] for g in gw],
                                        "what": "hipGraph replay of the full forward (graph stage + encoder + interactor + decoder) per stream"}
                del gfs
            except Exception as e:                                          # a secondary figure never costs the main line
                line["graph_replay"] = {"error": repr(e)[:300]}
            # BASELINE configs[1] (64 scenes x 128 agents), the round-1 headline, as a secondary figure
            if SECONDARY != args.workload:
                w2l = Workload(SECONDARY)
                for i in range(2):
                    w2l.step(i)
                w2 = [timed_window(w2l, args.steps, 100 + w * args.steps) for w in range(3)]
                el2 = float(np.median(w2))
                line["config2_64x128"] = {"value": w2l.scenes * args.steps / el2, "ms_per_step": 1e3 * el2 / args.steps, "unit": "scenes/s",
                                          "workload": "BASELINE configs[1]: 64 scenes x 128 agents, K=6, 20 SDE steps",
                                          "aa_edges_per_step": int(w2l.e_aa()), "streams_per_gpu": n_streams, "windows_ms": [1e3 * w for w in w2]}
                del w2l
            # BASELINE configs[2] (Argoverse-shaped: 32 scenes x 48 agents, 150 lanes, K=6, T=30 -> 31 Euler steps incl. the solver's
            # micro-step): forward throughput, and minADE / minFDE of one of its scenes against the oracle on the same Philox seed
            try:
                c3 = CONFIGS["config3"]
                m3 = PredictionModelSDENet(**build_cfg(c3), init_seed=0).eval().to(dev)
                skw3 = dict(c3["synth"])
                cpu3 = synth(**skw3)
                b3s, y3s = [], []
                for st in streams:
                    with torch.cuda.stream(st):
                        b = cpu3.to(dev)
                        b3s.append(b)
                        y3s.append(b.y.clone())
                torch.cuda.synchronize()

                def step3(i):
                    k = i % n_streams
                    with torch.cuda.stream(streams[k]):
                        b3s[k].y = y3s[k]
                        m3(b3s[k], noise=NoiseSpec(seed=30_000 + i))
                for i in range(2 * n_streams):
                    step3(i)
                w3 = []
                for w in range(3):
                    sync_all()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        step3(100 + w * args.steps + i)
                    sync_all()
                    w3.append(time.perf_counter() - t0)
                el3 = float(np.median(w3))
                entry = {"value": skw3["S"] * args.steps / el3, "ms_per_step": 1e3 * el3 / args.steps, "unit": "scenes/s",
                         "workload": "BASELINE configs[2] shape: 32 scenes x 48 agents, 150 lanes, K=6, T=30 (31 Euler steps), source = Argoverse; "
                                     "synthetic stand-in for the Argoverse val split (no preprocessed data in the image)",
                         "streams_per_gpu": n_streams, "windows_ms": [1e3 * w for w in w3]}
                if not args.no_cpu_baseline:
                    sys.path.insert(0, os.path.join(ROOT, "oracle"))
                    import restate
                    from trajsde_amd.metrics import ADE_T, FDE_T
                    one = synth(**dict(skw3, S=1))
                    P3 = {k: v.detach().cpu().clone() for k, v in m3.state_dict().items()}
                    want = restate.forward(P3, build_cfg(c3), one, restate.PhiloxNoise(4321))
                    bg = one.to(dev)
                    got = m3(bg, noise=NoiseSpec(seed=4321))
                    idx = one["agent_index"]
                    vals = {}
                    for name, (loc, mask, y) in (("gpu", (got["loc"].cpu(), got["reg_mask"].cpu(), bg.y.cpu())),
                                                 ("cpu", (want["loc"], want["reg_mask"], want["y"]))):
                        last = loc.shape[2] - 1
                        ade, fde = ADE_T("Argoverse", [last, last]), FDE_T("Argoverse", [last, last])
                        a = (loc[:, idx, :, :2], y[idx], mask[idx], one["source"])
                        ade.update(*a)
                        fde.update(*a)
                        vals[name] = (float(ade.compute()), float(fde.compute()))
                    entry["minade_match"] = {"minADE_gpu": vals["gpu"][0], "minADE_cpu_oracle": vals["cpu"][0], "minFDE_gpu": vals["gpu"][1],
                                             "minFDE_cpu_oracle": vals["cpu"][1],
                                             "max_abs_loc_diff": float((got["loc"].cpu() - want["loc"]).abs().max()), "tolerance": 1e-4,
                                             "sample": "one 48-agent scene of the same generator, Philox seed 4321"}
                line["config3_argo_t30"] = entry
                del m3, b3s, y3s
            except Exception as e:
                line["config3_argo_t30"] = {"error": repr(e)[:300]}
        # The strict-precision twin: the same workload on the 24-bit-operand build of the library (bf16x6 split,
        # variants/libtrajsde_strict24.so), in a CHILD process -- one process holds one build -- and its trajectories on the
        # CPU-baseline sample (same generator, Philox seed 1234) against this process's fp16x3 ones
        if split_products == 3 and rank == 0:
            try:
                import tempfile
                from trajsde_amd import build as build_mod
                if not os.path.isfile(build_mod.STRICT_LIB):
                    raise FileNotFoundError(build_mod.STRICT_LIB + " (python -m trajsde_amd.build --strict, or __graft_entry__.build())")
                sample = synth(**dict(spec["synth"], S=CPU_SAMPLE_SCENES)).to(dev)
                with torch.no_grad():
                    loc0 = model(sample, noise=NoiseSpec(seed=1234))["loc"].cpu().numpy()
                with tempfile.TemporaryDirectory() as td:
                    npy = os.path.join(td, "loc_fp16x3.npy")
                    np.save(npy, loc0)
                    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRAJSDE_BENCH_SPAWN",
                                                                            "TRAJSDE_BENCH_FORCE_DIST")}
                    env["TRAJSDE_LIB"] = build_mod.STRICT_LIB
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--strict24-child", npy, "--steps", str(min(args.steps, 10)),
                                        "--streams", str(n_streams), "--workload", args.workload], env=env, timeout=600,
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-300:])
                st = json.loads(r.stdout.strip().splitlines()[-1])
                st["dtype"] = DTYPE[6]
                st["ratio_to_value"] = st["value"] / line["value"]
                st["what"] = ("the headline workload on variants/libtrajsde_strict24.so (TRAJSDE_SPLIT=bf16x6: three bf16 pieces, six products, "
                              "24-bit operands), timed in a child process; max_abs_loc_diff: its trajectories against the fp16x3 build's on "
                              "the CPU-baseline sample, same Philox seed")
                line["strict24"] = st
            except Exception as e:
                line["strict24"] = {"error": repr(e)[:300]}
        # the step-granular decoder SDE step (state round-trips HBM every Euler step: SURVEY 8(d)'s 512 B / path-step
        # variant) in both views: algorithmic HBM GB/s -- the figure the north star names -- and the FLOP/s beside it
        try:
            import ctypes as C
            from trajsde_amd.schedule import decoder_schedule
            tab = np.ascontiguousarray(decoder_schedule(spec["future_steps"], spec["max_fut_t"]).step_table())
            dblob = model.decoder._rt.blob()
            nz = _lib.Noise(C.c_uint64(7), None, None)
            cur = torch.cuda.current_stream().cuda_stream
            views = {}
            for label, rows in (("workload", spec["num_modes"] * int(wl.batches[0]["x"].shape[0])), ("stress_786k", 786_432)):
                ya, yb = torch.randn(rows, 64, device=dev), torch.empty(rows, 64, device=dev)

                def sde_steps(n):
                    for k in range(n):
                        e = tab[k % tab.shape[0]].ctypes.data_as(C.POINTER(C.c_float))
                        src, dst = (ya, yb) if k % 2 == 0 else (yb, ya)
                        _lib.check(lib.trajsde_sde_step(rows, dblob.data_ptr(), src.data_ptr(), dst.data_ptr(), e, k, C.byref(nz), cur))
                # 500 warm-up launches (~75 ms): the chip's clock ramps over the first few hundred launches of a burst, and a figure taken
                # after 20 read 0.29-0.31 of the HBM peak where the warmed kernel does 0.33-0.36 (profiles/r05_sde_step_floor.md section 1)
                sde_steps(500)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sde_steps(400)
                e1.record()
                torch.cuda.synchronize()
                sms = e0.elapsed_time(e1) / 400
                gbs, tfl = rows * 512 / (sms * 1e-3) / 1e9, rows * 41.8e3 / (sms * 1e-3) / 1e12
                views[label] = {"rows": rows, "avg_launch_ms": sms, "launches": 400, "warmup_launches": 500,
                                "hbm_view": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_path_step": 512},
                                "flop_view": {"achieved": tfl, "peak": peak_fp32_equiv, "unit": "TFLOP/s", "frac": tfl / peak_fp32_equiv,
                                              "flop_per_path_step": 41.8e3}}
                del ya, yb
            line["roofline_sde_step"] = dict(views, kernel=f"k_sde_step (one Euler-Maruyama step per launch, state in HBM; {split_name} split-precision MFMA, "
                                             "fp32-accurate)", note="82 FLOP/B: the ridge of the split-precision peak is 833/8 = 104 FLOP/B, so the step sits on the HBM "
                                             "side of it; the fused decoder (the product path) never writes the state at all")
        except Exception as e:
            line["roofline_sde_step"] = {"error": repr(e)[:300]}
    # secondary figures (not `value`): the training step -- forward + L2/DiffBCE + the three stage backward calls + AdamW
    #  -- at BASELINE configs[1] (64 x 128, K=6, T=20) and at the shape of configs[3], the shipped
    # training recipe (128 scenes x 48 agents per GPU, K=10, T=60, mixed sources: CFG:9-22,106)
    def train_figure(name, what, collective=False, steps=8):
        """`collective`: the multi-rank loop of driver.train -- every rank its own scenes, the flat gradient buffer averaged by the
        ONE all-reduce of the design (FlatGrads: decoder + aggregator slice early on the collective stream under the
        encoder backward, the rest after it) -- timed with and without the collective, max over ranks"""
        from trajsde_amd.driver import FlatTraining
        from trajsde_amd import runtime
        from trajsde_amd.data import TemporalData
        twl = Workload(name)
        tspec = CONFIGS[name]
        tmodel = PredictionModelSDENet(**build_cfg(tspec), init_seed=0).to(dev).train()
        flat = FlatTraining(tmodel)                                       # the training loop's handle (driver.train)
        tb = twl.batches[0]
        tbase = {k: v for k, v in tb.as_dict().items() if not k.startswith("_")}
        tbase["y"] = twl.y0s[0]
        tside = runtime.side_stream(dev)
        early_ok = flat.grads.early_plan({id(p): n for n, p in tmodel.named_parameters()})
        flat.grads.force_collective = bool(collective) and world == 1     # one forced-dist rank still issues the RCCL launches

        def tfresh(i):                  # the loop of driver.train: the next step's copy of the batch, rotated and through the graph
            with torch.cuda.stream(tside):                                # stage (one host synchronisation) on the side stream
                b = TemporalData(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in tbase.items()})
                tmodel.prefetch_graph(b, NoiseSpec(seed=5000 + 100_000 * rank + i))
            return b
        def all_ranks_ok(ok, stage):
            """A rank that failed must not leave the others inside a collective it will never join: every rank reports, all of them
            leave together (the flag's own all-reduce is the only collective a failed rank still takes part in)."""
            if dist is None or world == 1:
                return ok
            flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and ok:
                raise RuntimeError(f"train_figure({name}): another rank failed during {stage}; this rank leaves the leg with it")
            return ok

        # set-up and ONE step without any collective first: allocation failures and layout refusals show up here, on their own rank
        err = None
        try:
            tnext = [tfresh(0)]
            flat.grads.early_enabled = False
            flat.zero()
            tmodel.training_step(tnext[0], 0, noise=NoiseSpec(seed=5000 + 100_000 * rank)).backward()
            flat.step()
            tnext[0] = tfresh(1)
            torch.cuda.synchronize()
        except Exception as e:        # noqa: BLE001 -- reported below, after the other ranks know
            err = e
        all_ranks_ok(err is None, "the collective-free first step")
        if err is not None:
            raise err

        def tstep(i, reduce_):
            flat.zero()
            tmodel.training_step(tnext[0], i, noise=NoiseSpec(seed=5000 + 100_000 * rank + i)).backward()
            if reduce_:
                flat.all_reduce_mean()
            flat.step()
            tnext[0] = tfresh(i + 1)

        def timed(first, reduce_):
            flat.grads.early_enabled = bool(reduce_ and early_ok)
            for i in range(4):                                             # optimizer state, allocator pools (a pipelined loop's block sizes
                                                                           # take more than two steps to settle: a late hipMalloc inside the
                                                                           # timed window was a 45 ms outlier in 2 of ~20 runs), communicator
                tstep(first + i, reduce_)
            sync_all()
            t0 = time.perf_counter()
            for i in range(steps):
                tstep(first + 4 + i, reduce_)
            sync_all()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el / steps * 1e3
        try:
            tms = timed(0, bool(collective))
        except Exception:
            # inside the collective loop there is no point at which the other ranks could be told: they sit in an all-reduce this rank
            # will never join.  Leave hard, so that the launcher tears the job down at once instead of after the RCCL watchdog's minutes.
            if collective and dist is not None and world > 1:
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                os._exit(3)
            raise
        S = tspec["synth"]["S"]
        fig = {"ms_per_step": tms, "scenes_per_s": world * S / tms * 1e3, "steps": steps, "workload": what,
               "what": "training_step + backward (HIP backward kernels of all three stages) + AdamW, fp32 gradients, a fresh "
                       "copy of the batch per step, the loop of driver.train (next batch's graph stage on a side stream)"
                       + ("; every rank its own scenes, gradients averaged over the ranks by RCCL" if collective else ""),
               "loss_L2": float(tmodel.last_losses["L2"]), "loss_DiffBCE": float(tmodel.last_losses["DiffBCE"]),
               "peak_mem_GB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
        scaling = None
        if collective:
            t_off = timed(100, False)                                     # the same steps with no collective at all
            scaling = {"ms_per_step": tms, "ms_per_step_without_collective": t_off, "allreduce_exposed_ms": tms - t_off,
                       "bytes": int(flat.grads.flat.numel()) * 4, "overlap": bool(early_ok),
                       "early_slice_bytes": (int(flat.grads.flat.numel()) - _encoder_floats(tmodel, flat)) * 4 if early_ok else 0,
                       "ranks": world, "scenes_per_s": world * S / tms * 1e3, "steps": steps,
                       "collective": "ONE all-reduce(sum) of the flat fp32 gradient buffer per step over RCCL, in two slices: decoder + "
                                     "aggregator block on the collective stream under the encoder backward, encoder block after it "
                                     "(driver.FlatGrads.early_reduce / all_reduce_mean; train.py:35,54, CFG:106)",
                       "note": "max over ranks, barrier + synchronize on both sides; with one rank (forced dist) the collectives are "
                               "one-rank RCCL launches: the stream choreography is exercised, the wire is not"}
        flat.grads.force_collective = False
        return fig, scaling

    def _encoder_floats(tmodel, flat):
        enc = {id(p) for n, p in tmodel.named_parameters() if n.startswith("encoder.")}
        return sum(p.numel() for p in flat.grads.params if id(p) in enc)

    if not args.no_train_step and dist is not None:
        # N > 1 (or one forced-dist rank): the shipped training recipe's shape per rank with the design's one collective in the step
        try:
            fig, scaling = train_figure("config4", "BASELINE configs[3] shape per GPU: 128 scenes x 48 agents, 150 lanes, K=10, T=60 "
                                                   "(61 Euler steps), mixed nuScenes / Argoverse sources, dropout 0.1", collective=True)
            if rank == 0:
                line["config4_train"] = fig
                line["train_scaling"] = scaling
        except Exception as e:
            if rank == 0:
                line["config4_train"] = {"error": repr(e)[:300]}
                line["train_scaling"] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and dist is None and not args.no_train_step:
        for key, name, what in (("train_step", SECONDARY, "BASELINE configs[1]: 64 scenes x 128 agents, K=6, 20 SDE steps"),
                                ("config4_train", "config4", "BASELINE configs[3] shape per GPU: 128 scenes x 48 agents, 150 lanes, K=10, T=60 "
                                                             "(61 Euler steps), mixed nuScenes / Argoverse sources, dropout 0.1")):
            try:
                line[key] = train_figure(name, what)[0]
            except Exception as e:                                          # never let a secondary figure cost the main line
                line[key] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
    if rank == 0 and args.kernel_table:
        lib.trajsde_profile_mode(2)
        with torch.no_grad():
            for i in range(3):
                wl.step(900 + i, single_stream=True)
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
