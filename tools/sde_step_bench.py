"""Step-granular decoder SDE step (k_sde_step): one Euler-Maruyama step per launch with the state round-tripping
HBM -- the 512 B / path-step variant of SURVEY.md 8(d).  Prints achieved algorithmic GB/s and TFLOP/s so the
north star's "HBM roofline in the SDE step" can be read next to the FLOP roofline that actually binds.

    python tools/sde_step_bench.py [rows]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from trajsde_amd import _lib  # noqa: E402
from trajsde_amd.schedule import decoder_schedule  # noqa: E402


def make_runner(rows):
    """run(i): 20 launches of trajsde_sde_step on `rows` rows (tools/phase_stamps.py)"""
    dev = torch.device("cuda:0")
    model, cfg = H.build_model(6, 20, 2.0, init_seed=0)
    model = model.to(dev)
    blob = model.decoder._rt.blob()
    L = _lib.lib()
    sched = decoder_schedule(20, 2.0)
    tab = np.ascontiguousarray(sched.step_table())
    y = [torch.randn(rows, 64, device=dev), torch.empty(rows, 64, device=dev)]
    noise = _lib.Noise(C.c_uint64(7), None, None)

    def run(i):
        st = torch.cuda.current_stream().cuda_stream
        for k in range(20):
            e = tab[k % sched.n_euler].ctypes.data_as(C.POINTER(C.c_float))
            _lib.check(L.trajsde_sde_step(rows, blob.data_ptr(), y[k & 1].data_ptr(), y[(k + 1) & 1].data_ptr(), e, k, C.byref(noise), st))
        run.keep = (model, blob, y)
    return run


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6 * 8192        # K*N of BASELINE config 2
    dev = torch.device("cuda:0")
    model, cfg = H.build_model(6, 20, 2.0, init_seed=0)
    model = model.to(dev)
    blob = model.decoder._rt.blob()
    L = _lib.lib()
    sched = decoder_schedule(20, 2.0)
    tab = np.ascontiguousarray(sched.step_table())
    y = [torch.randn(rows, 64, device=dev), torch.empty(rows, 64, device=dev)]
    noise = _lib.Noise(C.c_uint64(7), None, None)
    st = torch.cuda.current_stream().cuda_stream

    def run(n):
        for k in range(n):
            e = tab[k % sched.n_euler].ctypes.data_as(C.POINTER(C.c_float))
            _lib.check(L.trajsde_sde_step(rows, blob.data_ptr(), y[k & 1].data_ptr(), y[(k + 1) & 1].data_ptr(), e, k, C.byref(noise), st))

    run(int(os.environ.get("SDE_STEP_WARMUP", "500")))          # the clock ramps over the first ~100 launches: time a warmed chip
    torch.cuda.synchronize()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    beg.record()
    run(n)
    end.record()
    torch.cuda.synchronize()
    ms = beg.elapsed_time(end) / n
    gbs = rows * 512 / (ms * 1e-3) / 1e9
    tfl = rows * 41.8e3 / (ms * 1e-3) / 1e12
    print(json.dumps({"kernel": "k_sde_step", "rows": rows, "ms_per_step": ms, "algorithmic_bytes_per_path_step": 512,
                      "hbm_GBps": gbs, "hbm_frac_of_8TBps": gbs / 8000, "flop_per_path_step": 41.8e3, "TFLOPps": tfl,
                      "mfma_f32_frac": tfl / 157.3, "arithmetic_intensity_flop_per_byte": 41.8e3 / 512}))


if __name__ == "__main__":
    main()
