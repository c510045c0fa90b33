"""The encoder backward called N times on copies of one tape (BASELINE config 2): tape, scratch and gradients compared word for word with
the first call's.  Prints, per differing buffer, how many words differ and where (float index, row = index / 64, column = index % 64).

    python tools/encoder_bwd_repeat.py [--calls 12] [--config config2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def where(a, b, name):
    fa, fb = a.view(torch.float32) if a.dtype == torch.uint8 else a.reshape(-1), b.view(torch.float32) if b.dtype == torch.uint8 else b.reshape(-1)
    ia, ib = fa.view(torch.int32), fb.view(torch.int32)
    idx = (ia != ib).nonzero().reshape(-1)
    if idx.numel() == 0:
        return 0
    rel = ((fa[idx] - fb[idx]).abs() / fa[idx].abs().clamp_min(1e-30))
    head = [(int(i), int(i) // 64, int(i) % 64, float(fa[i]), float(fb[i])) for i in idx[:24].cpu()]
    rows = sorted({h[1] for h in head})
    print(f"   {name}: {idx.numel()} words differ, rel max {float(rel.max()):.2e}; first index {head[0][0]} of {fa.numel()}; rows {rows[:8]} cols {[h[2] for h in head[:16]]}")
    for h in head[:6]:
        print(f"        [{h[0]}] row {h[1]} col {h[2]}: {h[3]!r} / {h[4]!r}")
    return idx.numel()


def main():
    import helpers as H
    from trajsde_amd import runtime
    from trajsde_amd.synth import CONFIGS, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=12)
    ap.add_argument("--config", default="config2")
    a = ap.parse_args()
    spec = CONFIGS[a.config]
    dev = torch.device("cuda:0")
    K, T = spec["num_modes"], spec["future_steps"]
    model, cfg = H.build_model(K, T, spec["max_fut_t"], init_seed=0)
    model = model.to(dev).train()
    batch = synth(**spec["synth"]).to(dev)
    noise = runtime.NoiseSpec(seed=100, dropout_seed=101)
    with torch.no_grad():
        rot, y_rot = runtime.rotate_inputs(batch)
        batch.y, batch["rotate_mat"] = y_rot, rot
        outs, (ws, nbytes) = model.encoder._rt.encoder_forward_train(batch, noise)
        local = outs[0]
        g = torch.Generator().manual_seed(1)
        d_local = (torch.randn(local.shape[0], 64, generator=g) * 1e-3).to(dev)
        ref = None
        bad_calls = 0
        for call in range(a.calls):
            w2 = ws.clone()
            r = model.encoder._rt.encoder_backward(batch, d_local, noise, diff_weight=0.5, tape=(w2, nbytes), keep_scratch=True)
            torch.cuda.synchronize()
            cur = (w2, r["_scratch"], {k: v.clone() for k, v in r["grads"].items()})
            if ref is None:
                ref = cur
                print(f"tape {w2.numel() / 2 ** 20:.0f} MiB, scratch {cur[1].numel() / 2 ** 20:.0f} MiB")
                continue
            n = where(ref[0], cur[0], "tape") + where(ref[1], cur[1], "scratch")
            for k in cur[2]:
                n += where(ref[2][k], cur[2][k], "grad " + k)
            print(f"call {call}: {'identical' if n == 0 else str(n) + ' words differ'}")
            bad_calls += n != 0
            del cur
        print(f"{bad_calls} of {a.calls - 1} calls differ from the first")


if __name__ == "__main__":
    main()
