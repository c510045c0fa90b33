"""Stage-by-stage parity report on the GPU box (not a pytest file): prints max-abs differences of every
intermediate against the golden fixtures so that a mismatch can be localised in one run."""
import sys
import os
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H  # noqa: E402
from trajsde_amd import philox  # noqa: E402
from trajsde_amd.runtime import GraphContext, NoiseSpec, rotate_inputs  # noqa: E402
from trajsde_amd.schedule import decoder_schedule  # noqa: E402


def d(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def main():
    dev = torch.device("cuda:0")
    for name in H.GOLDEN:
        print(f"==== {name}")
        batch, meta, out, mid = H.load_fixture(name)
        model, cfg = H.build_model(meta)
        model = model.to(dev)
        seed = int(meta["noise_seed"])
        K, T = int(meta["num_modes"]), int(meta["future_steps"])
        N, A = batch.num_nodes, batch["agent_index"].numel()
        try:
            # decoder alone, fed with the reference's embeddings
            data = batch.to(dev)
            dec = model.decoder(data=data, local_embed=mid["local_embed"].to(dev), global_embed=mid["global_embed"].to(dev),
                                noise=NoiseSpec(seed=seed))
            print(f"  decoder(philox)   loc {d(dec['loc'], out['loc']):.3e}  pi {d(dec['pi'], out['pi']):.3e}")
            sched = decoder_schedule(T, float(meta["max_fut_t"]))
            z_dec = torch.from_numpy(np.stack([philox.normals(seed, philox.STREAM_DECODER, k, np.arange(K * N), 64)
                                               for k in range(sched.n_euler)])).to(dev)
            dec = model.decoder(data=data, local_embed=mid["local_embed"].to(dev), global_embed=mid["global_embed"].to(dev),
                                noise=NoiseSpec(seed=0, z_dec=z_dec))
            print(f"  decoder(injected) loc {d(dec['loc'], out['loc']):.3e}  pi {d(dec['pi'], out['pi']):.3e}")
            # rotation + encoder
            data = batch.to(dev)
            rot, y_rot = rotate_inputs(data)
            data["rotate_mat"] = rot
            print(f"  rotate            rot {d(rot, out['rotate_mat']):.3e}  y {d(y_rot, out['y_rot']):.3e}")
            model.encoder.capture_intermediates = True
            loc_e, di, do, li, lo = model.encoder(data=data, noise=NoiseSpec(seed=seed))
            im = model.encoder.last_intermediates
            print(f"  graph             E_aa {im['E_aa']} E_g {im['E_g']} E_la {im['E_la']}")
            print(f"  encoder           aa_out {d(im['aa_out'], mid['aa_out']):.3e}  latent {d(im['latent_ys'], mid['latent_ys']):.3e}  "
                  f"local {d(loc_e, mid['local_embed']):.3e}  diff_in {d(di, out['diff_in']):.3e}  diff_out {d(do, out['diff_out']):.3e}")
            for t in (20, 10, 0):
                print(f"     aa_out[t={t}] {d(im['aa_out'][t], mid['aa_out'][t]):.3e}   latent[idx={20 - t}] {d(im['latent_ys'][20 - t], mid['latent_ys'][20 - t]):.3e}")
            # aggregator fed with the reference's local embedding
            g = model.aggregator(data=data, local_embed=mid["local_embed"].to(dev))
            print(f"  aggregator        global {d(g, mid['global_embed']):.3e}")
            # whole model
            data = batch.to(dev)
            o = model(data, noise=NoiseSpec(seed=seed))
            print(f"  model.forward     loc {d(o['loc'], out['loc']):.3e}  pi {d(o['pi'], out['pi']):.3e}  "
                  f"diff_in {d(o['diff_in'], out['diff_in']):.3e}  reg_mask {bool((o['reg_mask'].cpu() == out['reg_mask']).all())}")
        except Exception:
            traceback.print_exc()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
