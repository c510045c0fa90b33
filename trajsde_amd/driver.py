"""Minimal evaluation driver for the MI355X build, mirroring the reference's `test.py:39-58`
(YAML load -> class resolve through the {file_path, module_name, kwargs} registry -> optional checkpoint ->
loop of `test_step` over batches -> metrics JSON; `MODEL:133-165`).  pytorch_lightning / PyG are not in the image,
so the loop is spelled out; the model class keeps Lightning's hook signatures and also runs under a real Trainer.

    python -m trajsde_amd.driver --config trajsde_amd/configs/mi355x_sde_encoder_decoder.yml \
        --synthetic config1 [--batches 4] [--ckpt path.ckpt] [--ood] [--gpus N via torch.distributed.run]
    python -m trajsde_amd.driver --config ... --data [--nu_dir D --argo_dir D]      # the YAML's data module

With `--data` the `datamodule_specific` section is resolved like `test.py:54-58` does and the model is evaluated on
its `test_dataloader()` (flat scene shards, trajsde_amd/dataset.py); ranks take disjoint scene sets.

Training (`train.py:42-66`) needs the backward kernels and is not built yet.
"""
import argparse
import json
import os
from typing import Iterable, Optional

import torch
import yaml

from trajsde_amd.models.model_base_mix_sde import resolve_class
from trajsde_amd.runtime import NoiseSpec
from trajsde_amd.synth import CONFIGS, synth


def build_model(cfg: dict, ckpt: Optional[str] = None, device="cuda", init_seed: Optional[int] = None):
    ms = cfg["model_specific"]
    kwargs = dict(cfg)
    if init_seed is not None:
        kwargs["init_seed"] = init_seed
    model = resolve_class(ms["file_path"], ms["module_name"])(**kwargs)          # test.py:48-49
    if ckpt is not None:
        state = torch.load(ckpt, map_location="cpu")
        model.load_state_dict(state.get("state_dict", state))                       # Lightning checkpoints nest it
    return model.eval().to(device)


@torch.no_grad()
def evaluate(model, batches: Iterable, seed: int = 0) -> dict:
    """trainer.test(...): test_step per batch, then the epoch-end metric dump (MODEL:133-165)."""
    for metric in model.metrics_vl:
        metric.reset()
    for i, batch in enumerate(batches):
        model.test_step(batch, i) if seed is None else _seeded_test_step(model, batch, i, seed + i)
    return model.metric_results()


def _seeded_test_step(model, data, batch_idx, seed):
    output = model(data, noise=NoiseSpec(seed=seed))
    if data.y is not None:
        y_hat, y, mask, source = model._agent_eval_tensors(data, output)
        for metric in model.metrics_vl:
            metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
    return output


def synthetic_batches(name: str, n: int, device, rank: int = 0, world: int = 1):
    spec = CONFIGS[name]
    for i in range(rank, n, world):                                                   # scene-batches shard over ranks
        yield synth(**dict(spec["synth"], seed=spec["synth"]["seed"] + 97 * i)).to(device)


def datamodule_batches(cfg: dict, device, rank: int = 0, world: int = 1, nu_dir=None, argo_dir=None):
    """test.py:54-58: resolve the data module through the registry, iterate its test loader"""
    dm_cfg = cfg["datamodule_specific"]
    kwargs = dict(dm_cfg["kwargs"], rank=rank, world_size=world, device=device)
    if nu_dir:
        kwargs["nu_dir"] = nu_dir
    if argo_dir:
        kwargs["Argo_dir"] = argo_dir
    dm = resolve_class(dm_cfg["file_path"], dm_cfg["module_name"])(**kwargs)
    dm.setup("test")
    return dm.test_dataloader()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--synthetic", default="config1", choices=sorted(CONFIGS))
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--ood", action="store_true")
    ap.add_argument("--data", action="store_true", help="evaluate on the YAML's data module instead of synthetic batches")
    ap.add_argument("--nu_dir", default=None)
    ap.add_argument("--argo_dir", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    if not args.data:
        spec = CONFIGS[args.synthetic]
        K, T = spec["num_modes"], spec["future_steps"]
        cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
        cfg["aggregator"]["kwargs"]["num_modes"] = K
        cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T, max_fut_t=spec["max_fut_t"])
        for m in cfg["metric_args"]:
            m["end_idcs"] = [T - 1, T - 1]
    if args.ood:
        cfg["model_specific"]["kwargs"]["ood"] = True                                # test.py:45-46
    dev = torch.device("cuda", local_rank)
    model = build_model(cfg, args.ckpt, dev, init_seed=0 if args.ckpt is None else None)
    if args.data:
        batches = datamodule_batches(cfg, dev, rank, world, args.nu_dir, args.argo_dir)
    else:
        batches = synthetic_batches(args.synthetic, args.batches, dev, rank, world)
    res = evaluate(model, batches)                                                    # metric states all-reduce in compute()
    if rank == 0:
        text = json.dumps(res)
        if args.out:
            with open(args.out, "w") as f:
                f.write(text)
        print(text)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
