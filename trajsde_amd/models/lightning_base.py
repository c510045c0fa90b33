"""What the two glue modules share with the reference's `pl.LightningModule`s (models/model_base_mix_sde.py:22-207,
models/model_base_mix.py:22-209): the base class and the hooks `train.py:54-66` / `test.py:58` reach through a Trainer.

`ModuleBase` IS `pytorch_lightning.LightningModule` wherever that package can be imported -- `pl.Trainer.fit / test`
only accept such modules (train.py:54, test.py:58), so the model-level swap in the YAML needs it -- and plain
`torch.nn.Module` in this image, where the package is absent and `trajsde_amd.driver` spells the loops out.  The hooks
below work under both: `self.log` is Lightning's when there is one and a dict (`self.logged`) otherwise; `hparams` is
Lightning's `save_hyperparameters()` record or a plain deep copy; the checkpoint path of `test_epoch_end` comes from
the Trainer (`self.trainer._ckpt_path`, MODEL:152) or from `result_ckpt_path` set by the driver."""
import json
import os
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

try:                                                             # the reference pins pytorch-lightning==1.6.5 (env.yml:247)
    import pytorch_lightning as _pl
    ModuleBase = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:                                                # not in this image: the driver's own loops are used
    _pl = None
    ModuleBase = nn.Module
    HAVE_LIGHTNING = False


class LightningHooks(ModuleBase):
    """mixin-style base of PredictionModelSDENet / PredictionModel"""

    def _record_hparams(self, kwargs: dict) -> None:
        """MODEL:28 `self.save_hyperparameters()`: under Lightning its own record (what its checkpoints store), else a copy"""
        clean = {k: v for k, v in kwargs.items() if k != "init_seed"}
        if HAVE_LIGHTNING and hasattr(self, "save_hyperparameters"):
            try:
                self.save_hyperparameters(clean)
                return
            except Exception:
                pass
        try:
            self.hparams = deepcopy(clean)
        except AttributeError:                                   # a read-only property of the base: keep our own name
            self._hparams_copy = deepcopy(clean)

    # ---- logging (MODEL:112-113) ---------------------------------------------------------------------
    def log_value(self, name: str, value, **kw) -> None:
        """`self.log(name, value, ...)` under a Trainer; always mirrored in `self.logged` (the driver's loops read that)"""
        if not hasattr(self, "logged"):
            self.logged = {}
        self.logged[name] = value.detach() if torch.is_tensor(value) else value
        base_log = getattr(super(), "log", None)
        if base_log is not None and self._attached_trainer() is not None:
            base_log(name, value, **kw)

    def _attached_trainer(self):
        try:
            return self.trainer                                  # Lightning: raises / returns None when not attached
        except Exception:
            return getattr(self, "_trainer_stub", None)

    def _trainer_world_size(self) -> int:
        """ranks of the attached Trainer's strategy (1 without a Trainer)"""
        tr = self._attached_trainer()
        if tr is None:
            return 1
        for name in ("world_size", "num_devices"):
            try:
                n = int(getattr(tr, name))
            except Exception:
                continue
            if n > 1:
                return n
        return 1

    def _direct_accumulation(self) -> bool:
        """May the path loss write `.grad` itself (no gradient THROUGH autograd, so no per-parameter hook fires)?  Not under a
        Trainer that runs more than one rank: Lightning's DDP strategy wraps the module in torch DDP, whose reducer learns of a
        gradient from the AccumulateGrad hook of each parameter -- without them the ranks train unsynchronised or the reducer
        raises 'expected to have finished reduction' on the second step.  There the gradients are returned through autograd.
        (`driver.train` has no Trainer: it all-reduces the flat gradient buffer itself.)"""
        if not bool(getattr(self, "direct_grad_accumulation", True)):
            return False
        return self._trainer_world_size() <= 1

    def current_lr(self):
        """MODEL:113 logs `scheduler.get_lr()[0]`; the driver's training handles point `self.scheduler` at the live schedule"""
        sch = getattr(self, "scheduler", None)
        if sch is None:
            return None
        try:
            return float(sch.get_last_lr()[0])
        except Exception:
            return None

    # ---- test_epoch_end (MODEL:150-165) --------------------------------------------------------------
    def results_path(self) -> str:
        """`<checkpoint dir>/../out/result_<checkpoint stem>.json` (MODEL:152-163)"""
        tr = self._attached_trainer()
        ck = getattr(tr, "_ckpt_path", None) if tr is not None else None
        ck = ck or getattr(self, "result_ckpt_path", None)
        if ck is None:
            raise RuntimeError("test_epoch_end: no checkpoint path (neither trainer._ckpt_path nor model.result_ckpt_path): "
                               "the reference names its result file after the tested checkpoint (MODEL:152-163)")
        ck = Path(ck)
        out_dir = os.path.join(ck.parent.parent, "out")
        if not os.path.isdir(out_dir):
            os.mkdir(out_dir)
        return os.path.join(out_dir, f"result_{ck.stem}.json")

    def test_epoch_end(self, outputs=None) -> None:
        """the metric dump `test.py` leaves next to the checkpoint: {metric module name: value}.  `compute()` all-reduces the
        metric states, so EVERY rank calls this (as Lightning calls the hook on every rank); rank 0 writes the file."""
        metrics = {name: float(m.compute()) for name, m in zip(self.metric_names, self.metrics_vl)}
        self.write_results(metrics)

    def write_results(self, metrics: dict) -> None:
        """already reduced metric values -> the result JSON (no collective in here; rank 0 only)"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        with open(self.results_path(), "w") as f:
            json.dump(metrics, f)

    # ---- only_agent (MODEL:136-137, 168-202) -----------------------------------------------------------
    @staticmethod
    def leave_only_agent(data, output) -> None:
        """Cut batch and outputs down to the target agents, in place, exactly as MODEL:168-202 does (field by field, the
        same order): what `test_step` does first when the YAML sets `only_agent`.  Index plumbing on the batch; fields the
        stored scenes do not carry (`has_goal`, `goal_idcs`, `category`) are cut when present."""
        idx = data["agent_index"]
        data.num_nodes = data.x.size(0)                          # (MODEL:169 -- before x is cut, as there)
        data.bos_mask = data.bos_mask[idx]
        if data.y is not None:
            data.y = data.y[idx]
        data.x = data.x[idx]
        if "category" in data:
            data.category = data.category[idx]
        data.positions = data.positions[idx]
        data.rotate_mat = data.rotate_mat[idx]
        data.rotate_angles = data.rotate_angles[idx]
        if "has_goal" in data:
            data.has_goal = data.has_goal[idx]
        data.padding_mask = data.padding_mask[idx]
        lai = data["lane_actor_index"]
        al_agent_mask = torch.isin(lai[1], idx)
        agent_has_lane = torch.isin(idx, lai[1])
        if "goal_idcs" in data:
            data.goal_idcs = data.goal_idcs[al_agent_mask]
        data.lane_actor_vectors = data.lane_actor_vectors[al_agent_mask]
        output["loc"] = output["loc"][:, idx]
        output["pi"] = output["pi"][idx]
        output["reg_mask"] = output["reg_mask"][idx]
        for k, sel in (("cls_mask", idx), ("goal_prob", al_agent_mask), ("goal_cls_mask", al_agent_mask)):
            if k in output:
                output[k] = output[k][sel]
        lai = lai[:, al_agent_mask].clone()
        # actor ids -> positions in the agent list, one agent after the other like MODEL:194-196 (for the ascending agent lists
        # that collate produces, a new id can never equal a later agent's old one; any other list is rewritten as the reference
        # would rewrite it)
        for i, agent_i in enumerate(idx.tolist()):
            if bool(agent_has_lane[i]):
                lai[1][lai[1] == agent_i] = i
        data.lane_actor_index = lai
        n = data.x.size(0)
        dev = data.x.device
        data.agent_index = torch.arange(n, device=dev)
        data.av_index = torch.arange(n, device=dev)
        data.batch = torch.arange(n, device=dev)
