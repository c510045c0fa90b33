"""PredictionModel -- glue module of the vanilla HiVT variant of the path (reference models/model_base_mix.py:22-209):
same YAML registry, `forward(data) -> dict`, the same in-place side effects on `data` (rotate_mat, rotated y) and the
same validation / test step bookkeeping as the SDE model; the stages are the HIP-backed LocalEncoder,
GlobalInteractor and MLPDecoder.  Deterministic (no SDE noise).  `training_step` differentiates the shipped loss of
this configuration (L2, configs/nusargo/hivt_nuSArgo_trmenc_mlpdec.yml:62-66) through the HIP backward entry points
(trajsde_mlp_decoder_l2_backward -> trajsde_aggregator_backward_heads -> trajsde_encoder_grid_backward); the `ts_drop`
augmentation (models/model_base_mix.py:95-100) masks history steps of the batch before the forward, as there.  The YAML's
`nodecay` flag is stored and, as in the reference (no code reads it), has no effect: AdamW runs over all parameters.
"""
from copy import deepcopy
from typing import Optional

import torch
import torch.nn as nn

from trajsde_amd import runtime
from trajsde_amd.models.lightning_base import LightningHooks
from trajsde_amd.models.model_base_mix_sde import resolve_class


class _GridPathLoss(torch.autograd.Function):
    """w * L2 as one autograd node over the parameters (see model_base_mix_sde._PathLoss)"""

    @staticmethod
    def forward(ctx, model, data, noise, w_l2, *params):
        with torch.no_grad():
            out = model(data, noise=noise)
            local, glob = out["local_embed"], out["global_embed"]
            dec = model.decoder._rt.mlp_decoder_l2_backward(data, local, glob, out)
            agg = model.aggregator._rt.aggregator_backward(data, local, dec["d_global_embed"], noise)
            enc = model.encoder._rt.encoder_grid_backward(data, dec["d_local_embed"] + agg["d_local_embed"], noise)
            by_name = {"decoder." + n: g for n, g in dec["grads"].items()}
            by_name.update({"aggregator." + n: g for n, g in agg["grads"].items()})
            by_name.update({"encoder." + n: g for n, g in enc["grads"].items()})
            ctx.grads = [by_name.get(n) for n in model._param_names]
            ctx.w = w_l2
            model.last_output = out
            model.last_losses = {"L2": dec["loss"].detach()}
            return (w_l2 * dec["loss"]).clone()

    @staticmethod
    def backward(ctx, g):
        have = [x for x in ctx.grads if x is not None]
        scaled = iter(torch._foreach_mul(have, g * ctx.w))
        return (None, None, None, None) + tuple(None if x is None else next(scaled) for x in ctx.grads)


class PredictionModel(LightningHooks):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self._record_hparams(kwargs)                                      # models/model_base_mix.py:28 save_hyperparameters()
        init_seed: Optional[int] = kwargs.get("init_seed")
        for key, value in kwargs.items():
            if key == "training_specific":
                for k, v in value.items():
                    setattr(self, k, v)
            elif key == "model_specific":
                for k, v in value["kwargs"].items():
                    setattr(self, k, v)

        def build(section, offset):
            args = kwargs[section]
            kw = dict(args["kwargs"])
            if init_seed is not None:
                kw["init_seed"] = init_seed + offset
            return resolve_class(args["file_path"], args["module_name"])(**kw)

        self.encoder = build("encoder", 1)
        self.aggregator = build("aggregator", 2)
        self.decoder = build("decoder", 3)
        self.losses, self.loss_names = [], []
        for i, path in enumerate(kwargs.get("losses", [])):
            name = kwargs["losses_module"][i]
            self.losses.append(resolve_class(path, name)(**dict(kwargs["loss_args"][i])))
            self.loss_names.append(name)
        self.loss_weights = kwargs.get("loss_weights", [])
        self.metrics_tr, self.metrics_vl, self.metric_names = [], [], []
        for i, path in enumerate(kwargs.get("metrics", [])):
            name = kwargs["metrics_module"][i]
            metric = resolve_class(path, name)(**dict(kwargs["metric_args"][i]))
            self.metrics_tr.append(metric)
            self.metrics_vl.append(deepcopy(metric))
            self.metric_names.append(name)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def forward(self, data, noise: Optional["runtime.NoiseSpec"] = None):
        """models/model_base_mix.py:74-92.  `noise` (optional, ours): the key of the train-mode dropout masks; the default draws a fresh
        one from torch's global generator, like the reference's dropout draws fresh masks.  Eval mode uses no randomness."""
        if self.training:
            noise = runtime.NoiseSpec.resolve(noise)
        if not self.rotate:
            raise NotImplementedError("rotate=False is not built (shipped config: rotate: true)")
        if not runtime.consume_rotation(data):               # (done ahead of time by prefetch_graph)
            rotate_mat, y_rot = runtime.rotate_inputs(data)
            if y_rot is not None:
                data.y = y_rot
            data["rotate_mat"] = rotate_mat
        local_embed = self.encoder(data=data, noise=noise)
        global_embed = self.aggregator(data=data, local_embed=local_embed, noise=noise)
        return self.decoder(data=data, local_embed=local_embed, global_embed=global_embed)

    def prefetch_graph(self, data, noise: Optional["runtime.NoiseSpec"] = None, main_stream=None) -> None:
        """rotation + graph stage of the batch the training loop uses next, on the side stream (runtime.prefetch_graph; this variant's
        graph has no fake agents and does not depend on the step's noise)"""
        enc = self.encoder
        runtime.prefetch_graph(data, float(enc.local_radius), int(enc.historical_steps), runtime.NoiseSpec(seed=0), fake_agents=False,
                               main_stream=main_stream)

    def params_with_gradient(self):
        from trajsde_amd import _lib
        reached = set()
        for stage, sid in (("encoder", _lib.STAGE_ENCODER_GRID_BWD), ("aggregator", _lib.STAGE_AGGREGATOR_BWD),
                           ("decoder", _lib.STAGE_DECODER_MLP_BWD)):
            reached |= {f"{stage}.{n}" for n in getattr(self, stage)._rt.param_names(sid)}
        return [p for n, p in self.named_parameters() if n in reached]

    def apply_ts_drop(self, data, generator: Optional[torch.Generator] = None) -> None:
        """models/model_base_mix.py:96-100: drop history steps at random (probability `ts_drop`), never a step that begins a
        track (bos) nor the current one: the inputs of a dropped step are zeroed and the step is marked as padding -- in place,
        on the batch, like the reference does.  Index plumbing on the inputs; the kernels see an ordinary batch."""
        h = int(self.historical_steps)
        x = data.x
        mask = torch.rand(x.size(0), h, device=x.device, generator=generator) > (1 - float(self.ts_drop))
        mask[data.bos_mask] = False
        mask[:, -1] = False
        x[mask] = 0
        data.padding_mask[:, :h] = data.padding_mask[:, :h] | mask

    def training_step(self, data, batch_idx, noise=None):
        """models/model_base_mix.py:94-114 for the shipped loss (L2).  In train mode the stages' `dropout` (0.1 in the reference's YAML) is
        applied at the reference's 36 sites -- the four of every attention block and of every TemporalEncoder layer -- with masks cut
        from the Philox stream of `noise` (csrc/dropout.hpp); `model.eval()` switches it off."""
        if getattr(self, "ts_drop", False):
            self.apply_ts_drop(data)
        if self.loss_names != ["L2"]:
            raise NotImplementedError(f"training_step differentiates L2 through the HIP kernels; configured: {self.loss_names}")
        if not getattr(self.decoder, "uncertain", True):      # (losses/L2.py:12 chunks loc | scale out of four channels: see the SDE model)
            raise NotImplementedError("training with `uncertain: False` is not built: the reference's L2 regresses x against both targets "
                                      "on a two-channel output (losses/L2.py:12)")
        if data.y is None:
            raise ValueError("training_step needs targets (data.y)")
        noise = runtime.NoiseSpec.resolve(noise)
        if not hasattr(self, "_param_names"):
            self._param_names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        loss = _GridPathLoss.apply(self, data, noise, float(self.loss_weights[0]), *params)
        self.log_value("train/L2", self.last_losses["L2"], prog_bar=True, on_step=True, on_epoch=True,
                       batch_size=int(self.last_output["loc"].size(1)))          # models/model_base_mix.py:112
        lr = self.current_lr()
        if lr is not None:
            self.log_value("lr", lr, prog_bar=False, on_step=False, on_epoch=True, batch_size=1)
        return loss

    def configure_optimizers(self):
        """models/model_base_mix.py:205-208: AdamW + StepLR(scheduler_step, scheduler_gamma).  The shipped YAML does not
        define those two keys (the reference would fail there); without them the SDE model's cosine schedule is used."""
        self.optimizer = torch.optim.AdamW(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        if hasattr(self, "scheduler_step"):
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=self.scheduler_step, gamma=self.scheduler_gamma)
        else:
            self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.T_max, eta_min=0.0)
        return [self.optimizer], [self.scheduler]

    def _agent_eval_tensors(self, data, output):
        idx = data["agent_index"]
        return output["loc"][:, idx, :, :2], data.y[idx], output["reg_mask"][idx], data["source"]

    def validation_step(self, data, batch_idx):
        output = self(data)
        y_hat, y, mask, source = self._agent_eval_tensors(data, output)
        for metric in self.metrics_vl:
            metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
        return output

    def test_step(self, data, batch_idx):
        output = self(data)
        if getattr(self, "only_agent", False):                                # models/model_base_mix.py:136-137
            self.leave_only_agent(data, output)
        if data.y is not None:
            y_hat, y, mask, source = self._agent_eval_tensors(data, output)
            for metric in self.metrics_vl:
                metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
        return output

    def metric_results(self):
        return {n: float(m.compute()) for n, m in zip(self.metric_names, self.metrics_vl)}
