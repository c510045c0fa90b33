"""PredictionModelSDENet -- the glue module of the hot path, MI355X build.

Mirrors the reference's `PredictionModelSDENet` (models/model_base_mix_sde.py:22-207) at the boundary:
built from the same YAML dict, stages resolved through the same {file_path, module_name, kwargs}
registry (MODEL:38-45), `forward(data) -> dict` with the same keys and the same in-place side effects
on `data` (MODEL:83-85), `validation_step/test_step` feeding the same metric formulas.  It is a
`pytorch_lightning.LightningModule` wherever that package is importable (what `pl.Trainer.fit / test` of train.py:54-66,
test.py:58 require of the model class) and a plain nn.Module in this image, where it is not and trajsde_amd.driver spells
the loops out (models/lightning_base.py); `test_epoch_end`, `only_agent` / `leave_only_agent` and the `self.log` calls
of the reference's steps are there under both.
"""
import os
from copy import deepcopy
from importlib.machinery import SourceFileLoader
from typing import Optional

import torch
import torch.nn as nn

from trajsde_amd import runtime
from trajsde_amd.models.lightning_base import LightningHooks

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def resolve_class(file_path: str, module_name: str):
    """The reference's registry: getattr(SourceFileLoader(name, path).load_module(name), name)."""
    path = file_path if os.path.isfile(file_path) else os.path.join(_REPO_ROOT, file_path)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"stage file '{file_path}' not found (cwd or {_REPO_ROOT})")
    return getattr(SourceFileLoader(module_name, path).load_module(module_name), module_name)


class GradSet:
    """the three stages' gradient buffers of one training step: (name prefix, runtime.GradBuffers, multiplier) each.  The training
    loop's sink takes them whole (driver.FlatGrads.accumulate_bundles); `by_name()` spells them out per parameter for everyone else."""

    def __init__(self) -> None:
        self.bundles = []

    def add(self, prefix: str, grads, mult: float = 1.0) -> None:
        self.bundles.append((prefix, grads, float(mult)))

    def by_name(self) -> dict:
        out = {}
        for prefix, grads, mult in self.bundles:
            for n, g in grads.items():
                out[prefix + n] = g if mult == 1.0 else g * mult
        return out


class _PathLoss(torch.autograd.Function):
    """loss = sum_i w_i * loss_i over the configured {L2, DiffBCE} set as ONE autograd node whose inputs are the model
    parameters: forward runs the HIP forward and, right behind it, the three stage backward entry points of the
    C-ABI (decoder -> aggregator -> encoder); backward hands the gradients to autograd, so `loss.backward()`,
    optimizers, gradient accumulation and torch DDP hooks behave as with the reference's autograd graph."""

    @staticmethod
    def forward(ctx, model, data, noise, w_l2, w_diff, *params):
        """`w_l2`: the weight of the regression loss -- L2, or LaplaceNLLLoss when the model is configured with it"""
        with torch.no_grad():
            loss, gs = model._loss_and_gradients(data, noise, w_l2, w_diff)
            ctx.direct = model._direct_accumulation()
            ctx.sink = getattr(model, "_grad_sink", None) if ctx.direct else None
            ctx.n_inputs = len(params)
            ctx.model = model
            # the training loop's sink takes whole stage buffers: no per-parameter tensor is made for it
            ctx.gradset = gs if (isinstance(gs, GradSet) and ctx.sink is not None and hasattr(ctx.sink, "accumulate_bundles")) else None
            ctx.grads = None if ctx.gradset is not None else _PathLoss._spell_out(model, gs)
            return loss

    @staticmethod
    def _spell_out(model, gs):
        by_name = gs.by_name() if isinstance(gs, GradSet) else gs        # (a plain {name: gradient} dict is accepted too)
        return [by_name.get(n) for n in model._param_names]               # None: no path from these losses

    @staticmethod
    def backward(ctx, g):
        nothing = (None,) * (5 + ctx.n_inputs)
        if ctx.gradset is not None:
            if ctx.sink.accumulate_bundles(ctx.gradset.bundles, g):
                return nothing
            ctx.grads = _PathLoss._spell_out(ctx.model, ctx.gradset)       # (a layout the sink does not take whole)
        have = [x for x in ctx.grads if x is not None]
        if ctx.direct:
            # Leaf accumulation done here in two fused launches instead of one AccumulateGrad node (an add kernel and ~5 us of
            # host time) per parameter: `.grad += g * grad`, or `.grad = ...` where there is none yet -- what autograd's
            # accumulation would have left.  The parameters then receive no gradient THROUGH autograd, so per-parameter
            # hooks (torch DDP's reducer) do not fire: `model.direct_grad_accumulation = False` restores the plain route.
            params = [p for _, p in ctx.model.named_parameters()]
            sink = ctx.sink                                                  # driver.FlatGrads: all of it in six launches
            if sink is not None and sink.accumulate(params, ctx.grads, g):
                return nothing
            torch._foreach_mul_(have, g)                                     # our own buffers, fresh every step
            dst, src = [], []
            for p, x in zip(params, ctx.grads):
                if x is None or not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = x
                else:
                    dst.append(p.grad)
                    src.append(x)
            if dst:
                torch._foreach_add_(dst, src)
            return nothing
        scaled = iter(torch._foreach_mul(have, g))                          # a handful of fused launches, not one per tensor
        return (None, None, None, None, None) + tuple(None if x is None else next(scaled) for x in ctx.grads)


class PredictionModelSDENet(LightningHooks):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self._record_hparams(kwargs)                                      # MODEL:28 save_hyperparameters()
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

    def forward(self, data, noise: Optional["runtime.NoiseSpec"] = None, preserve_side_effects: Optional[bool] = None):
        """MODEL:74-102.  `noise` (optional, ours) selects the Philox seed or injected normals; the default
        draws a fresh Philox seed from torch's global generator, like the reference draws fresh noise.
        `preserve_side_effects=True` (or `preserve_side_effects: true` among the model / encoder kwargs of the YAML) makes
        the encoder also write `data['edge_index_{t}']`, `data['edge_attr_{t}']` (ENC:107-110); `rotate_mat` and the rotated
        `y` (MODEL:83-85) are always written."""
        if preserve_side_effects is None:
            preserve_side_effects = getattr(self, "preserve_side_effects", None)
        ood = bool(getattr(self, "ood", False))                              # test.py --ood injects this flag (test.py:45-46)
        noise = runtime.NoiseSpec.resolve(noise)
        if self.rotate:
            if not runtime.consume_rotation(data):
                rotate_mat, y_rot = runtime.rotate_inputs(data)          # MODEL:76-85
                if y_rot is not None:
                    data.y = y_rot
                data["rotate_mat"] = rotate_mat
        else:
            raise NotImplementedError("rotate=False is not built (shipped config: rotate: true, CFG:18)")
        prepared = None
        if ood:
            local_embed, stds = self.encoder.forward_ood(data=data, noise=noise)            # MODEL:89-90
        else:
            # the aggregator's relative-pose embedding depends on the graph stage alone: on a side stream that the encoder call
            # forks where its recurrence starts, it shares the chip with that serial kernel (runtime.arm_rel_prefetch; eval mode only)
            rt = getattr(self.aggregator, "_rt", None)
            side = rt.arm_rel_prefetch(data, self.encoder) if (not self.training and rt is not None) else None
            local_embed, diff_in, diff_out, label_in, label_out = self.encoder(data=data, noise=noise,
                                                                               preserve_side_effects=preserve_side_effects)
            prepared = rt.launch_rel_prefetch(data, side) if side is not None else None
        global_embed = (self.aggregator(data=data, local_embed=local_embed, noise=noise, prepared=prepared) if prepared is not None
                        else self.aggregator(data=data, local_embed=local_embed, noise=noise))
        out = self.decoder(data=data, local_embed=local_embed, global_embed=global_embed, noise=noise)
        if ood:
            out["stds"] = stds                                                                 # MODEL:97-98
        else:
            out["diff_in"], out["diff_out"], out["label_in"], out["label_out"] = diff_in, diff_out, label_in, label_out
        return out

    @staticmethod
    def check_range() -> None:
        """raise if a launch since the last check left the fp16 range of the split-precision products (csrc/range.hpp);
        synchronises the current stream -- call it where the host waits anyway (epoch end, when a loss is read)"""
        from trajsde_amd import _lib
        _lib.check_range()

    def _regression_loss(self):
        """(name, eps) of the configured regression loss: "L2" (losses/L2.py, the shipped one) or "LaplaceNLLLoss"
        (losses/laplace_nll_loss.py: the scale head is trained as well)"""
        for name, fn in zip(self.loss_names, self.losses):
            if name == "LaplaceNLLLoss":
                return name, float(getattr(fn, "eps", 1e-6))
        return "L2", None

    def params_with_gradient(self):
        """the parameters the configured losses reach (everything except the decoder's pi / scale heads and unused
        buffers-as-parameters): the reference's autograd leaves the others' `.grad` at None, so AdamW skips them"""
        from trajsde_amd import _lib
        reached = set()
        dec_stage = _lib.STAGE_DECODER_NLL_BWD if self._regression_loss()[0] == "LaplaceNLLLoss" else _lib.STAGE_DECODER_BWD
        for stage, sid in (("encoder", _lib.STAGE_ENCODER_BWD), ("aggregator", _lib.STAGE_AGGREGATOR_BWD),
                           ("decoder", dec_stage)):
            reached |= {f"{stage}.{n}" for n in getattr(self, stage)._rt.param_names(sid)}
        return [p for n, p in self.named_parameters() if n in reached]

    def _loss_and_gradients(self, data, noise, w_l2: float, w_diff: float):
        """the HIP forward and, right behind it, the three stage backward entry points of the C-ABI (decoder -> aggregator ->
        encoder): (weighted loss, {parameter name: gradient}).  Sets `last_output` / `last_losses` like the reference's step."""
        enc_rt, agg_rt, dec_rt = self.encoder._rt, self.aggregator._rt, self.decoder._rt
        for rt in (enc_rt, agg_rt, dec_rt):                   # no parameter changes inside this call: one stamp walk per stage
            rt.pin_stamp()
        try:
            if runtime.single_call_forms():
                self._step_pack_set().refresh()               # the six weight images of the step in one call (runtime.PackSet)
            return self._loss_and_gradients_pinned(data, noise, w_l2, w_diff)
        finally:
            for rt in (enc_rt, agg_rt, dec_rt):
                rt.unpin_stamp()

    def _step_pack_set(self) -> "runtime.PackSet":
        """forward and backward images of the three stages, as one packing call per optimizer step"""
        from trajsde_amd import _lib
        dec_stage = _lib.STAGE_DECODER_NLL_BWD if self._regression_loss()[0] == "LaplaceNLLLoss" else _lib.STAGE_DECODER_BWD
        ps = self.__dict__.get("_pack_set_obj")
        if ps is None or ps.entries[-1][1] != dec_stage:
            enc_rt, agg_rt, dec_rt = self.encoder._rt, self.aggregator._rt, self.decoder._rt
            ps = runtime.PackSet([(enc_rt, _lib.STAGE_ENCODER), (enc_rt, _lib.STAGE_ENCODER_BWD), (agg_rt, _lib.STAGE_AGGREGATOR),
                                  (agg_rt, _lib.STAGE_AGGREGATOR_BWD), (dec_rt, _lib.STAGE_DECODER), (dec_rt, dec_stage)])
            self.__dict__["_pack_set_obj"] = ps
        return ps

    def _loss_and_gradients_pinned(self, data, noise, w_l2: float, w_diff: float):
        enc_rt, agg_rt, dec_rt = self.encoder._rt, self.aggregator._rt, self.decoder._rt
        reg = self._regression_loss()
        # one forward per step: the encoder and aggregator run their tape-keeping forward, the backward entry points
        # then walk those tapes instead of recomputing the stage (runtime.*_forward_train)
        out, local, glob, enc_tape, agg_tape = self._forward_stages(data, noise, keep_tapes=True)
        if reg[0] == "LaplaceNLLLoss":
            dec = dec_rt.decoder_nll_backward(data, local, glob, out, noise, eps=reg[1])
        else:
            dec = dec_rt.decoder_l2_backward(data, local, glob, out, noise)
        d_glob, d_local = dec["d_global_embed"], dec["d_local_embed"]
        if w_l2 != 1.0:
            d_glob, d_local = d_glob * w_l2, d_local * w_l2
        agg = agg_rt.aggregator_backward(data, local, d_glob, noise, tape=agg_tape)
        del agg_tape
        # multi-rank training loop (driver.train): the decoder's and aggregator's gradients are final -- their slice of the flat
        # gradient buffer goes to the all-reduce now, on the collective stream, under the encoder backward (driver.FlatGrads)
        gs = GradSet()
        gs.add("decoder.", dec["grads"], w_l2)
        gs.add("aggregator.", agg["grads"])
        sink = getattr(self, "_grad_sink", None)
        if sink is not None and getattr(sink, "early_enabled", False) and self._direct_accumulation():
            if not (hasattr(sink, "early_reduce_bundles") and sink.early_reduce_bundles(gs.bundles)):
                early = gs.by_name()                                         # (a sink without the whole-buffer entry points)
                named = dict(self.named_parameters())
                sink.early_reduce([named[n] for n in early], [early[n] for n in early])
        enc = enc_rt.encoder_backward(data, d_local + agg["d_local_embed"], noise, diff_weight=w_diff, tape=enc_tape)
        del enc_tape
        gs.add("encoder.", enc["grads"])
        self.last_output = out
        self.last_losses = {reg[0]: dec["loss"].detach(), "DiffBCE": enc["diff_loss"].detach() / w_diff if w_diff else None}
        return (w_l2 * dec["loss"] + enc["diff_loss"]).clone(), gs

    def prefetch_graph(self, data, noise: "runtime.NoiseSpec", main_stream=None) -> None:
        """prepare `data` for the training_step that will follow with the same `noise`: rotation + graph stage on the side stream
        (runtime.prefetch_graph; driver.train calls it for batch i + 1 right after it has enqueued step i)"""
        enc = self.encoder
        runtime.prefetch_graph(data, float(enc.local_radius), int(enc.historical_steps), runtime.NoiseSpec.resolve(noise),
                               main_stream=main_stream)

    def _forward_stages(self, data, noise, keep_tapes: bool = False):
        """forward() that also hands back the two stage boundaries the backward entry points need; with `keep_tapes` the
        encoder and the aggregator run their tape-keeping forward and the tapes are returned too"""
        if not runtime.consume_rotation(data):               # (done ahead of time by prefetch_graph)
            rotate_mat, y_rot = runtime.rotate_inputs(data)
            if y_rot is not None:
                data.y = y_rot
            data["rotate_mat"] = rotate_mat
        enc_tape = agg_tape = None
        if keep_tapes:
            (local_embed, diff_in, diff_out, label_in, label_out), enc_tape = self.encoder._rt.encoder_forward_train(data, noise)
            global_embed, agg_tape = self.aggregator._rt.aggregator_forward_train(data, local_embed, noise)
        else:
            local_embed, diff_in, diff_out, label_in, label_out = self.encoder(data=data, noise=noise)
            global_embed = self.aggregator(data=data, local_embed=local_embed, noise=noise)
        out = self.decoder(data=data, local_embed=local_embed, global_embed=global_embed, noise=noise)
        out["diff_in"], out["diff_out"], out["label_in"], out["label_out"] = diff_in, diff_out, label_in, label_out
        if keep_tapes:
            return out, local_embed, global_embed, enc_tape, agg_tape
        return out, local_embed, global_embed

    # -- Lightning-style hooks (MODEL:104-148) ------------------------------------------------------
    def training_step(self, data, batch_idx, noise: Optional["runtime.NoiseSpec"] = None):
        """MODEL:104-116: forward, the weighted sum of the configured losses, as a tensor whose `.backward()` fills
        `.grad` through the HIP backward kernels.  In train mode (`model.train()`) the stages' `dropout` is applied at the
        reference's sites (attention weights, out_proj output, the two FFN activations of every attention block) with masks
        cut from the Philox stream of `noise` (csrc/dropout.hpp); `model.eval()` switches it off, as in the reference.  The kernels implement the
        shipped loss set (losses/L2.py + losses/diff_BCE.py, CFG:78-83) and losses/laplace_nll_loss.py in place of L2; any other loss
        is refused rather than silently differentiated elsewhere."""
        if not self.rotate:
            raise NotImplementedError("rotate=False is not built (shipped config: rotate: true, CFG:18)")
        if not getattr(self.decoder, "uncertain", True):
            # losses/L2.py:12 and losses/laplace_nll_loss.py:28 take `loc, scale = output['loc'].chunk(2, dim=-1)`: on the
            # two-channel output of `uncertain: False` that makes `loc` the x coordinate alone, broadcast against BOTH target
            # coordinates -- the reference's losses are only meaningful with the scale head.  Inference is supported.
            raise NotImplementedError("training with `uncertain: False` is not built: the reference's own losses chunk loc | scale out of "
                                      "FOUR channels (losses/L2.py:12); on the two-channel output they regress x against both targets")
        weights = dict(zip(self.loss_names, self.loss_weights))
        unknown = set(self.loss_names) - {"L2", "LaplaceNLLLoss", "DiffBCE"}
        reg_name = self._regression_loss()[0]
        if unknown or reg_name not in weights or ("L2" in weights and "LaplaceNLLLoss" in weights):
            raise NotImplementedError("training_step differentiates ONE regression loss (L2 or LaplaceNLLLoss) + DiffBCE through the HIP "
                                      f"kernels; configured: {self.loss_names}")
        if data.y is None:
            raise ValueError("training_step needs targets (data.y)")
        noise = runtime.NoiseSpec.resolve(noise)
        if not hasattr(self, "_param_names"):
            self._param_names = [n for n, _ in self.named_parameters()]
        direct = self._direct_accumulation()
        # (named_parameters() walks the ~270 modules of the tree: once per step on the autograd route, not at all on the direct one,
        #  whose single input is looked up once)
        one = self.__dict__.get("_one_param")               # (kept out of nn.Module's parameter registry: plain instance dict)
        if direct and one is not None and one.requires_grad:
            params = [one]
        else:
            params = [p for _, p in self.named_parameters()]
        if direct:
            # the gradients do not travel through autograd (`.grad` is written directly): ONE parameter as the node's input is
            # enough to make the loss differentiable, and 250 fewer inputs are 250 fewer edges for the engine to walk every step
            params = [p for p in params if p.requires_grad][:1]
            self.__dict__["_one_param"] = params[0] if params else None
        loss = _PathLoss.apply(self, data, noise, float(weights[reg_name]), float(weights.get("DiffBCE", 0.0)), *params)
        n_rows = int(self.last_output["loc"].size(1))
        for name in self.loss_names:                                          # MODEL:112: one entry per configured loss
            if self.last_losses.get(name) is not None:
                self.log_value(f"train/{name}", self.last_losses[name], prog_bar=True, on_step=True, on_epoch=True, batch_size=n_rows)
        lr = self.current_lr()
        if lr is not None:                                                    # MODEL:113 (once configure_optimizers has run)
            self.log_value("lr", lr, prog_bar=False, on_step=False, on_epoch=True, batch_size=1)
        return loss

    def _agent_eval_tensors(self, data, output):
        idx = data["agent_index"]
        return output["loc"][:, idx, :, :2], data.y[idx], output["reg_mask"][idx], data["source"]

    def validation_step(self, data, batch_idx):
        output = self(data)
        y_hat, y, mask, source = self._agent_eval_tensors(data, output)
        if not self.is_gtabs:
            y_hat, y = torch.cumsum(y_hat, dim=-2), torch.cumsum(y, dim=-2)
        for metric in self.metrics_vl:
            metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
        return output

    def test_step(self, data, batch_idx):
        output = self(data)
        if getattr(self, "only_agent", False):                                # MODEL:136-137
            self.leave_only_agent(data, output)
        if data.y is not None:
            y_hat, y, mask, source = self._agent_eval_tensors(data, output)
            for metric in self.metrics_vl:
                metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
        return output

    def metric_results(self):
        return {n: float(m.compute()) for n, m in zip(self.metric_names, self.metrics_vl)}

    def configure_optimizers(self):
        """AdamW + per-epoch cosine annealing (MODEL:204-207)."""
        self.optimizer = torch.optim.AdamW(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.T_max, eta_min=0.0)
        return [self.optimizer], [self.scheduler]
