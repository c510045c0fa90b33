"""Minimal evaluation driver for the MI355X build, mirroring the reference's `test.py:39-58`
(YAML load -> class resolve through the {file_path, module_name, kwargs} registry -> optional checkpoint ->
loop of `test_step` over batches -> metrics JSON; `MODEL:133-165`).  pytorch_lightning / PyG are not in the image,
so the loop is spelled out here; where pytorch_lightning IS importable the model classes derive from
`pl.LightningModule` (models/lightning_base.py) and `pl.Trainer.fit / test` take them as train.py / test.py do.

    python -m trajsde_amd.driver --config trajsde_amd/configs/mi355x_sde_encoder_decoder.yml \
        --synthetic config1 [--batches 4] [--ckpt path.ckpt] [--ood] [--gpus N via torch.distributed.run]
    python -m trajsde_amd.driver --config ... --data [--nu_dir D --argo_dir D]      # the YAML's data module

With `--data` the `datamodule_specific` section is resolved like `test.py:54-58` does and the model is evaluated on
its `test_dataloader()` (flat scene shards, trajsde_amd/dataset.py); ranks take disjoint scene sets.

    python -m trajsde_amd.driver --config ... --train --synthetic config1 --batches 8 --epochs 2    # train.py:42-66

`--train` runs the reference's optimisation recipe (AdamW + per-epoch cosine annealing, MODEL:204-207) over
`training_step`; with WORLD_SIZE > 1 every rank trains on its own scenes and the gradients are averaged with ONE
all-reduce per step over a flat bucket that the parameters' `.grad` tensors are views of (RCCL over xGMI: ~1.4 MB,
latency-bound, so a single collective beats bucketing).
"""
import argparse
import inspect
import json
import os
import sys
from typing import Iterable, Optional

import torch
import yaml

from trajsde_amd.models.model_base_mix_sde import resolve_class
from trajsde_amd import runtime
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
    stochastic = "noise" in inspect.signature(model.forward).parameters      # the vanilla HiVT variant has no noise to seed
    rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    for i, batch in enumerate(batches):
        if seed is None or not stochastic:
            model.test_step(batch, i)
        else:
            _seeded_test_step(model, batch, i, seed + i + RANK_SEED_STRIDE * rank)
    if hasattr(model, "check_range"):
        model.check_range()                                   # no silently saturated fp16x3 operand in this epoch
    return model.metric_results()


def _seeded_test_step(model, data, batch_idx, seed):
    output = model(data, noise=NoiseSpec(seed=seed))
    if data.y is not None:
        y_hat, y, mask, source = model._agent_eval_tensors(data, output)
        for metric in model.metrics_vl:
            metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
    return output


class FlatGrads:
    """One contiguous gradient buffer; every parameter's `.grad` is a view into it, so zeroing is one memset and
    the data-parallel gradient sync is one all-reduce (the only collective of the training path)."""

    def __init__(self, params) -> None:
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, device=ref.device, dtype=ref.dtype)
        off = 0
        self.views, self.offsets = [], []
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(p.grad)
            self.offsets.append(off)
            off += p.numel()
        self._gather = {}

    def zero(self) -> None:
        self.flat.zero_()
        self._early = None
        self._early_prefixes = None
        self._early_taken = False

    # ---- the gradient all-reduce in two slices, the first one under the encoder backward ----
    # The stage backward entry points run decoder -> aggregator -> encoder; the decoder's and the aggregator's gradients are final
    # when the encoder backward -- two thirds of the backward's time -- is only being enqueued.  `early_reduce` (called by the path
    # loss between the two, when `early_enabled`) adds those gradients into their block of `flat` and starts that block's
    # all-reduce on the collective stream; `accumulate` later skips what is already in, and `all_reduce_mean` reduces the rest and
    # waits.  On two ranks every element is the same two-operand sum whatever piece it travels in (bit-equal in the gloo test);
    # on larger rings the position of an element inside its piece picks the rank that starts its sum, so the two forms may then
    # differ in the last place -- like any change of bucket size under torch DDP.
    #
    # Protocol, per zero(): at most ONE early_reduce, then ONE accumulate, then all_reduce_mean.  Gradient accumulation over
    # micro-batches does not fit that (the early block would be reduced before the later micro-batches are in it): a second
    # early_reduce or a second accumulate before zero() RAISES instead of dropping gradients.  The early block also assumes that
    # backward() arrives with a unit gradient (driver.train never scales the loss).
    early_enabled = False
    _early = None
    _early_taken = False
    # bench.py / tests: issue the collectives with ONE rank too (an RCCL all-reduce over a one-rank communicator is a real launch on
    # the collective stream), so the stream choreography below is exercised on a one-GPU box
    force_collective = False

    def _dist(self):
        """torch.distributed when a collective is due (more than one rank, or `force_collective`), else None"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective):
            return dist
        return None

    def _block_of(self, params, grads):
        """(ids, lo, hi) of the parameters among `params` that hold a gradient AND are in this buffer (frozen parameters and
        parameters the losses do not reach are not), if they form ONE block of `flat`"""
        own = {id(p) for p in self.params}
        have = {id(p) for p, g in zip(params, grads) if g is not None and id(p) in own}
        idx = [i for i, p in enumerate(self.params) if id(p) in have]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        lo = self.offsets[idx[0]]
        hi = self.offsets[idx[-1]] + self.params[idx[-1]].numel()
        return have, lo, hi

    def early_plan(self, names_of_params: dict, prefixes=("decoder.", "aggregator.")) -> bool:
        """Decided ONCE, from static properties (the parameter list and its names), whether the early slice applies: the
        parameters under `prefixes` that are in this buffer must form one block of it.  Every rank holds the same model, so
        every rank decides alike -- the sequence of collectives can not differ between ranks.  `names_of_params`: id -> name."""
        sub = [p for p in self.params if names_of_params.get(id(p), "").startswith(tuple(prefixes))]
        return bool(sub) and self._block_of(sub, [True] * len(sub)) is not None

    def early_reduce(self, params, grads) -> bool:
        """accumulate the given (final) gradients with scale 1 and start the all-reduce of their block.  The caller guarantees
        that `backward()` will be called with a unit gradient (driver.train does: it never scales the loss).  False (nothing
        done) when the gradients are not one block of the buffer in flat-buffer layout -- a static property of the model."""
        if self._early is not None:
            raise RuntimeError("FlatGrads.early_reduce: a block of this buffer was already reduced since zero() -- the early slice "
                               "serves ONE backward per zero(); switch `early_enabled` off for gradient accumulation over micro-batches")
        blk = self._block_of(params, grads)
        if blk is None:
            return False
        ids, lo, hi = blk
        sub_p = [p for p in params if id(p) in ids]
        sub_g = [g for p, g in zip(params, grads) if id(p) in ids]
        one = torch.ones((), device=self.flat.device, dtype=self.flat.dtype)
        if not self._accumulate_runs(sub_p, sub_g, one):
            return False
        work = self._start_early_collective(lo, hi)
        self._early = (ids, lo, hi, work)
        self._early_prefixes = None
        self._early_taken = False
        return True

    def accumulate(self, params, grads, scale) -> bool:
        """`.grad += scale * g` for all parameters at once, when the gradients arrive as views of a few flat buffers (what the
        stage backward entry points fill: runtime.StageRuntime._grad_buffers) and every parameter still holds the slice of
        `flat` it was given: per source buffer one gather into this buffer's order and one fused multiply-add on the block of
        `flat` that buffer's parameters cover -- six launches a step instead of ~250 slice pairs walked by multi-tensor kernels.
        `params` / `grads`: parallel lists (a gradient may be None); `scale`: 0-dim tensor.  Returns False, having done
        nothing, whenever the layout is not that simple (the caller then accumulates parameter by parameter)."""
        own = {id(p) for p in self.params}
        if self._early is not None:
            if self._early_taken:
                raise RuntimeError("FlatGrads.accumulate: second backward since zero() while a block was reduced early -- its "
                                   "gradients for that block would be dropped; switch `early_enabled` off for gradient accumulation")
            done = self._early[0]
        else:
            done = ()
        pairs = [(p, g) for p, g in zip(params, grads) if g is not None and id(p) in own and id(p) not in done]
        if len(pairs) + len(done) != len(self.params):
            if self._early is not None:
                raise RuntimeError("FlatGrads: gradients reduced early do not add up with the rest to the parameter list")
            return False
        ok = self._accumulate_runs([p for p, _ in pairs], [g for _, g in pairs], scale)
        if not ok and self._early is not None:
            raise RuntimeError("FlatGrads: a block was reduced early but the rest of the gradients is not in flat-buffer layout")
        if ok and self._early is not None:
            self._early_taken = True
        return ok

    # ---- the same two entry points over WHOLE stage buffers ("bundles": (name prefix, runtime.GradBuffers, multiplier)) -------------
    # A stage's gradients arrive as one flat tensor with a static layout, and the parameters of a stage are one stretch of this
    # buffer: the gather index of a (stage, layout) pair is derived once, from names alone, and every later step is ONE
    # index_select + addcmul per stage -- no per-parameter view is ever made, none is walked (0.5 ms of host time a step).
    # Needs the parameters' names (`names`: id -> dotted name, set by FlatTraining / PlainTraining); False = not applicable.
    names = None

    def _bundle_plan(self, prefix, gb):
        """(first offset in `flat`, gather index into gb.flat, ids of the parameters covered) or None if they are not one block"""
        key = ("bundle", prefix, id(gb.layout))
        plan = self._gather.get(key)
        if plan is None:
            if self.names is None:
                return None
            lay = gb.layout
            idx, pieces, ids = [], [], []
            for i, p in enumerate(self.params):
                name = self.names.get(id(p), "")
                if not name.startswith(prefix):
                    continue
                j = lay.index.get(name[len(prefix):])
                if j is None or lay.sizes[j] != p.numel() or p.grad is not self.views[i]:
                    return None
                idx.append(i)
                pieces.append((lay.offs[j], lay.sizes[j]))
                ids.append(id(p))
            if not idx or idx != list(range(idx[0], idx[-1] + 1)):
                return None
            gather = torch.cat([torch.arange(o, o + n, dtype=torch.int64) for o, n in pieces]).to(self.flat.device)
            plan = (self.offsets[idx[0]], gather, frozenset(ids))
            self._gather[key] = plan
        return plan

    def _apply_bundle(self, plan, gb, scale) -> None:
        first, gather, _ = plan
        self.flat[first:first + gather.numel()].addcmul_(gb.flat.index_select(0, gather), scale)

    def _apply_bundles(self, plans, bundles, scale) -> None:
        """flat[block] += stage buffer[gather] * (scale * mult) for every (plan, (prefix, buffers, mult)); on a GPU all of them in ONE
        launch (trajsde_grad_gather_add) -- two element-wise launches per stage otherwise.  `scale`: a 0-dim tensor or None for 1"""
        gpu = runtime.single_call_forms() and self.flat.is_cuda and all(gb.flat.is_cuda and gb.flat.dtype == torch.float32 and gb.flat.is_contiguous() for _, gb, _ in bundles)
        if gpu and (scale is None or (scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1)) and 0 < len(plans) <= 8:
            from trajsde_amd import _lib
            items = (_lib.GatherItem * len(plans))()
            base = self.flat.data_ptr()
            for it, (first, gather, _), (_, gb, mult) in zip(items, plans, bundles):
                it.dst, it.src, it.index, it.n, it.mult = base + 4 * first, gb.flat.data_ptr(), gather.data_ptr(), gather.numel(), float(mult)
            with torch.cuda.device(self.flat.device):
                _lib.check(_lib.lib().trajsde_grad_gather_add(items, len(plans), None if scale is None else scale.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "trajsde_grad_gather_add")
            return
        one = torch.ones((), device=self.flat.device, dtype=self.flat.dtype) if scale is None else scale
        for pl, (_, gb, mult) in zip(plans, bundles):
            self._apply_bundle(pl, gb, one if mult == 1.0 else one * mult)

    def early_reduce_bundles(self, bundles) -> bool:
        """early_reduce over whole stage buffers (decoder + aggregator); same protocol"""
        if self._early is not None:
            raise RuntimeError("FlatGrads.early_reduce: a block of this buffer was already reduced since zero() -- the early slice "
                               "serves ONE backward per zero(); switch `early_enabled` off for gradient accumulation over micro-batches")
        if self.flat.dtype != torch.float32:
            return False
        plans = [self._bundle_plan(prefix, gb) for prefix, gb, _ in bundles]
        if not plans or any(pl is None for pl in plans):
            return False
        spans = sorted((pl[0], pl[0] + pl[1].numel()) for pl in plans)
        if any(a[1] != b[0] for a, b in zip(spans, spans[1:])):
            return False                                     # the stages' blocks are not adjacent: not one slice of the buffer
        lo, hi = spans[0][0], spans[-1][1]
        self._apply_bundles(plans, bundles, None)
        self._early = (frozenset().union(*[pl[2] for pl in plans]), lo, hi, self._start_early_collective(lo, hi))
        self._early_prefixes = tuple(prefix for prefix, _, _ in bundles)
        self._early_taken = False
        return True

    def accumulate_bundles(self, bundles, scale) -> bool:
        """accumulate over whole stage buffers; False (nothing done) when the layout does not allow it"""
        if self.flat.dtype != torch.float32:
            return False
        early = ()
        if self._early is not None:
            if self._early_taken:
                raise RuntimeError("FlatGrads.accumulate: second backward since zero() while a block was reduced early -- its "
                                   "gradients for that block would be dropped; switch `early_enabled` off for gradient accumulation")
            early = getattr(self, "_early_prefixes", None)
            if early is None:
                return False                                 # the early block went in through the per-parameter route: finish there
        todo = [(prefix, gb, mult) for prefix, gb, mult in bundles if prefix not in early]
        plans = [self._bundle_plan(prefix, gb) for prefix, gb, _ in todo]
        if any(pl is None for pl in plans):
            if self._early is not None:
                raise RuntimeError("FlatGrads: a block was reduced early but the rest of the gradients is not in flat-buffer layout")
            return False
        covered = sum(len(pl[2]) for pl in plans) + (len(self._early[0]) if self._early is not None else 0)
        if covered != len(self.params):
            if self._early is not None:
                raise RuntimeError("FlatGrads: gradients reduced early do not add up with the rest to the parameter list")
            return False
        self._apply_bundles(plans, todo, scale)
        if self._early is not None:
            self._early_taken = True
        return True

    def _start_early_collective(self, lo, hi):
        """the all-reduce of flat[lo:hi] on the collective stream (None: no collective due)"""
        work = None
        dist = self._dist()
        if dist is not None:
            piece = self.flat[lo:hi]
            if self.flat.is_cuda:
                main = torch.cuda.current_stream(self.flat.device)
                coll = runtime.collective_stream(self.flat.device)   # its own stream: not queued behind the next batch's prefetch
                coll.wait_stream(main)                       # the block is complete on the main stream up to here
                with torch.cuda.stream(coll):
                    work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)
            else:
                work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)
        return work

    def _accumulate_runs(self, params, grads, scale) -> bool:
        by_id = {id(p): g for p, g in zip(params, grads) if g is not None}
        runs = []                                        # (source flat, first offset in self.flat, [(offset in source, numel)])
        for p, view, off in zip(self.params, self.views, self.offsets):
            g = by_id.get(id(p))
            if g is None:
                continue
            if p.grad is not view or g._base is None or not g.is_contiguous() or g.numel() != p.numel() \
                    or g.dtype != self.flat.dtype or g._base.dim() != 1:
                return False
            if not runs or runs[-1][0] is not g._base:
                if any(r[0] is g._base for r in runs):
                    return False                         # a source buffer's parameters are not one block of `flat`
                runs.append((g._base, off, []))
            elif off != runs[-1][1] + sum(n for _, n in runs[-1][2]):
                return False                             # ... or not contiguous in it
            runs[-1][2].append((g.storage_offset() - g._base.storage_offset(), p.numel()))
        for src, first, pieces in runs:
            key = (first, src.numel(), tuple(pieces))
            idx = self._gather.get(key)
            if idx is None:
                idx = torch.cat([torch.arange(o, o + n, dtype=torch.int64) for o, n in pieces]).to(self.flat.device)
                if len(self._gather) > 8:
                    self._gather.clear()
                self._gather[key] = idx
            self.flat[first:first + idx.numel()].addcmul_(src.index_select(0, idx), scale)
        return True

    def all_reduce_mean(self) -> None:
        dist = self._dist()
        if dist is None:
            self._early = None
            return
        if self._early is None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        else:
            _, lo, hi, work = self._early
            if lo > 0:
                dist.all_reduce(self.flat[:lo], op=dist.ReduceOp.SUM)
            if hi < self.flat.numel():
                dist.all_reduce(self.flat[hi:], op=dist.ReduceOp.SUM)
            if work is not None:
                work.wait()                              # (CUDA: the current stream waits for the collective stream's all-reduce)
            self._early = None
        if dist.get_world_size() > 1:
            self.flat.div_(dist.get_world_size())


class FlatAdamW(torch.optim.AdamW):
    """torch.optim.AdamW whose step over a contiguous fp32 GPU tensor is ONE launch (trajsde_adamw_step: torch's operations in torch's
    order, its scalars formed here as torch forms them) instead of nine element-wise ones -- 2 MB of parameters make each of them ~5 us
    of GPU time and ~15 us of host time, however little they compute.  `form`: whose roundings the launch reproduces, bit for bit --
    "foreach" (default): torch's multi-tensor implementation, i.e. what `AdamW(model.parameters())` of MODEL:205 runs on a GPU;
    "single": the single-tensor implementation (`foreach=False`).  The two differ in one operation (exp_avg_sq.sqrt() divided by
    sqrt(1 - beta2^step): a true division in the multi-tensor form, a product with the reciprocal in the other).  State (`step`,
    `exp_avg`, `exp_avg_sq`), param_groups, state_dict and the lr schedulers are torch's own; anything the launch does not cover (amsgrad,
    maximize, capturable, a CPU tensor, another dtype) goes to torch's step unchanged."""

    def __init__(self, params, *args, form: str = "foreach", **kwargs) -> None:
        if form not in ("foreach", "single"):
            raise ValueError(f"FlatAdamW form {form!r}: 'foreach' or 'single'")
        self.form = form
        # what torch runs where the launch does not apply, and what a checkpoint's param_groups say: the multi-tensor form is torch's
        # own choice on a GPU when `foreach` is None (the reference's `AdamW(model.parameters())` stores None), the other is False
        kwargs.setdefault("foreach", None if form == "foreach" else False)
        super().__init__(params, *args, **kwargs)

    def _fast(self, group) -> bool:
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("fused") or group.get("differentiable"):
            return False
        if torch.is_tensor(group["lr"]) or torch.is_tensor(group["betas"][0]) or torch.is_tensor(group["betas"][1]):
            return False
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32 and g.is_contiguous()
                    and not g.is_sparse and g.device == p.device):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._fast(g) for g in self.param_groups):
            return super().step(closure)
        from trajsde_amd import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:                                          # as torch's _init_group
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                step = float(state["step"])
                bias1, bias2 = 1 - beta1 ** step, 1 - beta2 ** step
                with torch.cuda.device(p.device):
                    divide = self.form == "foreach"
                    _lib.check(L.trajsde_adamw_step(p.data_ptr(), p.grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                                    p.numel(), 1 - lr * wd, 1 - beta1, beta2, 1 - beta2,
                                                    bias2 ** 0.5 if divide else 1.0 / (bias2 ** 0.5), int(divide), eps, -(lr / bias1),
                                                    torch.cuda.current_stream().cuda_stream), "trajsde_adamw_step")
        return loss


class FlatTraining:
    """One flat parameter tensor, one flat gradient tensor.  The parameters the losses reach are re-pointed at slices of a
    single buffer, and the model's optimizer (AdamW, MODEL:204-207) runs over that ONE tensor: the same update element by
    element as over ~150 tensors (AdamW is element-wise), one launch sequence instead of a multi-tensor one, and ~0.2 ms of
    host time per step instead of ~2.5 ms of per-tensor bookkeeping -- which is what a slow host otherwise exposes once the
    GPU side of a training step is down to ~13 ms.  Parameters the losses do not reach keep `.grad = None` and are not in the
    optimizer, exactly what torch's AdamW does with them (it skips parameters without a gradient, weight decay included).
    The optimizer state holds one tensor; checkpoints store it cut back into the per-parameter layout of the reference's
    `AdamW(self.parameters())` (optimizer_state_dict / load_optimizer_state_dict), so either side resumes the other's."""

    def __init__(self, model) -> None:
        params = [p for p in model.params_with_gradient() if p.requires_grad]
        ref = params[0]
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(ref.dtype).contiguous()
        self.flat_param = torch.nn.Parameter(flat)
        off = 0
        for p in params:
            p.data = self.flat_param.data[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.params = params
        self._named = list(model.named_parameters())
        self._all_params = [p for _, p in self._named]       # the order of `AdamW(self.parameters())` (MODEL:205)
        self._index = {id(p): i for i, p in enumerate(self._all_params)}
        self.grads = FlatGrads(params)                       # p.grad: slices of one gradient buffer, in the same order
        self.grads.names = {id(p): n for n, p in self._named}
        self.flat_param.grad = self.grads.flat
        model._grad_sink = self.grads                        # the path loss hands its gradients over in a few launches (accumulate)
        self._stages = [m for m in model.modules() if hasattr(m, "touch")]
        # The optimizer over the ONE tensor.  torch's multi-tensor ("foreach", the default on a GPU) and `fused=True` forms cut one
        # tensor into 64 K-element chunks -- ten workgroups for this model: nine launches of 25-55 us each, 0.33 ms of a 10 ms step (kernel
        # trace, round 4) -- and its single-tensor form is nine chip-wide launches of ~5 us.  FlatAdamW does the update in ONE launch and
        # ends on the very bits of the multi-tensor form, i.e. of `AdamW(model.parameters())` as the reference constructs it (MODEL:205;
        # tests/test_gpu_backward.py::test_flat_training_is_the_per_parameter_adamw_bit_for_bit, tests/test_gpu_step_launches.py).
        # `model.adamw_foreach = True`: torch's own multi-tensor implementation; `model.adamw_form = "single"`: the roundings of torch's
        # `foreach=False` (what this loop ran in round 4: a product with a reciprocal where the multi-tensor form divides).
        if getattr(model, "adamw_foreach", False) or not runtime.single_call_forms():
            self.optimizer = torch.optim.AdamW([self.flat_param], lr=model.lr, weight_decay=model.weight_decay,
                                               foreach=bool(getattr(model, "adamw_foreach", False)))
        else:
            self.optimizer = FlatAdamW([self.flat_param], lr=model.lr, weight_decay=model.weight_decay,
                                       form=str(getattr(model, "adamw_form", "foreach")))
        if hasattr(model, "scheduler_step"):
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=model.scheduler_step, gamma=model.scheduler_gamma)
        else:
            self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=model.T_max, eta_min=0.0)
        model.optimizer, model.scheduler = self.optimizer, self.scheduler      # `current_lr()` (the 'lr' log of MODEL:113) reads the live schedule

    def zero(self) -> None:
        self.grads.zero()

    def all_reduce_mean(self) -> None:
        self.grads.all_reduce_mean()

    def step(self) -> None:
        self.optimizer.step()
        for m in self._stages:                               # the slices' version counters did not move: tell the weight packers
            m.touch()

    # -- checkpoints: the optimizer state in the layout of `AdamW(model.parameters())` (MODEL:205) ----------------------
    def optimizer_state_dict(self) -> dict:
        """The state of the ONE flat tensor cut back into per-parameter entries, indexed like `model.parameters()` -- what
        the reference's `configure_optimizers` (and a Lightning checkpoint of it) holds.  Parameters the losses do not reach
        have no entry, as in torch (AdamW creates state on a parameter's first gradient)."""
        sd = self.optimizer.state_dict()
        group = dict(sd["param_groups"][0])
        group["params"] = list(range(len(self._all_params)))
        state = {}
        flat_state = sd["state"].get(0)
        if flat_state is not None:
            off = 0
            for p in self.params:
                n = p.numel()
                state[self._index[id(p)]] = {
                    "step": flat_state["step"].clone() if torch.is_tensor(flat_state["step"]) else flat_state["step"],
                    "exp_avg": flat_state["exp_avg"][off:off + n].view_as(p).clone(),
                    "exp_avg_sq": flat_state["exp_avg_sq"][off:off + n].view_as(p).clone()}
                off += n
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """accepts the per-parameter layout above (ours, the reference's, Lightning's) and the one-tensor layout this loop
        wrote before it had a converter (one parameter in the group whose state is as long as the flat tensor)"""
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError(f"optimizer state has {len(groups)} parameter groups; the recipe of MODEL:204-207 has one")
        n_flat = self.flat_param.numel()
        legacy = len(groups[0]["params"]) == 1 and len(self._all_params) != 1
        if legacy:
            st = sd["state"].get(groups[0]["params"][0])
            if st is not None and st["exp_avg"].numel() != n_flat:
                raise ValueError(f"flat optimizer state of {st['exp_avg'].numel()} elements does not fit this model's "
                                 f"{n_flat} optimised elements")
            flat_sd = {"state": {} if st is None else {0: st}, "param_groups": [dict(groups[0], params=[0])]}
            self.optimizer.load_state_dict(flat_sd)
            return
        if len(groups[0]["params"]) != len(self._all_params):
            raise ValueError(f"optimizer state covers {len(groups[0]['params'])} parameters, the model has "
                             f"{len(self._all_params)}: not a checkpoint of this architecture")
        ids = groups[0]["params"]                                             # saved id of the i-th parameter
        have = [ids[self._index[id(p)]] in sd["state"] for p in self.params]
        if not any(have):
            state = {}
        else:
            if not all(have):
                missing = [n for n, p in self._named if id(p) in {id(q) for q, h in zip(self.params, have) if not h}]
                raise ValueError(f"optimizer state lacks moments for optimised parameters: {missing[:4]} ...")
            ref = self.flat_param
            exp_avg = torch.empty(n_flat, device=ref.device, dtype=ref.dtype)
            exp_avg_sq = torch.empty_like(exp_avg)
            off, steps = 0, set()
            for p in self.params:
                st = sd["state"][ids[self._index[id(p)]]]
                if st["exp_avg"].numel() != p.numel():
                    raise ValueError("optimizer state shape mismatch for a parameter of %d elements" % p.numel())
                exp_avg[off:off + p.numel()] = st["exp_avg"].reshape(-1).to(ref.device, ref.dtype)
                exp_avg_sq[off:off + p.numel()] = st["exp_avg_sq"].reshape(-1).to(ref.device, ref.dtype)
                steps.add(float(st["step"]))
                off += p.numel()
            if len(steps) != 1:
                raise ValueError(f"parameters were stepped a different number of times ({sorted(steps)}): one flat tensor "
                                 "has one step counter")
            step = torch.tensor(steps.pop(), dtype=torch.float32)
            state = {0: {"step": step, "exp_avg": exp_avg, "exp_avg_sq": exp_avg_sq}}
        self.optimizer.load_state_dict({"state": state, "param_groups": [dict(groups[0], params=[0])]})


class PlainTraining:
    """the same handle over the model's own `configure_optimizers()` (per-parameter optimizer, flat gradient bucket): for
    modules that define their optimizer themselves and not the (lr, weight_decay, T_max) recipe FlatTraining rebuilds"""

    def __init__(self, model) -> None:
        (self.optimizer,), (self.scheduler,) = model.configure_optimizers()
        model.optimizer, model.scheduler = self.optimizer, self.scheduler
        self.grads = FlatGrads(model.params_with_gradient())
        self.grads.names = {id(p): n for n, p in model.named_parameters()}
        model._grad_sink = self.grads

    def zero(self) -> None:
        self.grads.zero()

    def all_reduce_mean(self) -> None:
        self.grads.all_reduce_mean()

    def step(self) -> None:
        self.optimizer.step()

    def optimizer_state_dict(self) -> dict:
        return self.optimizer.state_dict()

    def load_optimizer_state_dict(self, sd: dict) -> None:
        self.optimizer.load_state_dict(sd)


def rank_is_zero() -> bool:
    d = torch.distributed
    return not (d.is_available() and d.is_initialized()) or d.get_rank() == 0


RANK_SEED_STRIDE = 1_000_003          # noise seeds of rank r: base + step + r * stride (distinct streams per rank)


def assert_equal_step_counts(batches) -> None:
    """One blocking gradient all-reduce per step means every rank must run the same number of steps: check it before the
    first one instead of hanging in RCCL after the last.  Needs `len(batches)` (SceneLoader and synthetic_batches have it)."""
    import torch.distributed as dist
    if not hasattr(batches, "__len__"):
        raise ValueError("multi-rank training needs a sized batch source (len()) so that step counts can be checked up front")
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(batches), -len(batches)], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    hi, lo = int(n[0]), -int(n[1])
    if hi != lo:
        raise RuntimeError(f"ranks disagree on the number of steps per epoch ({lo}..{hi}): pad the scene list to a multiple "
                           "of the world size (SceneLoader(even=True), synthetic_batches(even=True)) or drop the tail")


def save_checkpoint(path: str, model, optimizer, scheduler, epoch: int, step: int) -> None:
    """A Lightning-shaped checkpoint (`state_dict` at the top level like the reference's `ModelCheckpoint` files, so
    either side loads the other's weights with `load_state_dict`; `optimizer_states[0]` per parameter in the order of
    `AdamW(self.parameters())`, MODEL:205), plus what the loop needs to resume.  `optimizer`: a FlatTraining /
    PlainTraining handle (its per-parameter view is stored) or a plain torch optimizer."""
    tmp = path + ".tmp"
    opt_state = optimizer.optimizer_state_dict() if hasattr(optimizer, "optimizer_state_dict") else optimizer.state_dict()
    torch.save({"state_dict": model.state_dict(), "optimizer_states": [opt_state],
                "lr_schedulers": [scheduler.state_dict()], "epoch": epoch, "global_step": step}, tmp)
    os.replace(tmp, path)


def train(model, batches_per_epoch, epochs: int, seed: int = 0, log=None, ckpt_path: Optional[str] = None,
          resume: Optional[str] = None) -> list:
    """trainer.fit(...) spelled out (train.py:60-66): per batch zero -> training_step -> backward -> gradient
    all-reduce -> AdamW step; scheduler step per epoch.  `batches_per_epoch(epoch)` yields this rank's batches.
    `ckpt_path`: rank 0 writes a checkpoint after every epoch; `resume`: continue from such a file (weights, AdamW
    moments, schedule, epoch and step counters -- the noise seeds continue where they stopped, so a resumed run
    retraces the uninterrupted one).  Returns the per-step loss values of this rank."""
    model.train()
    first_epoch, step = 0, 0
    state = torch.load(resume, map_location=model.device) if resume is not None else None
    if state is not None:
        model.load_state_dict(state["state_dict"])
    recipe = all(hasattr(model, a) for a in ("lr", "weight_decay")) and (hasattr(model, "T_max") or hasattr(model, "scheduler_step"))
    flat = FlatTraining(model) if recipe else PlainTraining(model)
    optimizer, scheduler = flat.optimizer, flat.scheduler
    if state is not None:
        flat.load_optimizer_state_dict(state["optimizer_states"][0])
        scheduler.load_state_dict(state["lr_schedulers"][0])
        first_epoch, step = int(state["epoch"]) + 1, int(state["global_step"])
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    # two-slice gradient all-reduce, the decoder + aggregator slice under the encoder backward (FlatGrads.early_reduce); this loop
    # never scales the loss, which is what the early slice assumes.  `model.overlap_grad_allreduce = False` keeps the one-piece form
    # Early or one-piece is decided HERE, once, from the parameter list (FlatGrads.early_plan) -- never per step from the gradients --
    # so every rank issues the same sequence of collectives.
    flat.grads.early_enabled = (dist_on and bool(getattr(model, "overlap_grad_allreduce", True))
                                and flat.grads.early_plan({id(p): n for n, p in model.named_parameters()}))
    if rank_is_zero() and log:
        opt = getattr(flat, "optimizer", None)
        if isinstance(opt, FlatAdamW):
            form = ("one launch over the flat tensor, the bits of torch's multi-tensor AdamW = the reference's AdamW(model.parameters())"
                    if opt.form == "foreach" else "one launch over the flat tensor, the bits of torch's single-tensor AdamW (foreach=False)")
        else:
            form = "torch's multi-tensor AdamW (foreach)" if getattr(model, "adamw_foreach", False) else "torch's / the model's own optimizer"
        print(f"[trajsde_amd.driver] AdamW form: {form}; gradient all-reduce: "
              f"{'two slices, decoder+aggregator early' if flat.grads.early_enabled else ('one piece' if dist_on else 'none (one rank)')}",
              file=sys.stderr)
    rank = torch.distributed.get_rank() if dist_on else 0
    rank0 = rank == 0
    history = []
    # The loop is pipelined by one step on the host side: batch i + 1 is fetched, moved to the device, rotated and run through the
    # graph stage on a side stream right after step i has been enqueued (runtime.prefetch_graph: the step's one host synchronisation
    # then waits for a few small kernels instead of draining the whole step), and the loss of step i is read -- a synchronisation
    # too -- only after step i + 1 has been enqueued.  Values, order of the log calls and checkpoints are those of the plain loop.
    pipelined = bool(getattr(model, "pipeline_training", True)) and hasattr(model, "prefetch_graph") and model.device.type == "cuda"
    side = runtime.side_stream(model.device) if pipelined else None

    def noise_of(step_):     # every rank draws its own Philox streams (rank folded into the key); rank 0's are those of a single-GPU run
        return NoiseSpec(seed=seed + step_ + RANK_SEED_STRIDE * rank)

    def fetch(it, step_):
        if not pipelined:
            return next(it, None)
        main = torch.cuda.current_stream(model.device)        # the stream the steps run on: what is prepared is handed to it
        with torch.cuda.stream(side):
            nxt = next(it, None)
            if nxt is not None:
                model.prefetch_graph(nxt, noise_of(step_), main_stream=main)
        return nxt

    def settle(pending):
        if pending is not None:
            loss_t, epoch_, i_, parts = pending
            history.append(float(loss_t))
            if log:
                log(epoch_, i_, history[-1], parts)

    for epoch in range(first_epoch, epochs):
        batches = batches_per_epoch(epoch)
        if dist_on:
            assert_equal_step_counts(batches)
        it = iter(batches)
        batch, i, pending = fetch(it, step), 0, None
        while batch is not None:
            flat.zero()
            loss = model.training_step(batch, i, noise=noise_of(step))
            loss.backward()
            flat.all_reduce_mean()
            flat.step()
            mine = (loss.detach(), epoch, i, dict(getattr(model, "last_losses", None) or {}))
            step += 1
            batch = fetch(it, step)
            settle(pending)
            pending = mine
            i += 1
        settle(pending)
        scheduler.step()
        if hasattr(model, "check_range"):
            model.check_range()
        if ckpt_path is not None and rank0:
            save_checkpoint(ckpt_path, model, flat, scheduler, epoch, step)
    return history


class synthetic_batches:
    """`n` synthetic scene-batches of BASELINE configuration `name`, sharded round-robin over the ranks.  `even` (the
    training loop): the batch list is padded by wrapping around to a multiple of `world`, so every rank runs the same
    number of steps (the DistributedSampler rule Lightning applies for the reference) -- 5 batches on 2 ranks are 3 steps
    each, batch 0 twice.  Evaluation passes even=False: no per-step collective, no batch counted twice."""

    def __init__(self, name: str, n: int, device, rank: int = 0, world: int = 1, even: bool = False):
        self.spec, self.device = CONFIGS[name], device
        order = list(range(n))
        if even and world > 1 and n % world:
            pad = world - n % world
            order += (order * (pad // max(n, 1) + 1))[:pad]
        self.ids = order[rank::world]

    def __len__(self) -> int:
        return len(self.ids)

    def __iter__(self):
        for i in self.ids:
            yield synth(**dict(self.spec["synth"], seed=self.spec["synth"]["seed"] + 97 * i)).to(self.device)


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


def datamodule_train_batches(cfg: dict, device, rank: int = 0, world: int = 1, nu_dir=None, argo_dir=None):
    """train.py:56-66: the YAML's data module, `setup('fit')`, its train loader.  Returns `per_epoch(epoch)` for `train()`:
    the loader ITSELF, re-seeded for the epoch -- it is sized (`len()` honours the even padding over the ranks), which
    `assert_equal_step_counts` needs; `iter(loader)` is a generator and is not."""
    dm_cfg = cfg["datamodule_specific"]
    kwargs = dict(dm_cfg["kwargs"], rank=rank, world_size=world, device=device)
    if nu_dir:
        kwargs["nu_dir"] = nu_dir
    if argo_dir:
        kwargs["Argo_dir"] = argo_dir
    dm = resolve_class(dm_cfg["file_path"], dm_cfg["module_name"])(**kwargs)
    dm.setup("fit")
    loader = dm.train_dataloader()

    def per_epoch(epoch):
        loader.set_epoch(epoch)
        return loader
    return per_epoch


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--synthetic", default="config1", choices=sorted(CONFIGS))
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--ood", action="store_true")
    ap.add_argument("--train", action="store_true", help="run the optimisation loop instead of the evaluation loop")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--save", default=None, help="--train: checkpoint written after every epoch")
    ap.add_argument("--resume", default=None, help="--train: continue from a checkpoint written by --save")
    ap.add_argument("--data", action="store_true", help="evaluate on the YAML's data module instead of synthetic batches")
    ap.add_argument("--nu_dir", default=None)
    ap.add_argument("--argo_dir", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        from trajsde_amd.shard import pin_rank_to_cores
        pin_rank_to_cores(local_rank)                        # this rank's share of the host cores, torch's pool capped to it
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
        cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T)
        if "max_fut_t" in cfg["decoder"]["kwargs"]:
            cfg["decoder"]["kwargs"]["max_fut_t"] = spec["max_fut_t"]
        for m in cfg["metric_args"]:
            m["end_idcs"] = [T - 1, T - 1]
    if args.ood:
        cfg["model_specific"]["kwargs"]["ood"] = True                                # test.py:45-46
    dev = torch.device("cuda", local_rank)
    model = build_model(cfg, args.ckpt, dev, init_seed=0 if args.ckpt is None else None)
    if args.train:
        if args.data:
            per_epoch = datamodule_train_batches(cfg, dev, rank, world, args.nu_dir, args.argo_dir)
        else:
            def per_epoch(epoch):
                return synthetic_batches(args.synthetic, args.batches, dev, rank, world, even=True)
        hist = train(model, per_epoch, args.epochs, ckpt_path=args.save, resume=args.resume,
                     log=(lambda e, i, l, parts: print(f"epoch {e} step {i} loss {l:.5f}")) if rank == 0 else None)
        if rank == 0:
            print(json.dumps({"steps": len(hist), "first_loss": hist[0], "last_loss": hist[-1]}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.data:
        batches = datamodule_batches(cfg, dev, rank, world, args.nu_dir, args.argo_dir)
    else:
        batches = synthetic_batches(args.synthetic, args.batches, dev, rank, world)
    res = evaluate(model, batches)                                                    # metric states all-reduce in compute(): every rank
    if rank == 0:
        text = json.dumps(res)
        print(text)                                                                   # first: nothing below can lose the numbers
        if args.out:
            with open(args.out, "w") as f:
                f.write(text)
        if args.ckpt is not None:
            # test_epoch_end's side file (MODEL:150-165), the result JSON next to the tested checkpoint, from the values already
            # reduced above -- no second compute() on one rank (its all-reduces would have no partner on the others)
            model.result_ckpt_path = args.ckpt
            try:
                model.write_results(dict(res))
            except OSError as e:
                print(f"warning: result file next to the checkpoint not written ({e})", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
