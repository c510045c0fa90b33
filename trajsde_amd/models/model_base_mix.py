"""PredictionModel -- glue module of the vanilla HiVT variant of the path (reference models/model_base_mix.py:22-209):
same YAML registry, `forward(data) -> dict`, the same in-place side effects on `data` (rotate_mat, rotated y) and the
same validation / test step bookkeeping as the SDE model; the stages are the HIP-backed LocalEncoder,
GlobalInteractor and MLPDecoder.  Deterministic (no SDE noise).  Inference only: the backward kernels cover the SDE
configuration (8 heads), so `training_step` raises here instead of differentiating anything elsewhere.
"""
from copy import deepcopy
from typing import Optional

import torch
import torch.nn as nn

from trajsde_amd import runtime
from trajsde_amd.models.model_base_mix_sde import resolve_class


class PredictionModel(nn.Module):
    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.hparams = deepcopy({k: v for k, v in kwargs.items() if k != "init_seed"})
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

    def forward(self, data):
        """models/model_base_mix.py:74-92"""
        if not self.rotate:
            raise NotImplementedError("rotate=False is not built (shipped config: rotate: true)")
        rotate_mat, y_rot = runtime.rotate_inputs(data)
        if y_rot is not None:
            data.y = y_rot
        data["rotate_mat"] = rotate_mat
        local_embed = self.encoder(data=data)
        global_embed = self.aggregator(data=data, local_embed=local_embed)
        return self.decoder(data=data, local_embed=local_embed, global_embed=global_embed)

    def training_step(self, data, batch_idx):
        raise NotImplementedError("the vanilla HiVT variant is inference-only here; train the SDE configuration")

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
        if data.y is not None:
            y_hat, y, mask, source = self._agent_eval_tensors(data, output)
            for metric in self.metrics_vl:
                metric.update(y_hat.detach(), y.detach(), mask.detach(), source.detach())
        return output

    def metric_results(self):
        return {n: float(m.compute()) for n, m in zip(self.metric_names, self.metrics_vl)}
