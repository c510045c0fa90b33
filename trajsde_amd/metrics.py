"""ADE_T / FDE_T / MR_T with the reference's formulas (metrics/ade_t.py:39-66, metrics/fde_t.py:39-57,
metrics/mr_t.py:41-70) as device-side reductions: predictions stay on the GPU, the running sum/count
are device scalars and are only read back by `compute()` (the reference copies every batch to the host,
model_base_mix_sde.py:131,148).  `sum`/`count` are summed across ranks like torchmetrics'
dist_reduce_fx='sum' states when torch.distributed is initialised.
"""
from typing import Sequence

import torch
import torch.distributed as dist


class _MinOverModes:
    def __init__(self, dataset: str, end_idcs: Sequence[int], sources: Sequence[int] = (0, 1), **_) -> None:
        self.dataset, self.end_idcs, self.target_sources = dataset, list(end_idcs), list(sources)
        self.reset()

    def reset(self) -> None:
        self.sum = torch.zeros((), dtype=torch.float64)
        self.count = torch.zeros((), dtype=torch.float64)

    def _acc(self, s: torch.Tensor, c: torch.Tensor) -> None:
        self.sum = self.sum.to(s.device) + s.double()
        self.count = self.count.to(s.device) + c.double()

    def _end_index(self, source: torch.Tensor, horizon: int) -> torch.Tensor:
        """per-agent last index: end_idcs[0] for sources[0] (nuScenes, 59), end_idcs[1] otherwise (29).
        The reference builds this with repeat_interleave over the *counts*, i.e. it assumes agents are
        ordered by source; indexing by each agent's own source gives the same answer there and the right
        one when sources interleave."""
        e = torch.where(source == self.target_sources[0], self.end_idcs[0], self.end_idcs[1])
        return e.clamp(max=horizon - 1)

    def compute(self) -> torch.Tensor:
        s, c = self.sum.clone(), self.count.clone()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(s)
            dist.all_reduce(c)
        return (s / c).float().cpu()

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        new.__dict__.update({k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})
        return new


def _masked_l2(pred, target, reg_mask):
    l2 = torch.norm(pred - target.unsqueeze(0), p=2, dim=-1)          # [K, A, T]
    return l2 * reg_mask.unsqueeze(0)


class ADE_T(_MinOverModes):
    def update(self, pred, target, reg_mask, source) -> None:
        l2 = _masked_l2(pred, target, reg_mask)
        any_valid = reg_mask.any(-1)
        n_valid = reg_mask.sum(-1).clamp(min=1)
        ade = l2.sum(-1) / n_valid.unsqueeze(0)                           # [K, A]
        if self.dataset == "nuScenes":
            best = ade.argmin(0)
        elif self.dataset == "Argoverse":
            # the reference picks the mode on the MASKED distances (ade_t.py:45,58: l2[:, ~reg_mask] = 0 comes first), so an
            # agent whose end step is invalid sees fde = 0 for every mode and keeps mode 0
            end = self._end_index(source, l2.size(-1))
            fde = l2.gather(2, end.view(1, -1, 1).expand(l2.size(0), -1, 1)).squeeze(-1)
            best = fde.argmin(0)
        else:
            raise NotImplementedError("other dataset is not implemented")
        ade_best = ade.gather(0, best.unsqueeze(0)).squeeze(0)
        self._acc((ade_best * any_valid).sum(), any_valid.sum())


class FDE_T(_MinOverModes):
    def update(self, pred, target, reg_mask, source) -> None:
        K, A, T, _ = pred.shape
        end = self._end_index(source, T)
        ar = torch.arange(A, device=pred.device)
        l2 = torch.norm(pred[:, ar, end] - target[ar, end].unsqueeze(0), p=2, dim=-1)   # [K, A]
        valid = reg_mask[ar, end]
        self._acc((l2.min(0).values * valid).sum(), valid.sum())


class MR_T(_MinOverModes):
    def __init__(self, dataset, end_idcs, sources=(0, 1), miss_threshold: float = 2.0, **kw) -> None:
        super().__init__(dataset, end_idcs, sources)
        self.miss_threshold = miss_threshold

    def update(self, pred, target, reg_mask, source) -> None:
        if self.dataset == "nuScenes":
            l2 = _masked_l2(pred, target, reg_mask)
            valid = reg_mask.any(-1)
            missed = l2.max(-1).values.min(0).values > self.miss_threshold
        elif self.dataset == "Argoverse":
            K, A, T, _ = pred.shape
            end = self._end_index(source, T)
            ar = torch.arange(A, device=pred.device)
            l2 = torch.norm(pred[:, ar, end] - target[ar, end].unsqueeze(0), p=2, dim=-1)
            valid = reg_mask[ar, end]
            missed = l2.min(0).values > self.miss_threshold
        else:
            raise NotImplementedError("other dataset is not implemented")
        self._acc((missed & valid).sum(), valid.sum())
