"""Stand-in for torchmetrics==0.10.3 -- TEST INFRASTRUCTURE (Metric.add_state only; metrics/*.py:17)."""
import copy

import torch


class Metric(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = copy.deepcopy(default)
        setattr(self, name, copy.deepcopy(default))

    def reset(self):
        for k, v in self._defaults.items():
            setattr(self, k, copy.deepcopy(v))
