"""Stand-in for torchsde._core.base_solver: BaseSDESolver, interp, adaptive_stepping.

`integrate` restates the fixed-step loop of torchsde 0.2.5; the reference vendors a
stock-signature twin of it at models/utils/sdeint.py:400-445, which is what this follows.
"""
import types

import torch
from torch import nn

from ..settings import NOISE_TYPES


def _linear_interp(t0, y0, t1, y1, t):
    assert t0 <= t <= t1, f"Incorrect time order for linear interpolation: t0={t0}, t={t}, t1={t1}."
    y = (t1 - t) / (t1 - t0) * y0 + (t - t0) / (t1 - t0) * y1
    return y


interp = types.SimpleNamespace(linear_interp=_linear_interp)


def _no_adaptive(*a, **k):
    raise NotImplementedError("adaptive stepping is off on the hot path (CFG:45 adaptive: false)")


adaptive_stepping = types.SimpleNamespace(compute_error=_no_adaptive, update_step_size=_no_adaptive)


class BaseSDESolver(nn.Module):
    def __init__(self, sde, bm, dt, adaptive, rtol, atol, dt_min, options, **kwargs):
        super().__init__(**kwargs)
        if sde.sde_type != self.sde_type:
            raise ValueError(f"SDE is of type {sde.sde_type} but solver is for type {self.sde_type}")
        if sde.noise_type not in self.noise_types:
            raise ValueError(f"SDE has noise type {sde.noise_type} but solver only supports {self.noise_types}")
        if bm.levy_area_approximation not in self.levy_area_approximations:
            raise ValueError("Brownian levy_area_approximation unsupported by solver")
        if sde.noise_type == NOISE_TYPES.scalar and torch.Size(bm.shape[1:]).numel() != 1:
            raise ValueError("The Brownian motion for scalar SDEs must of dimension 1.")
        self.sde = sde
        self.bm = bm
        self.dt = dt
        self.adaptive = adaptive
        self.rtol = rtol
        self.atol = atol
        self.dt_min = dt_min
        self.options = options

    def init_extra_solver_state(self, t0, y0):
        return ()

    def step(self, t0, t1, y0, extra0):
        raise NotImplementedError

    def integrate(self, y0, ts, extra0):
        if self.adaptive:
            _no_adaptive()
        step_size = self.dt
        prev_t = curr_t = ts[0]
        prev_y = curr_y = y0
        curr_extra = extra0
        ys = [y0]
        for out_t in ts[1:]:
            while curr_t < out_t:
                next_t = min(curr_t + step_size, ts[-1])
                prev_t, prev_y = curr_t, curr_y
                curr_y, curr_extra = self.step(curr_t, next_t, curr_y, curr_extra)
                curr_t = next_t
            ys.append(_linear_interp(t0=prev_t, y0=prev_y, t1=curr_t, y1=curr_y, t=out_t))
        return torch.stack(ys, dim=0), curr_extra
