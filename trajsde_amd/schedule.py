"""Exact float32 replay of torchsde's fixed-step time bookkeeping.

The reference's solvers advance `curr_t` as a 0-dim float32 tensor plus a Python-float step and clip
it against ts[-1] (models/utils/sdeint.py:340-384; stock twin :400-445), then interpolate linearly
onto the requested grid.  With dt = 0.1 the accumulated float32 error makes the solver take an extra
micro-step for T in {30, 50, 60}; that step still injects noise, so both the CPU oracle and the HIP
kernels consume THIS table rather than assuming "T steps of 0.1".

All arithmetic below is done with torch CPU float32 0-dim tensors, i.e. the very operations the
reference executes on the host.
"""
import math
from dataclasses import dataclass
from typing import List

import numpy as np
import torch


@dataclass
class EulerSchedule:
    t0: np.ndarray        # [n_euler] f32  start time of each Euler step (argument of f and g)
    dt: np.ndarray        # [n_euler] f32  t1 - t0 as the float32 tensor subtraction the step uses
    sqrt_h: np.ndarray    # [n_euler] f32  scale of the Brownian increment, sqrt(float(t1)-float(t0))
    sin_t0: np.ndarray    # [n_euler] f32  torch.sin(float32(t0))  (FFunc/GFunc time features)
    cos_t0: np.ndarray    # [n_euler] f32
    out_step: np.ndarray  # [n_out] i32    emit output o after this many Euler steps have completed
    out_w0: np.ndarray    # [n_out] f32    weight on the state before the last step
    out_w1: np.ndarray    # [n_out] f32    weight on the state after the last step

    @property
    def n_euler(self) -> int:
        return int(self.t0.shape[0])

    @property
    def n_out(self) -> int:
        return int(self.out_step.shape[0])

    def step_table(self) -> np.ndarray:
        """[n_euler, 8] f32 rows (t0, dt, sqrt_h, sin, cos, 0, 0, 0): the layout the kernels read."""
        tab = np.zeros((self.n_euler, 8), dtype=np.float32)
        tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3], tab[:, 4] = self.t0, self.dt, self.sqrt_h, self.sin_t0, self.cos_t0
        return tab

    def out_table(self) -> np.ndarray:
        """[n_out, 4] f32 rows (out_step as float, w0, w1, 0)."""
        tab = np.zeros((self.n_out, 4), dtype=np.float32)
        tab[:, 0], tab[:, 1], tab[:, 2] = self.out_step.astype(np.float32), self.out_w0, self.out_w1
        return tab


def replay_fixed_step(ts: torch.Tensor, dt: float) -> EulerSchedule:
    """Run the solver's while-loop on time values only and record every step and interpolation."""
    ts = ts.detach().to("cpu")
    assert ts.dtype == torch.float32 and ts.dim() == 1
    t0s: List[float] = []
    dts: List[float] = []
    sqh: List[float] = []
    out_step: List[int] = []
    w0s: List[float] = []
    w1s: List[float] = []
    prev_t = curr_t = ts[0]
    n_done = 0
    for out_t in ts[1:]:
        while curr_t < out_t:
            next_t = min(curr_t + dt, ts[-1])
            prev_t = curr_t
            t0s.append(float(curr_t))
            dts.append(float(next_t - curr_t))                       # float32 tensor subtraction
            sqh.append(math.sqrt(float(next_t) - float(curr_t)))     # BrownianInterval works on float()
            curr_t = next_t
            n_done += 1
        out_step.append(n_done)
        w0s.append(float((curr_t - out_t) / (curr_t - prev_t)))
        w1s.append(float((out_t - prev_t) / (curr_t - prev_t)))
    t0 = torch.tensor(t0s, dtype=torch.float32)
    return EulerSchedule(
        t0=t0.numpy().copy(), dt=np.asarray(dts, dtype=np.float32), sqrt_h=np.asarray(sqh, dtype=np.float32),
        sin_t0=torch.sin(t0).numpy().copy(), cos_t0=torch.cos(t0).numpy().copy(),
        out_step=np.asarray(out_step, dtype=np.int32), out_w0=np.asarray(w0s, dtype=np.float32),
        out_w1=np.asarray(w1s, dtype=np.float32))


def decoder_schedule(future_steps: int, max_fut_t: float, dt: float = 0.1) -> EulerSchedule:
    """ts_pred = linspace(0, max_fut_t, T+1) (dec_hivt_nusargo_sde.py:72), one sdeint call (:88)."""
    return replay_fixed_step(torch.linspace(0, max_fut_t, future_steps + 1), dt)


def encoder_schedule(historical_steps: int = 21, max_past_t: float = 2.0, dt: float = 0.1,
                     run_backwards: bool = True) -> EulerSchedule:
    """The 21 single-interval sdeint_dual calls of enc_hivt_nusargo_sde_sep2.py:128-179, concatenated.

    Entry i of the returned schedule belongs to loop iteration idx = i, which consumes the
    agent-agent embedding of history step t = 20 - i (run_backwards) and emits latent state i."""
    if not run_backwards:
        raise NotImplementedError("the shipped config runs backwards (CFG:40)")
    past = -1 * torch.linspace(-max_past_t, 0, historical_steps)
    prev_t, t_i = past[-1] - 0.01, past[-1]
    parts = []
    order = list(reversed(range(historical_steps)))
    for idx, t in enumerate(order):
        parts.append(replay_fixed_step(torch.tensor([prev_t, t_i]), dt))
        if idx + 1 < historical_steps:
            prev_t, t_i = past[t], past[t - 1]
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    steps_before = np.cumsum([0] + [p.n_euler for p in parts[:-1]]).astype(np.int32)
    return EulerSchedule(t0=cat("t0"), dt=cat("dt"), sqrt_h=cat("sqrt_h"), sin_t0=cat("sin_t0"), cos_t0=cat("cos_t0"),
                         out_step=np.concatenate([p.out_step + o for p, o in zip(parts, steps_before)]).astype(np.int32),
                         out_w0=cat("out_w0"), out_w1=cat("out_w1"))
