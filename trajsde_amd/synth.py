"""Synthetic scene generator pinned by SURVEY.md 8(d): synth(S, n, L, F, box, seed).

CPU fp32, one torch.Generator.  Field rules cite the reference's preprocessing:
displacement inputs (dataset/nuScenes/nuScenes_hivt.py:232-235), absolute-GT futures (:229-231),
bos rule (:225-226), all ordered intra-scene pairs (dataset/Argoverse/Argoverse_abs.py:201),
full lane x actor product with lane end - actor position vectors (Argoverse_abs.py:423-426),
nuScenes grid sparsity (dataset/nuScenes_Argoverse/nuScenes_Argoverse.py:91-103).
"""
import math
from typing import Optional

import torch

from .data import TemporalData

T_HIST = 21


def synth(S: int, n: int, L: int, F: int, box: float, seed: int, mixed_source: bool = False,
          source: int = 0, history_dropout: float = 0.0, nus_sparsity: bool = False,
          with_y: bool = True) -> TemporalData:
    g = torch.Generator().manual_seed(int(seed))
    N = S * n
    pos0 = torch.rand(N, 2, generator=g) * box
    vel = torch.randn(N, 2, generator=g) * 0.5
    tau = torch.arange(-(T_HIST - 1), F + 1, dtype=torch.float32)          # -20 .. F
    positions = pos0[:, None, :] + vel[:, None, :] * tau[None, :, None]      # [N, 21+F, 2]
    x = torch.zeros(N, T_HIST, 2)
    x[:, 1:] = positions[:, 1:T_HIST] - positions[:, :T_HIST - 1]
    y = positions[:, T_HIST:] - positions[:, T_HIST - 1:T_HIST]
    padding_mask = torch.zeros(N, T_HIST + F, dtype=torch.bool)
    rotate_angles = (torch.rand(N, generator=g) * 2 - 1) * math.pi

    if history_dropout > 0:
        first = (torch.rand(N, generator=g) < history_dropout).long() * \
            torch.randint(1, T_HIST - 1, (N,), generator=g)
        padding_mask[:, :T_HIST] = torch.arange(T_HIST)[None, :] < first[:, None]
    if nus_sparsity:
        keep_past = torch.zeros(T_HIST, dtype=torch.bool)
        keep_past[[0, 5, 10, 15, 20]] = True
        keep_fut = torch.zeros(F, dtype=torch.bool)
        keep_fut[torch.arange(4, F, 5)] = True
        padding_mask[:, :T_HIST] |= ~keep_past[None, :]
        padding_mask[:, T_HIST:] |= ~keep_fut[None, :]
    # the current step is always observed (agents are selected that way in preprocessing)
    padding_mask[:, T_HIST - 1] = False
    valid = ~padding_mask[:, :T_HIST]
    bos_mask = torch.zeros(N, T_HIST, dtype=torch.bool)
    bos_mask[:, 0] = valid[:, 0]
    bos_mask[:, 1:] = valid[:, 1:] & ~valid[:, :-1]
    x = torch.where(valid[:, :, None], x, torch.zeros(()))
    # a displacement needs both endpoints (nuScenes_hivt.py:232-235 zeroes the others)
    both = torch.zeros_like(valid)
    both[:, 1:] = valid[:, 1:] & valid[:, :-1]
    if not nus_sparsity:
        x = torch.where(both[:, :, None], x, torch.zeros(()))

    idx = torch.arange(n)
    src, dst = torch.meshgrid(idx, idx, indexing="ij")
    keep = src != dst
    pairs = torch.stack([src[keep], dst[keep]])                              # [2, n(n-1)]
    edge_index = torch.cat([pairs + s * n for s in range(S)], dim=1)

    lane_positions = torch.rand(S * L, 10, 2, generator=g) * box
    lane_paddings = torch.zeros(S * L, 10)
    li, ai = torch.meshgrid(torch.arange(L), torch.arange(n), indexing="ij")
    la = torch.stack([li.reshape(-1), ai.reshape(-1)])
    lane_actor_index = torch.cat([la + torch.tensor([[s * L], [s * n]]) for s in range(S)], dim=1)
    lane_actor_vectors = lane_positions[lane_actor_index[0], -1] - positions[lane_actor_index[1], T_HIST - 1]

    agent_index = torch.arange(S) * n
    batch = torch.arange(S).repeat_interleave(n)
    if mixed_source:
        src_vec = torch.arange(S) % 2
    else:
        src_vec = torch.full((S,), int(source), dtype=torch.long)
    d = dict(x=x, positions=positions, padding_mask=padding_mask, bos_mask=bos_mask,
             rotate_angles=rotate_angles, edge_index=edge_index, lane_positions=lane_positions,
             lane_paddings=lane_paddings, lane_actor_index=lane_actor_index,
             lane_actor_vectors=lane_actor_vectors, agent_index=agent_index, av_index=agent_index.clone(),
             batch=batch, source=src_vec, num_nodes=N)
    if with_y:
        d["y"] = y
    return TemporalData(**d)


# BASELINE.json configs -> generator arguments
CONFIGS = {
    "config1": dict(synth=dict(S=1, n=32, L=40, F=5, box=100.0, seed=1, nus_sparsity=True, source=0),
                    num_modes=1, future_steps=5, max_fut_t=0.5),
    "config2": dict(synth=dict(S=64, n=128, L=64, F=20, box=200.0, seed=2, mixed_source=True),
                    num_modes=6, future_steps=20, max_fut_t=2.0),
    "config3": dict(synth=dict(S=32, n=48, L=150, F=30, box=150.0, seed=3, source=1),
                    num_modes=6, future_steps=30, max_fut_t=3.0),
    # BASELINE configs[3]: mixed nuScenes+Argoverse training as shipped (K=10, T=60, 128 scenes per GPU: CFG:9-22,106)
    "config4": dict(synth=dict(S=128, n=48, L=150, F=60, box=150.0, seed=4, mixed_source=True),
                    num_modes=10, future_steps=60, max_fut_t=6.0),
    # the workload BASELINE.json's metric is quoted on (K=6, 20 SDE steps, ~256 agents), batched to fill one GPU: 32 scenes
    # of 256 agents (N = 8192 actors, ~7.6 M surviving (t, edge) pairs per forward)
    "metric256": dict(synth=dict(S=32, n=256, L=64, F=20, box=200.0, seed=2, mixed_source=True),
                      num_modes=6, future_steps=20, max_fut_t=2.0),
    "scene256": dict(synth=dict(S=1, n=256, L=64, F=20, box=200.0, seed=2),
                     num_modes=6, future_steps=20, max_fut_t=2.0),
    "config5": dict(synth=dict(S=8, n=1024, L=256, F=50, box=600.0, seed=5, mixed_source=True),
                    num_modes=20, future_steps=50, max_fut_t=5.0),
}
