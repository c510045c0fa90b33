"""Scene sharding for multi-GPU runs: scenes are independent units, so ranks own disjoint
scene sets and the data path needs no collective.  What must be global is the *noise*: Philox counters are keyed
by global row ids so that a scene's sample paths do not depend on the rank or the batch it lands in."""
from typing import List, Sequence

import torch

from trajsde_amd.runtime import NoiseSpec


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """round-robin scene ids of `rank` (balances scene sizes that drift along a dataset)"""
    return list(range(rank, n_scenes, world))


def global_noise_spec(seed: int, scene_ids: Sequence[int], actors_per_scene: Sequence[int], num_modes: int,
                      device="cpu") -> NoiseSpec:
    """Row ids for the scenes `scene_ids` of a dataset whose scene s has actors_per_scene[s] actors (one target
    agent per scene).  Global layout: actor rows first (offset by the actors of all earlier scenes), then one fake
    row per scene, decoder rows k*N_total + actor."""
    counts = torch.as_tensor(list(actors_per_scene), dtype=torch.int64)
    offs = torch.cumsum(counts, 0) - counts
    n_total, s_total = int(counts.sum()), len(counts)
    actor_ids = torch.cat([torch.arange(int(offs[s]), int(offs[s] + counts[s])) for s in scene_ids])
    fake_ids = torch.as_tensor(list(scene_ids), dtype=torch.int64)
    enc = torch.cat([actor_ids, n_total + fake_ids])
    dec = torch.cat([k * n_total + actor_ids for k in range(num_modes)])
    assert n_total * max(num_modes, 1) + s_total < 2 ** 31
    i32 = lambda t: t.to(torch.int32).to(device).contiguous()
    return NoiseSpec(seed=seed, fake_row_ids=i32(fake_ids), enc_row_ids=i32(enc), dec_row_ids=i32(dec))
