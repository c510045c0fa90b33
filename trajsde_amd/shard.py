"""Scene sharding for multi-GPU runs: scenes are independent units, so ranks own disjoint
scene sets and the data path needs no collective.  What must be global is the *noise*: Philox counters are keyed
by global row ids so that a scene's sample paths do not depend on the rank or the batch it lands in."""
import os
from typing import List, Optional, Sequence

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


def core_share(cores: Sequence[int], local_rank: int, local_world: int) -> List[int]:
    """the host cores of rank `local_rank` of `local_world` ranks on one node: a contiguous share of the sorted core list (so that
    a rank's threads stay on neighbouring cores / one NUMA domain where the ids are laid out that way), every core handed to exactly
    one rank, shares differing by at most one core; with fewer cores than ranks the ranks share cores round-robin"""
    cores = sorted(cores)
    if local_world <= 1 or not cores:
        return list(cores)
    if len(cores) < local_world:
        return [cores[local_rank % len(cores)]]
    base, extra = divmod(len(cores), local_world)
    lo = local_rank * base + min(local_rank, extra)
    return list(cores[lo:lo + base + (1 if local_rank < extra else 0)])


def pin_rank_to_cores(local_rank: Optional[int] = None, local_world: Optional[int] = None, max_threads: int = 16) -> dict:
    """One process per GPU shares the node's host cores with its siblings: the training step needs ~2 ms of host time per ~7 ms of
    GPU time (launch enqueue, the next batch's collate), and eight ranks that each let torch spawn a thread per core, all free to
    migrate, turn that into contention the moment the node is full.  Call BEFORE the first GPU call of the process: the rank is
    confined to its share of the cores the process may use (os.sched_setaffinity) and torch's intra-op pool is capped to it
    (at most `max_threads`: the host side here is many small ops, more threads are slower -- bench.py cpu_baseline).
    Single-rank runs are left alone.  Returns what it did (for logs / the bench line)."""
    lr = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else int(local_world)
    info = {"local_rank": lr, "local_world": lw, "pinned": False}
    if lw <= 1:
        return info
    try:
        avail = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return info
    share = core_share(avail, lr, lw)
    try:
        os.sched_setaffinity(0, share)
        info["pinned"] = True
    except OSError:
        pass
    nt = max(1, min(len(share), int(max_threads)))
    torch.set_num_threads(nt)
    info.update(cores=len(share), first_core=share[0], last_core=share[-1], torch_threads=nt)
    return info
