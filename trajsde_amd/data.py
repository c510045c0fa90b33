"""Batch container with the access pattern of the reference's TemporalData
(models/utils/util.py:21-75): attribute and item get/set, `in`, `.keys`, `num_nodes`, plus collate
with the index-offset rules of PyG's DataLoader and the `__inc__` overrides at util.py:67-75.
"""
from typing import Dict, Iterable, List

import torch

# keys whose values index actors (offset by the running actor count when scenes are concatenated)
_ACTOR_INDEX_KEYS = ("edge_index", "agent_index", "av_index")
_PER_SCENE_SCALARS = ("source", "theta")


class TemporalData:
    def __init__(self, **tensors):
        object.__setattr__(self, "_store", {})
        for k, v in tensors.items():
            self._store[k] = v

    def __getattr__(self, key):
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        if key == "num_nodes":
            return store["x"].size(0)
        if key == "y":
            return None
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        self._store[key] = value

    def __delitem__(self, key):
        del self._store[key]

    def __contains__(self, key):
        return key in self._store

    @property
    def keys(self) -> List[str]:
        return list(self._store.keys())

    def to(self, device, non_blocking: bool = False) -> "TemporalData":
        out = TemporalData()
        for k, v in self._store.items():
            out._store[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def as_dict(self) -> Dict[str, object]:
        return dict(self._store)


def collate(scenes: Iterable[TemporalData]) -> TemporalData:
    """Concatenate scenes into one batch (what PyG's collate does for TemporalData):
    per-actor tensors along dim 0, `*index*` keys offset by the running actor count, `lane_actor_index`
    offset by [lanes; actors] (util.py:67-69), per-scene scalars stacked, plus the `batch` vector."""
    scenes = list(scenes)
    out: Dict[str, object] = {}
    n_off, l_off = 0, 0
    acc: Dict[str, list] = {}
    batch = []
    for s_id, sc in enumerate(scenes):
        n = sc.num_nodes
        n_lanes = sc["lane_positions"].size(0) if "lane_positions" in sc else 0
        for k in sc.keys:
            v = sc[k]
            if k == "num_nodes":
                continue
            if not torch.is_tensor(v):
                acc.setdefault(k, []).append(v)
                continue
            if k in _ACTOR_INDEX_KEYS:
                v = v + n_off
            elif k == "lane_actor_index":
                v = v + torch.tensor([[l_off], [n_off]], dtype=v.dtype, device=v.device)
            elif k == "lane_edge_index":
                v = v + l_off
            acc.setdefault(k, []).append(v)
        batch.append(torch.full((n,), s_id, dtype=torch.long))
        n_off += n
        l_off += n_lanes
    for k, parts in acc.items():
        v0 = parts[0]
        if not torch.is_tensor(v0):
            # PyG turns per-scene Python numbers into one tensor (e.g. `source`), anything else into a list
            out[k] = torch.tensor(parts) if isinstance(v0, (bool, int, float)) else parts
        elif k in _PER_SCENE_SCALARS or v0.dim() == 0:
            out[k] = torch.stack([p.reshape(()) for p in parts])
        elif "index" in k and v0.dim() == 2:          # [2, E] edge lists
            out[k] = torch.cat(parts, dim=-1)
        elif "index" in k:                             # per-scene actor ids
            out[k] = torch.cat([p.reshape(-1) for p in parts])
        else:
            out[k] = torch.cat(parts, dim=0)
    out["batch"] = torch.cat(batch)
    out["num_nodes"] = n_off
    return TemporalData(**out)
