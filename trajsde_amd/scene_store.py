"""Flat on-disk scene format.

The reference keeps one pickled PyG `TemporalData` per scene (`torch.load(self.processed_paths[idx])`,
dataset/nuScenes_Argoverse/nuScenes_Argoverse.py:141); reading those needs torch_geometric and costs one
small file open per scene.  Here a *shard* is one `.safetensors` file holding many scenes back to back:

    <key>        every scene's tensor concatenated along the key's concat axis
                 (last axis for [2, E] `*index*` keys, axis 0 otherwise); 0-dim values are stacked;
                 scenes of one shard share dtypes and the other axes, i.e. come from one source
    <key>.ptr    int64 [S+1] extents along that axis (absent for stacked scalars)
    <key>.has    uint8 [S], only when some scenes lack the key (e.g. `category` exists for nuScenes only)
    metadata     {"format": "trajsde-scenes-v1", "num_scenes": S, "strings": json {key: [S strings]}}

so a shard opens with one mmap, a scene is S slices, and a whole split (a few GB) can simply stay resident in
the 288 GB of HBM (`SceneShard(..., device="cuda")`).  `convert_scene` is duck-typed on the PyG object
(`.keys` + item access), so the one-off conversion runs wherever the pickles can be unpickled and nothing
here imports torch_geometric.
"""
import json
import os
from typing import Dict, Iterable, List, Mapping, Optional, Sequence

import torch

FORMAT = "trajsde-scenes-v1"


def _concat_axis(key: str, v: torch.Tensor) -> int:
    return -1 if ("index" in key and v.dim() == 2) else 0


def convert_scene(obj) -> Dict[str, object]:
    """PyG-style scene object (or mapping) -> plain dict of tensors / scalars / strings, `None`s dropped."""
    if isinstance(obj, Mapping):
        items = obj.items()
    else:
        keys = obj.keys() if callable(obj.keys) else obj.keys
        items = ((k, obj[k]) for k in keys)
    out = {}
    for k, v in items:
        if v is None or k in ("num_nodes", "batch", "ptr"):
            continue
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().contiguous()
        elif isinstance(v, bool):
            out[k] = torch.tensor(v)
        elif isinstance(v, int):
            out[k] = torch.tensor(v, dtype=torch.int64)
        elif isinstance(v, float):
            out[k] = torch.tensor(v, dtype=torch.float32)
        elif isinstance(v, str):
            out[k] = v
        else:
            raise TypeError(f"scene field {k!r}: unsupported type {type(v).__name__}")
    return out


def write_shard(path: str, scenes: Sequence[Mapping[str, object]]) -> None:
    from safetensors.torch import save_file
    scenes = [convert_scene(s) for s in scenes]
    n = len(scenes)
    keys: List[str] = []
    for s in scenes:
        keys += [k for k in s if k not in keys]
    tensors: Dict[str, torch.Tensor] = {}
    strings: Dict[str, List[str]] = {}
    for k in keys:
        have = [k in s for s in scenes]
        proto = next(s[k] for s in scenes if k in s)
        if isinstance(proto, str):
            strings[k] = [str(s.get(k, "")) for s in scenes]
            continue
        if not all(have):
            tensors[k + ".has"] = torch.tensor(have, dtype=torch.uint8)
        if proto.dim() == 0:
            fill = torch.zeros((), dtype=proto.dtype)
            tensors[k] = torch.stack([s.get(k, fill).to(proto.dtype) for s in scenes])
            continue
        ax = _concat_axis(k, proto)
        empty = proto.narrow(ax, 0, 0)
        parts = [s.get(k, empty) for s in scenes]
        rest = lambda t: [d for i, d in enumerate(t.shape) if i != (ax % t.dim())]
        for p in parts:
            if p.dtype != proto.dtype or p.dim() != proto.dim() or rest(p) != rest(proto):
                raise ValueError(f"scene field {k!r}: dtype/shape differs between scenes "
                                 "(one shard holds scenes of one source)")
        ext = torch.tensor([0] + [p.size(ax) for p in parts], dtype=torch.int64)
        tensors[k] = torch.cat(parts, dim=ax).contiguous()
        tensors[k + ".ptr"] = torch.cumsum(ext, 0)
    meta = {"format": FORMAT, "num_scenes": str(n), "strings": json.dumps(strings)}
    tmp = path + ".tmp"
    save_file(tensors, tmp, metadata=meta)
    os.replace(tmp, path)


class SceneShard:
    """One shard opened for random access.  `device=None` keeps the mmapped host tensors; a device keeps the whole
    shard resident there and `scene()` returns device views (no per-scene H2D copy)."""

    def __init__(self, path: str, device: Optional[str] = None):
        from safetensors import safe_open
        self.path = path
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            if meta.get("format") != FORMAT:
                raise ValueError(f"{path}: not a {FORMAT} shard")
            self.num_scenes = int(meta["num_scenes"])
            self._strings = json.loads(meta.get("strings", "{}"))
            raw = {k: f.get_tensor(k) for k in f.keys()}
        self._ptr = {k[:-4]: v.tolist() for k, v in raw.items() if k.endswith(".ptr")}
        self._has = {k[:-4]: v.bool().tolist() for k, v in raw.items() if k.endswith(".has")}
        self._data = {k: (v.to(device) if device else v) for k, v in raw.items()
                      if not (k.endswith(".ptr") or k.endswith(".has"))}

    def __len__(self) -> int:
        return self.num_scenes

    def extents(self, key: str) -> List[int]:
        """rows of the ragged field `key` per scene, from the shard's pointer table alone (no scene is materialised); zeros
        for scenes that do not carry the field"""
        ptr = self._ptr.get(key)
        if ptr is None:
            return [0] * self.num_scenes
        return [ptr[i + 1] - ptr[i] for i in range(self.num_scenes)]

    @property
    def keys(self) -> List[str]:
        return list(self._data) + list(self._strings)

    def scene(self, i: int) -> Dict[str, object]:
        if not 0 <= i < self.num_scenes:
            raise IndexError(i)
        out: Dict[str, object] = {}
        for k, v in self._data.items():
            if k in self._has and not self._has[k][i]:
                continue
            if k in self._ptr:
                lo, hi = self._ptr[k][i], self._ptr[k][i + 1]
                out[k] = v.narrow(_concat_axis(k, v), lo, hi - lo)
            else:
                out[k] = v[i]
        for k, v in self._strings.items():
            out[k] = v[i]
        return out


class SceneStore:
    """Several shards behind one index space (shard order = argument order)."""

    def __init__(self, paths: Iterable[str], device: Optional[str] = None):
        self.shards = [SceneShard(p, device) for p in paths]
        self._starts = [0]
        for s in self.shards:
            self._starts.append(self._starts[-1] + len(s))

    def __len__(self) -> int:
        return self._starts[-1]

    def extents(self, key: str) -> List[int]:
        return [e for s in self.shards for e in s.extents(key)]

    def scene(self, i: int) -> Dict[str, object]:
        if not 0 <= i < len(self):
            raise IndexError(i)
        lo, hi = 0, len(self.shards)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            lo, hi = (mid, hi) if self._starts[mid] <= i else (lo, mid)
        return self.shards[lo].scene(i - self._starts[lo])


def convert_pickles(paths: Sequence[str], out_path: str, scenes_per_call: int = 1 << 30) -> int:
    """One-off converter: torch.load each reference `.pt` (needs the environment that can unpickle them, i.e. with
    torch_geometric and the reference's `models.utils.util` importable) and write one shard."""
    scenes = []
    for p in paths[:scenes_per_call]:
        scenes.append(convert_scene(torch.load(p, weights_only=False)))
    write_shard(out_path, scenes)
    return len(scenes)


if __name__ == "__main__":
    import argparse
    import glob
    ap = argparse.ArgumentParser(description="convert reference per-scene .pt pickles into one flat shard")
    ap.add_argument("src", help="directory of processed .pt files")
    ap.add_argument("dst", help="output .safetensors shard")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.src, "*.pt")))
    print(f"{convert_pickles(files, a.dst)} scenes -> {a.dst}")
