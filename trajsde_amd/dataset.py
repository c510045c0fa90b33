"""Mixed nuScenes + Argoverse dataset over flat scene shards, and the loader that feeds the hot path.

Mirrors the reference's `nuArgoDataset` (dataset/nuScenes_Argoverse/nuScenes_Argoverse.py:40-268, "MIXDS"
below) and `DataModuleNuArgoMix` (dataset/Datamodule_nuargo_mix.py:14-46): same constructor arguments, same
per-scene output of `get(idx)`, same split-name table -- but scenes come from `trajsde_amd.scene_store`
shards under `<nu_dir>/<split>/*.safetensors` / `<Argo_dir>/<split>/*.safetensors` instead of PyG pickles, and
batches are collated by `trajsde_amd.data.collate`.

What `get` does to a stored scene (MIXDS:140-232):
  * drops the lane-graph extras the model never reads, tags `source` (0 nuScenes, 1 Argoverse), `seq_id` -> str;
  * nuScenes displacements are per 0.5 s, Argoverse per 0.1 s: nuScenes `x` (and `y`) are divided by 5;
  * unless `is_gtabs`, `y` becomes per-step displacements (first step relative to the origin);
  * actors whose `category` is not of interest are padded (the last 60 stored slots; all 17 for nuScenes);
  * `type: grid`: every tensor is scattered onto the common 10 Hz grid of 21 past / 60 future slots
    (nuScenes fills every 5th slot, Argoverse the last 20 past and first 30 future slots);
  * split `train`: random x / y flips (`random_flip`), drawing from Python's `random` like the reference.
"""
import glob
import os
import random
from typing import Dict, Iterator, List, Optional, Sequence

import torch

from trajsde_amd.data import TemporalData, collate
from trajsde_amd.scene_store import SceneStore

# MIXDS:34-37
SPLIT_NAME = {"nuScenes": {"train": "train", "val": "val", "test": "val", "mini_train": "mini_train",
                           "mini_val": "mini_val"},
              "Argoverse": {"train": "train", "val": "train", "test": "test_obs", "sample": "forecasting_sample"}}
DATA_SOURCE = {0: "nuScenes", 1: "Argoverse"}
CATEGORY_INTEREST = (0, 1, 2, 3, 4, 5, 7, 8)

_UNUSED_KEYS = ("traffic_controls", "turn_directions", "is_intersections", "city", "lane_rotate_angles",
                "lane_edge_index", "lane_edge_type", "lane_edge_index2_succ", "lane_edge_index2_pred",
                "lane_edge_index2_neigh")
MAX_PAST, MAX_FUT = 21, 60
_FLIP_VECTOR_KEYS = ("x", "y", "positions", "lane_positions", "lane_vectors", "lane_actor_vectors")


def _grid_slots(source: int):
    """(past slots [21] bool, future slots [60] bool) a source occupies on the common 10 Hz grid (MIXDS:89-109):
    nuScenes samples at 2 Hz (t = -20, -15, ..., 0 | 5, 10, ..., 60), Argoverse at 10 Hz (t = -19..0 | 1..30)."""
    past = torch.zeros(MAX_PAST, dtype=torch.bool)
    fut = torch.zeros(MAX_FUT, dtype=torch.bool)
    if source == 0:
        past[0::5] = True
        fut[4::5] = True
    elif source == 1:
        past[1:] = True
        fut[:30] = True
    else:
        raise KeyError("source should be nuScenes(0) or Argoverse(1)")
    return past, fut


def _div5(v: torch.Tensor) -> torch.Tensor:
    """v / 5 with the host's correctly rounded float32 result also for HBM-resident stores: the device's float32
    division may differ in the last bit, the float64 quotient rounded to float32 cannot (a quotient by 5 is never
    within 2^-27 of a rounding boundary)"""
    return (v.double() / 5.0).to(v.dtype) if v.is_cuda else v / 5


def _shards(root: str, sub: str) -> List[str]:
    files = sorted(glob.glob(os.path.join(root, sub, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no scene shards under {os.path.join(root, sub)} "
                                "(convert the processed .pt files with `python -m trajsde_amd.scene_store`)")
    return files


class nuArgoDataset(torch.utils.data.Dataset):
    def __init__(self, split: str, nu_root, Argo_root, nu_dir, Argo_dir, spec_args=None,
                 device: Optional[str] = None) -> None:
        self._split = split
        self.nu_root, self.Argo_root, self.nu_dir, self.Argo_dir = nu_root, Argo_root, nu_dir, Argo_dir
        self.nus = self.Argo = True
        self.type, self.is_gtabs, self.random_flip = "grid", True, False
        for k, v in (spec_args or {}).items():
            setattr(self, k, v)
        if self.type != "grid":
            raise NotImplementedError("only the grid layout is implemented (as in the reference, MIXDS:198)")
        self.nu_directory = SPLIT_NAME["nuScenes"][split]
        self.Argo_directory = SPLIT_NAME["Argoverse"][split]
        paths, self._sources = [], []
        self.stores: List[SceneStore] = []
        for use, root, sub, src in ((self.nus, nu_dir, self.nu_directory, 0),
                                    (self.Argo, Argo_dir, self.Argo_directory, 1)):
            if not use:
                continue
            st = SceneStore(_shards(root, sub), device)
            self.stores.append(st)
            self._sources += [src] * len(st)
        self._starts = [0]
        for st in self.stores:
            self._starts.append(self._starts[-1] + len(st))
        self.max_past, self.max_fut = MAX_PAST, MAX_FUT
        self._slots = {s: _grid_slots(s) for s in (0, 1)}

    def __len__(self) -> int:
        return self._starts[-1]

    len = __len__

    def __getitem__(self, idx: int) -> TemporalData:
        return self.get(idx)

    def scene_costs(self) -> List[float]:
        """what a scene costs a training step, from the shards' pointer tables alone: n^2 (the agent-agent candidate pairs of every
        history step and the fully connected global graph both grow with it: SURVEY.md 8(e)) + its lane segments.  Read by
        SceneLoader(balance="cost")."""
        out: List[float] = []
        for st in self.stores:
            n, lanes = st.extents("x"), st.extents("lane_vectors")
            out += [float(a) * float(a) + float(b) for a, b in zip(n, lanes)]
        return out

    def _raw(self, idx: int) -> Dict[str, object]:
        which = 0 if idx < self._starts[1] else 1
        return self.stores[which].scene(idx - self._starts[which])

    def get(self, idx: int) -> TemporalData:
        source = self._sources[idx]
        sc = {k: v for k, v in self._raw(idx).items() if k not in _UNUSED_KEYS}
        sc["source"] = source
        sc["seq_id"] = str(sc["seq_id"].item() if torch.is_tensor(sc["seq_id"]) else sc["seq_id"])
        for k in ("agent_index", "av_index"):
            sc[k] = torch.as_tensor(sc[k]).to(torch.long).reshape(())
        x, y = sc["x"], sc.get("y")
        x = _div5(x) if source == 0 else x            # nuScenes steps span 5 grid slots
        if not self.is_gtabs:
            y = torch.diff(y, dim=1, prepend=torch.zeros_like(y[:, :1]))
            y = _div5(y) if source == 0 else y
        pad = sc["padding_mask"].clone()
        if "category" in sc:
            cat = sc.pop("category").float()
            keep = torch.isin(cat, torch.tensor(CATEGORY_INTEREST, dtype=cat.dtype, device=cat.device))
            # the reference slices the *stored* mask with the grid's future length (MIXDS:177): on the 17-slot
            # nuScenes mask that is the whole row, i.e. such actors are padded at every step, past included
            pad[~keep, -MAX_FUT:] = True
        past, fut = (m.to(x.device) for m in self._slots[source])
        both = torch.cat((past, fut))
        n = x.size(0)
        gx = x.new_zeros(n, MAX_PAST, x.size(-1))
        gx[:, past] = x
        gbos = torch.zeros(n, MAX_PAST, dtype=torch.bool, device=x.device)
        gbos[:, past] = sc["bos_mask"]
        gpad = torch.ones(n, MAX_PAST + MAX_FUT, dtype=torch.bool, device=x.device)
        gpad[:, both] = pad
        pos = sc["positions"]
        gpos = pos.new_zeros(n, MAX_PAST + MAX_FUT, pos.size(-1))
        gpos[:, both] = pos
        if y is not None:
            gy = y.new_zeros(n, MAX_FUT, y.size(-1))
            gy[:, fut] = y
            sc["y"] = gy
        sc.update(x=gx, bos_mask=gbos, padding_mask=gpad, positions=gpos)
        if self._split == "train":
            self.augment(sc)
        return TemporalData(**sc)

    def augment(self, sc: Dict[str, object]) -> Dict[str, object]:
        """MIXDS:234-263: two independent coin flips, mirror about the y axis then about the x axis."""
        if not self.random_flip:
            return sc
        for axis in (0, 1):
            if not random.choice([0, 1]):
                continue
            sign = torch.ones(2)
            sign[axis] = -1.0
            for k in _FLIP_VECTOR_KEYS:
                if sc.get(k) is not None:
                    sc[k] = sc[k] * sign.to(sc[k].device)
            for k in ("theta", "rotate_angles"):
                a = torch.as_tensor(sc[k])
                c, s = torch.cos(a) * sign[0], torch.sin(a) * sign[1]
                sc[k] = torch.atan2(s, c)
        return sc


class SceneLoader:
    """Batches of collated scenes on `device`.  Ranks take disjoint round-robin scene sets (SURVEY.md 8(e):
    scenes are independent, no data-path collective); host batches are staged through pinned memory and copied
    on a side stream one batch ahead of the consumer."""

    def __init__(self, dataset, batch_size: int, shuffle: bool = False, device: Optional[str] = None,
                 rank: int = 0, world_size: int = 1, seed: int = 0, drop_last: bool = False, even: bool = True,
                 balance: str = "round_robin"):
        """`even` (default): every rank gets the same number of scenes -- the order is padded by wrapping around to a
        multiple of world_size, like torch's DistributedSampler that Lightning installs for the reference (train.py:54) --
        so that every rank runs the same number of steps and the per-step gradient all-reduce cannot be left waiting.
        Evaluation loaders pass even=False: no per-step collective there, and no scene is counted twice in the metrics.

        `balance`: "round_robin" (default: rank r takes every world_size-th scene of the order -- the DistributedSampler rule the
        reference trains under) or "cost": the SAME scenes per step, dealt so that the ranks' work is level.  A step's cost grows
        with the sum of n^2 over its scenes, and every step ends in a gradient all-reduce that waits for the heaviest rank; with
        "cost" the world_size * batch_size scenes of a step are dealt greedily, heaviest first, to the rank with the least work so
        far that still has a free slot (LPT with equal scene counts).  Same steps per rank, same scenes per step as round-robin:
        the averaged gradient of a step is over the same scenes, only who computes which changes."""
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.device = torch.device(device) if device is not None else None
        self.rank, self.world_size, self.seed, self.drop_last, self.even = rank, world_size, seed, drop_last, even
        if balance not in ("round_robin", "cost"):
            raise ValueError(f"SceneLoader: balance must be 'round_robin' or 'cost', not {balance!r}")
        self.balance = balance
        self._costs: Optional[List[float]] = None
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def scene_costs(self) -> List[float]:
        """cost of every scene of the dataset (cached): the dataset's own `scene_costs()` where it has one (nuArgoDataset reads
        the shard index), else n^2 + lane segments from the scenes themselves, loaded once"""
        if self._costs is None:
            if hasattr(self.dataset, "scene_costs"):
                self._costs = [float(c) for c in self.dataset.scene_costs()]
            else:
                def one(sc):
                    n = int(sc["x"].shape[0])
                    lv = sc["lane_vectors"] if "lane_vectors" in sc else None
                    return float(n * n + (int(lv.shape[0]) if lv is not None else 0))
                self._costs = [one(self.dataset[i]) for i in range(len(self.dataset))]
            if len(self._costs) != len(self.dataset):
                raise ValueError("scene_costs(): one cost per scene expected")
        return self._costs

    def _order(self) -> List[int]:
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        if self.even and self.world_size > 1 and n % self.world_size:
            pad = self.world_size - n % self.world_size
            order += (order * (pad // max(n, 1) + 1))[:pad]
        return order

    def scene_ids(self) -> List[int]:
        order, W = self._order(), self.world_size
        if self.balance != "cost" or W == 1:
            return order[self.rank::W]
        cost, mine = self.scene_costs(), []
        per_step = W * self.batch_size
        for lo in range(0, len(order), per_step):
            group = order[lo:lo + per_step]
            cap = [len(group) // W + (1 if r < len(group) % W else 0) for r in range(W)]     # what round-robin would hand out
            load, taken = [0.0] * W, [[] for _ in range(W)]
            # heaviest first; ties by position, so that every rank computes the same deal
            for pos in sorted(range(len(group)), key=lambda p: (-cost[group[p]], p)):
                r = min((r for r in range(W) if len(taken[r]) < cap[r]), key=lambda r: (load[r], r))
                taken[r].append(pos)
                load[r] += cost[group[pos]]
            mine += [group[p] for p in sorted(taken[self.rank])]
        return mine

    def step_costs(self) -> List[float]:
        """this rank's cost of every step of the epoch (diagnostics, tests)"""
        ids, cost = self.scene_ids(), self.scene_costs()
        return [sum(cost[i] for i in ids[lo:lo + self.batch_size]) for lo in range(0, len(ids), self.batch_size)]

    def __len__(self) -> int:
        n = len(self.scene_ids())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _host_batches(self) -> Iterator[TemporalData]:
        ids = self.scene_ids()
        for lo in range(0, len(ids), self.batch_size):
            chunk = ids[lo:lo + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                return
            yield collate(self.dataset[i] for i in chunk)

    def __iter__(self) -> Iterator[TemporalData]:
        if self.device is None or self.device.type != "cuda":
            for b in self._host_batches():
                yield b if self.device is None else b.to(self.device)
            return
        copy_stream = torch.cuda.Stream(self.device)
        pending = None
        for b in self._host_batches():
            with torch.cuda.stream(copy_stream):
                staged = TemporalData(**{k: (v.pin_memory() if torch.is_tensor(v) and not v.is_cuda else v)
                                         for k, v in b.as_dict().items()})
                dev = staged.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            if pending is not None:
                yield self._hand_over(*pending)
            pending = (dev, ev, staged)
        if pending is not None:
            yield self._hand_over(*pending)

    def _hand_over(self, done, copied, staged):
        """the batch was allocated and filled on the copy stream: make the consumer's stream wait for the copy and tell the
        caching allocator that the consumer's stream uses these blocks (so they are not recycled under its kernels)"""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(copied)
        for v in done.as_dict().values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return done


class DataModuleNuArgoMix:
    """Same YAML kwargs as the reference datamodule (Datamodule_nuargo_mix.py:16-46); `dataset_file_path` /
    `dataset_module_name` are accepted and resolved through the same SourceFileLoader registry."""

    def __init__(self, dataset_file_path=None, dataset_module_name="nuArgoDataset", **kwargs) -> None:
        self.train_batch_size = self.val_batch_size = 32
        self.shuffle = True
        self.rank, self.world_size, self.device = 0, 1, None
        for k, v in kwargs.items():
            setattr(self, k, v)
        if dataset_file_path:
            from importlib.machinery import SourceFileLoader
            here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            path = dataset_file_path if os.path.isfile(dataset_file_path) else os.path.join(here, dataset_file_path)
            mod = SourceFileLoader(dataset_module_name, path).load_module(dataset_module_name)
            self.dataset_module = getattr(mod, dataset_module_name)
        else:
            self.dataset_module = nuArgoDataset

    def setup(self, stage: Optional[str] = None) -> None:
        extra = {"device": self.resident_device} if getattr(self, "resident_device", None) else {}
        mk = lambda split, args: self.dataset_module(split, self.nu_root, self.Argo_root, self.nu_dir,
                                                     self.Argo_dir, spec_args=args, **extra)
        # the reference builds all three regardless of `stage`; with stage given only what that stage reads is opened
        if stage in (None, "fit"):
            self.train_dataset = mk("train", self.tr_dataset_args)
        if stage in (None, "fit", "validate"):
            self.val_dataset = mk("val", self.val_dataset_args)
        if stage in (None, "test"):
            self.test_dataset = mk("val", self.test_dataset_args)     # Datamodule_nuargo_mix.py:31

    def _loader(self, ds, bs, shuffle, even):
        # `balance` (YAML kwarg of the data module, default "round_robin"): SceneLoader's dealing rule for the train loader
        return SceneLoader(ds, bs, shuffle=shuffle, device=self.device, rank=self.rank, world_size=self.world_size, even=even,
                           balance=getattr(self, "balance", "round_robin") if even else "round_robin")

    def train_dataloader(self):
        return self._loader(self.train_dataset, self.train_batch_size, self.shuffle, True)     # equal step counts per rank

    def val_dataloader(self):
        return self._loader(self.val_dataset, self.val_batch_size, False, False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, self.val_batch_size, False, False)
