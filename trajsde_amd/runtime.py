"""Host-side runtime of the stage modules: weight packing, workspaces, noise selection and the ctypes
calls into libtrajsde_hip.so.  PyTorch is used for device memory and streams only."""
import collections.abc
import ctypes as C
import itertools
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from trajsde_amd import _lib
from trajsde_amd.schedule import EulerSchedule, decoder_schedule, encoder_schedule

D = 64


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.TrajsdeError(f"{what} must live on the GPU: the hot path has no CPU implementation "
                                "(oracle/ is test infrastructure, not a fallback)")


@dataclass
class NoiseSpec:
    """Where the path's randomness comes from (SURVEY.md App. F draw order).

    seed            in-kernel Philox4x32-7 key; host twin: trajsde_amd/philox.py
    z_fake/z_enc/z_dec   injected standard normals [A,21,2] / [21,Nt,64] / [n_euler,K*N,64] (parity tests)
    *_row_ids       int32 global row ids for the Philox counter (resharding-invariant streams)
    """
    seed: int = 0
    seed_dev: Optional[torch.Tensor] = None  # one uint64 (as int64) on the device: the key is read THERE when the kernels run
                                             # (`seed` ignored) -- what lets a captured graph of the forward draw fresh noise
    dropout_seed: Optional[int] = None       # key of the train-mode dropout masks (csrc/dropout.hpp); default: `seed`
    z_fake: Optional[torch.Tensor] = None
    z_enc: Optional[torch.Tensor] = None
    z_dec: Optional[torch.Tensor] = None
    fake_row_ids: Optional[torch.Tensor] = None
    enc_row_ids: Optional[torch.Tensor] = None
    dec_row_ids: Optional[torch.Tensor] = None

    @staticmethod
    def resolve(noise: Optional["NoiseSpec"]) -> "NoiseSpec":
        if noise is not None:
            return noise
        return NoiseSpec(seed=int(torch.randint(0, 2 ** 62, (1,)).item()))   # fresh stream per forward, like the reference

    def c_dropout(self, module) -> Optional[_lib.Dropout]:
        """the stage's `dropout` kwarg as the C struct while the module is in train mode, else None (eval: no dropout)"""
        p = float(getattr(module, "dropout", 0.0) or 0.0)
        if not getattr(module, "training", False) or p <= 0.0:
            return None
        if self.dropout_seed is None and self.seed_dev is not None:
            raise _lib.TrajsdeError("train-mode dropout with a device-resident Philox key (seed_dev): the mask key is a host "
                                    "value -- pass NoiseSpec(dropout_seed=...)")
        ds = self.seed if self.dropout_seed is None else self.dropout_seed
        return _lib.Dropout(C.c_float(p), C.c_uint64(int(ds) & 0xFFFFFFFFFFFFFFFF))

    def c_noise(self, z: Optional[torch.Tensor], row_ids: Optional[torch.Tensor]) -> _lib.Noise:
        if z is not None:
            assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous()
        if row_ids is not None:
            assert row_ids.is_cuda and row_ids.dtype == torch.int32 and row_ids.is_contiguous()
        if self.seed_dev is not None:
            assert self.seed_dev.is_cuda and self.seed_dev.dtype == torch.int64 and self.seed_dev.numel() == 1
        return _lib.Noise(C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), _ptr(z), _ptr(row_ids), _ptr(self.seed_dev))


class GraphedForward:
    """The inference forward of one batch captured in a HIP graph (torch.cuda.CUDAGraph) and replayed.

    Possible because the forward is sync-free (trajsde_graph_prepare_async: list lengths stay on the device, buffers and grids are
    sized from bounds) and the Philox key can live in device memory (NoiseSpec.seed_dev): a replay re-runs graph preparation,
    encoder, global interactor and decoder on whatever the batch tensors hold NOW, with the key written before the replay.
    The batch tensors must keep their shapes and addresses (update them in place); outputs are the captured tensors.

        gf = GraphedForward(model, batch)          # warm-up + capture
        out = gf(seed=123)                         # one graph launch; out["loc"] etc. are overwritten by the next call
    """

    def __init__(self, model, data, warmup: int = 2) -> None:
        x = data["x"]
        _require_gpu(x, "data['x']")
        if not sync_free():
            raise _lib.TrajsdeError("GraphedForward needs the sync-free forward (TRAJSDE_SYNC_FREE, default kernel forms)")
        if model.training:
            raise _lib.TrajsdeError("GraphedForward captures the inference forward: call model.eval() first")
        self.model, self.data = model, data
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=x.device)
        self.noise = NoiseSpec(seed=0, seed_dev=self.seed_dev)
        self._y = data.y                                                    # the forward REBINDS data.y to the rotated copy (MODEL:83-84):
        side = torch.cuda.Stream(device=x.device)                          # the captured kernels keep reading this tensor
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):                                  # weight images packed, LDS limits raised, pools warm
                data.y = self._y
                model(data, noise=self.noise)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        for mod in model.modules():                                          # every weight image is packed and the device is idle:
            rt = getattr(mod, "_rt", None)                                   # drop the pack-done events, so that blob() does not
            if isinstance(rt, StageRuntime):                                 # query one inside the capture (not capturable)
                rt.packs_complete()
        if GraphContext.KEY in data:                                         # the capture must contain the graph stage itself
            del data[GraphContext.KEY]
        self.graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=x.device)                             # the capture stream and its side stream (the relative-pose
        _side_stream(x.device, cap)                                          # prefetch forks onto it) exist before the capture starts
        with torch.cuda.graph(self.graph, stream=cap), torch.no_grad():
            data.y = self._y
            self.out = model(data, noise=self.noise)

    def __call__(self, seed: int):
        self.seed_dev.fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)
        self.graph.replay()
        return self.out


def set_state_storage(kind: str) -> str:
    """"fp32" (default) or "bf16": how the [rows][64] activations that stay inside a stage are stored between its kernels
    (include/trajsde_hip.h trajsde_state_storage; BASELINE configs[4] "bf16 hidden state").  Inference only.  Returns the
    previous setting."""
    if kind not in ("fp32", "bf16"):
        raise ValueError("state storage is 'fp32' or 'bf16'")
    return "bf16" if _lib.lib().trajsde_state_storage(1 if kind == "bf16" else 0) else "fp32"


# Off by default: measured on the metric workload (32 x 256 agents) the one-stream forward gains 0.6 % (2.781 against 2.798 ms)
# and the three-stream headline loses 1.2 % -- the recurrence's four waves per CU hold the whole register file of their SIMDs
# (512 registers each), so the embedding kernel only finds room on the CUs the recurrence leaves empty (HISTORY.md section 5).
_OVERLAP_REL = os.environ.get("TRAJSDE_OVERLAP_REL", "0") != "0"
_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def _side_stream(dev, cur) -> "torch.cuda.Stream":
    """one side stream per (device, stream the forward runs on): forwards dealt over several streams keep their own"""
    key = (str(dev), int(cur.cuda_stream))
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class RelPrefetch:
    """the aggregator's relative-pose rows on their way on a side stream (StageRuntime.prefetch_rel_embed)"""

    def __init__(self, gc, ws, ws_bytes, done, dev) -> None:
        self.gc, self.ws, self.ws_bytes, self.done, self.dev = gc, ws, ws_bytes, done, dev
        self.joined = False

    def join(self) -> None:
        """the current stream waits for the side stream's kernel (also what ends a captured fork)"""
        if not self.joined:
            torch.cuda.current_stream(self.dev).wait_event(self.done)
            self.joined = True


_SYNC_FREE = os.environ.get("TRAJSDE_SYNC_FREE", "1") != "0"
SYNC_FREE_MAX_BYTES = 4 << 30
# ... and for the training loop's prefetched graphs: TWO batches are alive at once there and the bound-sized lists keep their size
# after make_exact() (at 32 x 256 agents ~2 GB a batch where the exact build needs ~160 MB), so the prefetch path gets half the cap
# -- at most 2 x 2 GB of the 288 GB reserved by it (the shipped training shapes need 0.3-1.0 GB) -- and beyond it takes the
# synchronising form: its one host read then waits on the side stream, which is what the form was before round 5
PREFETCH_SYNC_FREE_MAX_BYTES = 2 << 30


def set_sync_free(on: bool) -> bool:
    """Sync-free inference forward (default on): the graph stage leaves the compacted lists' lengths on the device
    (trajsde_graph_prepare_async) and sizes buffers and grids from bounds, so a forward never waits on the GPU.  Entry points
    that need the lengths on the host (training, backward, captures) make the graph exact themselves.  Returns the previous
    setting."""
    global _SYNC_FREE
    prev, _SYNC_FREE = _SYNC_FREE, bool(on)
    return prev


def sync_free() -> bool:
    return _SYNC_FREE and bool(_lib.lib().trajsde_sync_free_supported())


def state_storage() -> str:
    L = _lib.lib()
    prev = L.trajsde_state_storage(0)
    L.trajsde_state_storage(prev)
    return "bf16" if prev else "fp32"


def rotate_inputs(data) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """MODEL:75-85 on the GPU: rotate_mat [N,2,2] and y @ rotate_mat."""
    ang = data["rotate_angles"]
    _require_gpu(ang, "data['rotate_angles']")
    N = ang.shape[0]
    y = data.y
    rot = torch.empty(N, 2, 2, device=ang.device, dtype=torch.float32)
    y_rot = torch.empty_like(y) if y is not None else None
    F = y.shape[1] if y is not None else 0
    _lib.check(_lib.lib().trajsde_rotate(_ptr(ang.contiguous()), N, _ptr(y.contiguous()) if y is not None else None, F,
                                         _ptr(rot), _ptr(y_rot), _stream()), "trajsde_rotate")
    return rot, y_rot


ROTATED_KEY = "_trajsde_rotated"          # set by prefetch_graph: MODEL:76-85 has been applied to this batch already, once
_PREFETCH_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def side_stream(device) -> "torch.cuda.Stream":
    """the stream next batches are prepared on while the current step runs (one per device)"""
    key = str(device)
    if key not in _PREFETCH_STREAMS:
        _PREFETCH_STREAMS[key] = torch.cuda.Stream(device=device)
    return _PREFETCH_STREAMS[key]


_COLLECTIVE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def collective_stream(device) -> "torch.cuda.Stream":
    """the stream the early slice of the gradient all-reduce runs on (driver.FlatGrads.early_reduce), one per device and NOT the
    prefetch stream: the collective must neither queue behind the next batch's graph stage nor hold that stage back"""
    key = str(device)
    if key not in _COLLECTIVE_STREAMS:
        _COLLECTIVE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COLLECTIVE_STREAMS[key]


def prefetch_graph(data, radius: float, H: int, noise: "NoiseSpec", fake_agents: bool = True, main_stream=None) -> None:
    """Rotation (MODEL:76-85) and graph stage of a batch the training loop will use NEXT, on the side stream -- call it under
    `torch.cuda.stream(side_stream(dev))`, with the batch's tensors produced on that stream as well (driver.train moves the next
    batch to the device there).  The graph stage holds the one host synchronisation of a training step (the list lengths size the
    tapes): taken here it waits for a few small kernels on the side stream while the main stream still holds milliseconds of the
    current step, and the next `training_step` starts enqueueing at once -- taken inside the step the GPU idles from the last
    optimizer kernel until the host has woken up and launched again.  Everything made here is handed to the main stream (event +
    allocator bookkeeping); `training_step` finds the rotation marked done and the graph cached on the batch."""
    dev = data["x"].device
    side = torch.cuda.current_stream(dev)
    main = main_stream if main_stream is not None else torch.cuda.default_stream(dev)      # the stream the training step runs on
    if side == main:
        raise _lib.TrajsdeError("prefetch_graph must run under torch.cuda.stream(runtime.side_stream(device))")
    rot, y_rot = rotate_inputs(data)
    if y_rot is not None:
        data.y = y_rot
    data["rotate_mat"] = rot
    data[ROTATED_KEY] = True
    # The list lengths (the step's one host read-back) are NOT waited for here: the graph is built in its sync-free form -- lengths on
    # the device, lists sized by their bounds -- and made exact at the top of the step that uses it, by a read issued on THIS stream
    # (GraphContext.make_exact honours `count_stream`), which by then has long drained.  Waiting here cost the loop 0.3-0.7 ms a
    # step: the side stream's few kernels share the chip with the step that was just enqueued.  Lists past
    # PREFETCH_SYNC_FREE_MAX_BYTES (two batches are alive in this loop, and bound-sized lists keep their size) take the synchronising form.
    gc = GraphContext.get(data, radius, H, noise, fake_agents=fake_agents, exact=not sync_free(), sync_free_cap=PREFETCH_SYNC_FREE_MAX_BYTES)
    gc.count_stream = side
    done = torch.cuda.Event()
    done.record(side)
    main.wait_event(done)
    for t in _tensors_in(list(data.as_dict().values()) + [gc.ws, gc.edges_ws, gc.rot] + list(gc._keep)):
        if t.is_cuda:
            t.record_stream(main)                       # freed while main-stream kernels still read it: not handed out again before they finish


def _tensors_in(obj):
    """every tensor inside nested lists / tuples / dicts (batch fields may hold containers of tensors)"""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_in(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_in(v)


def consume_rotation(data) -> bool:
    """True once for a batch that prefetch_graph has rotated already (the caller then skips MODEL:76-85)"""
    if ROTATED_KEY in data:
        del data[ROTATED_KEY]
        return True
    return False


def edge_snapshots(data, historical_steps: int) -> None:
    """The encoder's side effect on the batch (ENC:88-99, 107-110): `data['edge_index_{t}']` = the edges of the EXTENDED
    edge list (edge_index plus the in-edges of the target agents re-pointed at their fake copies, node ids N..N+A-1) whose
    two endpoints are valid at step t, in the order of the input list, and `data['edge_attr_{t}']` = pos[src,t] - pos[dst,t]
    (no radius filter: ENC:115 applies it only to the copies fed to the AA encoder).  The HIP path never materialises these
    21 lists (it works on one target-sorted, radius-filtered list); stages write them only when asked to
    (`preserve_side_effects=True`), with plain device-side tensor indexing -- they feed nothing on the hot path."""
    ei, pad, pos, agent = data["edge_index"], data["padding_mask"], data["positions"], data["agent_index"]
    N = data["x"].shape[0]
    to_agent = torch.isin(ei[1], agent)
    _, inv = torch.unique(ei[1][to_agent], return_inverse=True)
    new_edge = torch.cat((ei, torch.stack((ei[0][to_agent], inv + N))), dim=-1)
    orig = torch.cat((torch.arange(N, device=ei.device), agent))
    valid, pos_ext = ~pad[orig][:, :historical_steps], pos[orig]
    for t in range(historical_steps):
        keep = valid[new_edge[0], t] & valid[new_edge[1], t]
        e = new_edge[:, keep]
        data[f"edge_index_{t}"] = e
        data[f"edge_attr_{t}"] = pos_ext[e[0], t] - pos_ext[e[1], t]


class _TableCache:
    """float32 schedule tables (host replay of the solver's time bookkeeping) resident on the device."""

    def __init__(self) -> None:
        self._c: Dict[tuple, tuple] = {}

    def get(self, key: tuple, make, device) -> Tuple[EulerSchedule, torch.Tensor, torch.Tensor]:
        k = key + (str(device),)
        if k not in self._c:
            s: EulerSchedule = make()
            self._c[k] = (s, torch.from_numpy(s.step_table()).to(device), torch.from_numpy(s.out_table()).to(device))
        return self._c[k]


_TABLES = _TableCache()
_ENC_TABLES: Dict[tuple, np.ndarray] = {}
_TABLES_DEV: Dict[tuple, torch.Tensor] = {}          # device copies of the host step tables



class GradLayout:
    """names / shapes / offsets of a *_BWD stage's gradient buffers inside their flat tensor -- static per stage"""

    def __init__(self, names, shapes) -> None:
        self.names, self.shapes = list(names), list(shapes)
        self.sizes = [int(np.prod(sh)) if len(sh) else 1 for sh in self.shapes]
        # 16-byte aligned slices: kernels store float4 rows into some of them
        self.offs, total = [], 0
        for sz in self.sizes:
            self.offs.append(total)
            total += (sz + 3) // 4 * 4
        self.total = total
        self.index = {n: i for i, n in enumerate(self.names)}
        self.byte_offs = np.asarray(self.offs, dtype=np.uint64) * np.uint64(4)


class GradBuffers(collections.abc.Mapping):
    """The zero-initialised gradient buffers of a *_BWD stage: ONE flat tensor (one fill instead of one per parameter), keyed and
    ordered like the stage's parameter names.  A read-only mapping name -> view; the views are made when they are asked for -- the
    training loop's sink (driver.FlatGrads.accumulate_bundles) adds the flat tensor into its own in one gather and never asks
    (creating ~250 views a step was 0.25 ms of host time, and walking them again in the sink as much)."""

    def __init__(self, flat: torch.Tensor, layout: GradLayout) -> None:
        self.flat, self.layout, self._views = flat, layout, {}

    def __getitem__(self, name: str) -> torch.Tensor:
        v = self._views.get(name)
        if v is None:
            lay = self.layout
            i = lay.index[name]
            v = self._views[name] = self.flat[lay.offs[i]:lay.offs[i] + lay.sizes[i]].view(lay.shapes[i])
        return v

    def __iter__(self):
        return iter(self.layout.names)

    def __len__(self) -> int:
        return len(self.layout.names)

    def __contains__(self, name) -> bool:
        return name in self.layout.index

    def pointer_array(self):
        """(ctypes pointer to the void* array the C-ABI takes, keep-alive object): base address + static byte offsets"""
        ptrs = self.layout.byte_offs + np.uint64(self.flat.data_ptr())
        return ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), ptrs


def single_call_forms() -> bool:
    """TRAJSDE_STEP_SINGLE_CALLS=0: a training step packs its weight images stage by stage, gathers its gradients and runs AdamW through
    torch's element-wise launches, as before ABI 10 (same results bit for bit: tests/test_gpu_step_launches.py) -- the A/B switch of
    profiles/r05_ab_runs.md"""
    return os.environ.get("TRAJSDE_STEP_SINGLE_CALLS", "1") != "0"


def _drop_scale_channels(module, out: dict) -> None:
    """`uncertain: False` (DEC:100-101, dec_hivt_nusargo_grid.py:58-59): 'loc' is [K, N, T, 2].  The kernels always evaluate both
    heads into [K, N, T, 4] (the absent scale head is a zero stand-in, params.ParamTree.absent); the four-channel tensor stays
    under a private key for the backward entry points, which index it"""
    if not getattr(module, "uncertain", True):
        out["_loc4"] = out["loc"]
        out["loc"] = out["loc"][..., :2].contiguous()


def _loc4(out: dict) -> torch.Tensor:
    loc = out.get("_loc4", out["loc"])
    if loc.shape[-1] != 4:
        raise _lib.TrajsdeError("the decoder backward needs the forward's own output dict (its [K, N, T, 4] locations)")
    return loc.contiguous()


class PackSet:
    """The weight images of several (StageRuntime, stage id) pairs packed by ONE call (trajsde_pack_weights_many: three launches
    over a job table that stays on the device).  A training step re-packs six images after every optimizer step -- 26 launches and 6
    fills of ~5 us through `StageRuntime.blob()` one stage at a time; `refresh()` at the top of the step leaves every `blob()` call
    of the step a cache hit.  The blobs are allocated once and re-packed in place: what reads them is ordered behind the pack on the
    packing stream, or behind its event on any other."""

    def __init__(self, entries) -> None:
        self.entries = list(entries)                 # [(StageRuntime, stage id)]
        self._blobs = None
        self._items = None
        self._tables = None                          # (pinned host table, device table, bytes)
        self._tab_ids = None
        self._fresh = 1

    def refresh(self) -> None:
        stamps = [rt._stamp() for rt, _ in self.entries]
        if self._blobs is not None:
            for (rt, sid), stamp, blob in zip(self.entries, stamps, self._blobs):
                cached = rt._blobs.get(sid)
                if cached is None or cached[1] != stamp or cached[0] is not blob:
                    break
            else:
                return                               # every image is current
        L = _lib.lib()
        first = self.entries[0][0]._first_param()
        _require_gpu(first, "parameters")
        dev = first.device
        tabs = [rt._param_table(sid) for rt, sid in self.entries]
        if self._blobs is None or any(b.device != dev for b in self._blobs):
            self._blobs = [torch.empty(t[3], device=dev, dtype=torch.float32) for t in tabs]
            self._items = None
        if self._items is None or any(a is not b for a, b in zip(self._tab_ids, tabs)):
            items = (_lib.PackItem * len(tabs))()
            for it, (rt, sid), t, blob in zip(items, self.entries, tabs, self._blobs):
                nl, K = rt._dims()
                it.stage, it.num_layers, it.num_modes, it.n_params = sid, nl, K, t[2]
                it.params = C.cast(t[1], C.c_void_p)
                it.blob, it.blob_floats = blob.data_ptr(), t[3]
            need = int(L.trajsde_pack_many_table_bytes(items, len(tabs)))
            if need < 0:
                raise _lib.TrajsdeError("trajsde_pack_many_table_bytes: unknown stage")
            if self._tables is None or self._tables[2] < need or self._tables[1].device != dev:
                self._tables = (torch.empty(need, dtype=torch.uint8).pin_memory(), torch.empty(need, dtype=torch.uint8, device=dev), need)
                self._fresh = 1
            self._items, self._tab_ids = items, tabs
        host, table, nbytes = self._tables
        with torch.cuda.device(dev):
            # the images are re-packed IN PLACE: streams that read them since the last pack (StageRuntime.blob records them) may still
            # have kernels queued on the old contents -- the pack waits for what they hold now
            packing = torch.cuda.current_stream()
            for rt, sid in self.entries:
                for sid_, readers in list(rt._readers.items()):
                    if sid_ != sid:
                        continue
                    for handle, stream in readers.items():
                        if handle != packing.cuda_stream:
                            packing.wait_stream(stream)
                    readers.clear()
            _lib.check(L.trajsde_pack_weights_many(self._items, len(self.entries), host.data_ptr(), table.data_ptr(), nbytes, self._fresh,
                                                   _stream()), "trajsde_pack_weights_many")
            self._fresh = 0
            ev = torch.cuda.Event()
            cur = torch.cuda.current_stream()
            ev.record(cur)
        for (rt, sid), stamp, blob in zip(self.entries, stamps, self._blobs):
            rt._blobs[sid] = (blob, stamp, ev, cur.cuda_stream)


class StageRuntime:
    """Per-stage glue owned by a stage module (encoder / aggregator / decoder)."""

    STAGE_ID = {"encoder": _lib.STAGE_ENCODER, "aggregator": _lib.STAGE_AGGREGATOR, "decoder": _lib.STAGE_DECODER,
                "encoder_grid": _lib.STAGE_ENCODER_GRID, "decoder_mlp": _lib.STAGE_DECODER_MLP}

    def __init__(self, module, stage: str) -> None:
        object.__setattr__(self, "module", module)
        self.stage = stage
        self.stage_id = self.STAGE_ID[stage]
        self._blobs: Dict[int, tuple] = {}          # stage id -> (blob, parameter stamp, pack-done event, packing stream)
        self._readers: Dict[int, dict] = {}         # stage id -> {stream handle: stream} that read the image from another stream
        self._names: Dict[int, list] = {}

    # ---------------------------------------------------------------- weights
    def _dims(self) -> Tuple[int, int]:
        m = self.module
        if self.stage == "encoder_grid":             # this recipe takes the number of temporal layers
            return int(m.num_temporal_layers), 0
        return int(getattr(m, "num_layers", 0)), int(getattr(m, "num_modes", 0))

    def param_names(self, stage_id: Optional[int] = None):
        stage_id = self.stage_id if stage_id is None else stage_id
        if stage_id not in self._names:
            L = _lib.lib()
            nl, K = self._dims()
            n = L.trajsde_param_count(stage_id, nl, K)
            self._names[stage_id] = [L.trajsde_param_name(stage_id, i, nl, K).decode() for i in range(n)]
        return self._names[stage_id]

    def _first_param(self) -> torch.Tensor:
        """one parameter of the module (its device is the stage's); looked up through the cached slot, not `next(m.parameters())`
        -- that generator walks the module tree, 0.1 ms a call and nine calls a training step"""
        names = self.param_names()
        return self.module.p(names[0]) if names else next(self.module.parameters())

    def _grad_buffers(self, stage_id: int) -> "GradBuffers":
        """zero-initialised gradient buffers of a *_BWD stage as views of ONE flat tensor (one fill instead of one per
        parameter), keyed and ordered like param_names(stage_id)"""
        m = self.module
        lay = self._grad_layouts.get(stage_id) if hasattr(self, "_grad_layouts") else None
        if lay is None:                                                     # names / shapes / offsets do not change: computed once
            names = self.param_names(stage_id)
            lay = GradLayout(names, [tuple(m.p(n).shape) for n in names])
            if not hasattr(self, "_grad_layouts"):
                self._grad_layouts = {}
            self._grad_layouts[stage_id] = lay
        first = self._first_param()
        return GradBuffers(torch.zeros(lay.total, device=first.device, dtype=torch.float32), lay)

    # the parameters do not change between the forward and the backward calls of ONE training step: the path loss pins the stamp
    # for that stretch (pin_stamp / unpin_stamp), so the nine blob() lookups of a step walk the parameter list once per stage
    _pinned_stamp = None

    def pin_stamp(self) -> None:
        self._pinned_stamp = None
        self._pinned_stamp = self._stamp()

    def unpin_stamp(self) -> None:
        self._pinned_stamp = None

    def _stamp(self):
        if self._pinned_stamp is not None:
            return self._pinned_stamp
        return (self.module.version_stamp(), str(self._first_param().device))

    def _param_table(self, stage_id: int):
        """(identity, ctypes array of the parameter addresses, count, blob floats) of a stage's packing -- rebuilt (and the tensors
        re-checked) only when a tensor was replaced or moved: an optimizer step through the flat alias (touch()) re-packs from the
        same addresses"""
        m = self.module
        first = self._first_param()
        walk = m.walk_stamp() if hasattr(m, "walk_stamp") else None
        tab = self._ptr_tables.get(stage_id) if hasattr(self, "_ptr_tables") else None
        if tab is None or walk is None or tab[0] != (walk, str(first.device)):
            L = _lib.lib()
            nl, K = self._dims()
            names = self.param_names(stage_id)
            tensors = []
            for n in names:
                p = m.p(n)
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != first.device:
                    raise _lib.TrajsdeError(f"parameter {self.stage}.{n} must be contiguous fp32 on {first.device}")
                tensors.append(p)
            arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
            tab = ((walk, str(first.device)), arr, len(tensors), int(L.trajsde_blob_floats(stage_id, nl, K)))
            if not hasattr(self, "_ptr_tables"):
                self._ptr_tables = {}
            self._ptr_tables[stage_id] = tab
        return tab

    def blob(self, stage_id: Optional[int] = None) -> torch.Tensor:
        """Packed LDS images of this stage's weights (`stage_id`: the forward images by default, or the
        stage's backward images); re-packed whenever a parameter changed."""
        m = self.module
        stage_id = self.stage_id if stage_id is None else stage_id
        first = self._first_param()
        _require_gpu(first, f"{self.stage} parameters")
        stamp = self._stamp()
        cached = self._blobs.get(stage_id)
        if cached is None or stamp != cached[1]:
            L = _lib.lib()
            nl, K = self._dims()
            _, arr, n_tensors, n_floats = self._param_table(stage_id)
            blob = torch.empty(n_floats, device=first.device, dtype=torch.float32)
            with torch.cuda.device(first.device):
                _lib.check(L.trajsde_pack_weights(stage_id, nl, K, arr, n_tensors, blob.data_ptr(), n_floats, _stream()),
                           "trajsde_pack_weights")
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            self._blobs[stage_id] = (blob, stamp, ev, torch.cuda.current_stream(first.device).cuda_stream)
        blob, stamp_, ev, packed_on = self._blobs[stage_id]
        cur = torch.cuda.current_stream(first.device)
        if cur.cuda_stream != packed_on:
            # a reader on another stream (multi-stream inference, a validation stream beside the training loop): remembered, so that an
            # IN-PLACE re-pack of this image (PackSet.refresh) is ordered behind what the stream still has queued on it
            self._readers.setdefault(stage_id, {})[cur.cuda_stream] = cur
        if ev is not None:
            if ev.query():
                self._blobs[stage_id] = (blob, stamp_, None, packed_on)        # pack finished: nothing left to order against
            elif cur.cuda_stream != packed_on:
                cur.wait_event(ev)          # packed on another stream (multi-stream drivers): order the reads behind the pack
        return blob

    def packs_complete(self) -> None:
        """the caller has synchronised the device: no pack is in flight, nothing left to order reads against"""
        for sid, (blob, stamp, ev, packed_on) in list(self._blobs.items()):
            self._blobs[sid] = (blob, stamp, None, packed_on)

    # ---------------------------------------------------------------- decoder
    def decoder_forward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor,
                        noise: Optional[NoiseSpec] = None) -> Dict[str, torch.Tensor]:
        m = self.module
        noise = NoiseSpec.resolve(noise)
        _require_gpu(local_embed, "local_embed")
        dev = local_embed.device
        K, T = int(m.num_modes), int(m.future_steps)
        N = local_embed.shape[0]
        sched, step_tab, out_tab = _TABLES.get(("dec", T, float(m.max_fut_t), float(m.min_stepsize)),
                                               lambda: decoder_schedule(T, float(m.max_fut_t), float(m.min_stepsize)), dev)
        if noise.z_dec is not None and tuple(noise.z_dec.shape) != (sched.n_euler, K * N, D):
            raise _lib.TrajsdeError(f"z_dec must be [{sched.n_euler},{K * N},{D}] (one increment per Euler step, App. D)")
        L = _lib.lib()
        blob = self.blob()
        loc = torch.empty(K, N, T, 4, device=dev, dtype=torch.float32)
        pi = torch.empty(N, K, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_decoder_ws_bytes(N, K)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        cn = noise.c_noise(noise.z_dec, noise.dec_row_ids)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_decoder_forward(N, K, T, blob.data_ptr(), local_embed.contiguous().data_ptr(),
                                                 global_embed.contiguous().data_ptr(), step_tab.data_ptr(), sched.n_euler,
                                                 out_tab.data_ptr(), float(m.min_scale), C.byref(cn), ws.data_ptr(), ws_bytes,
                                                 loc.data_ptr(), pi.data_ptr(), _stream()), "trajsde_decoder_forward")
        out = {"loc": loc, "pi": pi, "reg_mask": ~data["padding_mask"][:, -T:]}          # DEC:104
        _drop_scale_channels(m, out)
        return out

    def decoder_nll_backward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor, out: Dict[str, torch.Tensor],
                             noise: NoiseSpec, eps: float = 1e-6) -> Dict[str, object]:
        """Laplace negative log-likelihood of the winning mode (losses/laplace_nll_loss.py:18-47) on `out` = decoder_forward(...)
        and its gradients -- like decoder_l2_backward, with the scale head differentiated as well (grads keyed by
        param_names(STAGE_DECODER_NLL_BWD))."""
        return self._decoder_backward(data, local_embed, global_embed, out, noise, nll_eps=float(eps))

    def decoder_l2_backward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor, out: Dict[str, torch.Tensor],
                            noise: NoiseSpec) -> Dict[str, object]:
        """Winner-takes-all L2 loss (losses/L2.py:10-27) on `out` = decoder_forward(...) and its gradients w.r.t.
        this stage's parameters and inputs.  `noise` must be the NoiseSpec the forward ran with (same seed / z), so
        that the replayed winning paths are the forward's.  Returns {"loss", "best_mode", "grads": {param name:
        tensor}, "d_local_embed", "d_global_embed"}; `pi.*` and `scale.*` get no gradient from this loss."""
        return self._decoder_backward(data, local_embed, global_embed, out, noise, nll_eps=None)

    def _decoder_backward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor, out: Dict[str, torch.Tensor],
                          noise: NoiseSpec, nll_eps: Optional[float]) -> Dict[str, object]:
        """shared body of decoder_l2_backward / decoder_nll_backward (`nll_eps` None: L2)"""
        m = self.module
        if noise is None:
            raise _lib.TrajsdeError("decoder_l2_backward needs the NoiseSpec of the forward pass (seed or z_dec)")
        if nll_eps is not None and not getattr(m, "uncertain", True):
            raise _lib.TrajsdeError("LaplaceNLLLoss needs the decoder's scale head: `uncertain: False` has none (DEC:56, "
                                    "losses/laplace_nll_loss.py:28 chunks loc | scale out of four channels)")
        _require_gpu(local_embed, "local_embed")
        dev = local_embed.device
        K, T = int(m.num_modes), int(m.future_steps)
        N = local_embed.shape[0]
        sched, step_tab, out_tab = _TABLES.get(("dec", T, float(m.max_fut_t), float(m.min_stepsize)),
                                               lambda: decoder_schedule(T, float(m.max_fut_t), float(m.min_stepsize)), dev)
        y = data["y"]
        if y is None or tuple(y.shape) != (N, T, 2):
            raise _lib.TrajsdeError(f"data['y'] must be [{N},{T},2] (rotated targets, MODEL:83-84)")
        y = y.to(torch.float32).contiguous()
        mask = out["reg_mask"].contiguous().view(torch.uint8)
        L = _lib.lib()
        stage = _lib.STAGE_DECODER_BWD if nll_eps is None else _lib.STAGE_DECODER_NLL_BWD
        names = self.param_names(stage)
        grads = self._grad_buffers(stage)
        arr, _keep = grads.pointer_array()
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        best = torch.empty(N, device=dev, dtype=torch.int32)
        d_local = torch.empty(N, D, device=dev, dtype=torch.float32)
        d_global = torch.empty(K, N, D, device=dev, dtype=torch.float32)
        cn = noise.c_noise(noise.z_dec, noise.dec_row_ids)
        head = (N, K, T, self.blob().data_ptr(), self.blob(stage).data_ptr(),
                local_embed.contiguous().data_ptr(), global_embed.contiguous().data_ptr(), step_tab.data_ptr(), sched.n_euler,
                out_tab.data_ptr(), C.byref(cn), _loc4(out).data_ptr(), y.data_ptr(), mask.data_ptr())
        with torch.cuda.device(dev):
            if nll_eps is None:
                ws_bytes = L.trajsde_decoder_backward_ws_bytes(N, K, T, sched.n_euler)
                ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
                _lib.check(L.trajsde_decoder_l2_backward(*head, ws.data_ptr(), ws_bytes, loss.data_ptr(), best.data_ptr(), arr, len(names),
                                                         d_local.data_ptr(), d_global.data_ptr(), _stream()), "trajsde_decoder_l2_backward")
            else:
                ws_bytes = L.trajsde_decoder_nll_backward_ws_bytes(N, K, T, sched.n_euler)
                ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
                _lib.check(L.trajsde_decoder_nll_backward(*head, float(nll_eps), float(m.min_scale), ws.data_ptr(), ws_bytes, loss.data_ptr(),
                                                          best.data_ptr(), arr, len(names), d_local.data_ptr(), d_global.data_ptr(),
                                                          _stream()), "trajsde_decoder_nll_backward")
        return {"loss": loss[0], "best_mode": best, "grads": grads, "d_local_embed": d_local, "d_global_embed": d_global}

    # ---------------------------------------------------------------- encoder
    def encoder_forward(self, data, noise: Optional[NoiseSpec] = None, preserve_side_effects: bool = False):
        """LocalEncoderSDESepPara2.forward (ENC:66-202) -> (local_embed, diff_in, diff_out, label_in, label_out).
        `preserve_side_effects`: also leave `edge_index_{t}` / `edge_attr_{t}` on the batch like ENC:107-110 does."""
        m = self.module
        noise = NoiseSpec.resolve(noise)
        cap = bool(getattr(m, "capture_intermediates", False))
        L = _lib.lib()
        prev = L.trajsde_export_senders(1) if cap else 0                  # edge lists with their senders, for the tests
        try:
            gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), noise, want_senders=cap,
                                  exact=True if cap else not sync_free())
        finally:
            if cap:
                L.trajsde_export_senders(prev)
        dev = gc.device
        blob = self.blob()
        H, N, A, Nt = gc.batch.H, gc.batch.N, gc.batch.A, gc.graph.Nt
        tab = self._enc_table()
        if noise.z_enc is not None and tuple(noise.z_enc.shape) != (H, Nt, D):
            raise _lib.TrajsdeError(f"z_enc must be [{H},{Nt},{D}]")
        local = torch.empty(N, D, device=dev, dtype=torch.float32)
        diff_pick = torch.empty(2 * A, D, device=dev, dtype=torch.float32)
        aa_out = torch.empty(H, Nt, D, device=dev, dtype=torch.float32) if cap else None
        latent = torch.empty(H, N, D, device=dev, dtype=torch.float32) if cap else None
        ws_bytes = L.trajsde_encoder_ws_bytes(C.byref(gc.batch), C.byref(gc.graph))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        cn = noise.c_noise(noise.z_enc, noise.enc_row_ids)
        dr = noise.c_dropout(m)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_forward(C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), blob.data_ptr(),
                                                 tab.ctypes.data_as(C.c_void_p), C.byref(cn), ws.data_ptr(), ws_bytes,
                                                 local.data_ptr(), diff_pick.data_ptr(), _ptr(aa_out), _ptr(latent),
                                                 C.byref(dr) if dr is not None else None, _stream()),
                       "trajsde_encoder_forward")
        if cap:
            if state_storage() == "bf16":                                  # aa_out was written with 2-byte elements
                aa_out = aa_out.view(-1).view(torch.bfloat16)[:H * Nt * D].view(H, Nt, D).float()
            m.last_intermediates = {"aa_out": aa_out, "latent_ys": latent, **gc.true_counts(), **gc.edge_lists()}
        if preserve_side_effects:
            edge_snapshots(data, H)                                                         # ENC:107-110
        diff_in, diff_out = torch.chunk(diff_pick, 2, 0)                                    # ENC:194
        return (local, diff_in, diff_out, torch.full_like(diff_in, m.real_label), torch.full_like(diff_out, m.fake_label))

    def encoder_forward_train(self, data, noise: NoiseSpec):
        """The encoder's forward for a training step: same outputs as encoder_forward (same weights, noise, dropout), computed
        by the kernels that keep the tape the backward needs.  Returns (outputs tuple, tape) -- hand `tape` to encoder_backward
        and the backward does not recompute the forward."""
        m = self.module
        gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), noise)
        dev = gc.device
        L = _lib.lib()
        N, A = gc.batch.N, gc.batch.A
        tab = self._enc_table()
        local = torch.empty(N, D, device=dev, dtype=torch.float32)
        diff_pick = torch.empty(2 * A, D, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_encoder_tape_bytes(C.byref(gc.batch), C.byref(gc.graph))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        cn = noise.c_noise(noise.z_enc, noise.enc_row_ids)
        dr = noise.c_dropout(m)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_forward_train(C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), self.blob().data_ptr(),
                                                       tab.ctypes.data_as(C.c_void_p), C.byref(cn), ws.data_ptr(), ws_bytes, local.data_ptr(),
                                                       diff_pick.data_ptr(), C.byref(dr) if dr is not None else None, _stream()),
                       "trajsde_encoder_forward_train")
        diff_in, diff_out = torch.chunk(diff_pick, 2, 0)
        outs = (local, diff_in, diff_out, torch.full_like(diff_in, m.real_label), torch.full_like(diff_out, m.fake_label))
        return outs, (ws, ws_bytes)

    def encoder_backward(self, data, d_local: torch.Tensor, noise: NoiseSpec, diff_weight: float = 1.0,
                         want_boundaries: bool = False, tape=None, keep_scratch: bool = False) -> Dict[str, object]:
        """Backward of LocalEncoderSDESepPara2.forward plus the DiffBCE term: `d_local` = dL/d local_embed [N,64],
        `noise` the forward's NoiseSpec.  Returns {"grads": {param name: tensor}, "diff_loss": diff_weight * DiffBCE}
        (+ "d_latent" [N,64], "d_aa_out" [H,Nt,64] with want_boundaries).  `tape`: what encoder_forward_train returned for
        this very step; without it the forward is recomputed inside.  `keep_scratch` (reproducibility checks): the backward's
        scratch buffer starts zeroed and is returned as "_scratch"."""
        m = self.module
        if noise is None:
            raise _lib.TrajsdeError("encoder_backward needs the NoiseSpec of the forward pass")
        gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), noise)
        dev = gc.device
        L = _lib.lib()
        H, N, Nt = gc.batch.H, gc.batch.N, gc.graph.Nt
        if tuple(d_local.shape) != (N, D):
            raise _lib.TrajsdeError(f"d_local must be [{N},{D}]")
        tab = self._enc_table()
        tab_dev = _TABLES_DEV.get((id(tab), str(dev)))                     # (not setdefault: its default would be built -- a blocking
        if tab_dev is None:                                                #  pageable host-to-device copy -- on every call)
            tab_dev = _TABLES_DEV[(id(tab), str(dev))] = torch.from_numpy(tab).to(dev)
        names = self.param_names(_lib.STAGE_ENCODER_BWD)
        grads = self._grad_buffers(_lib.STAGE_ENCODER_BWD)
        arr, _keep = grads.pointer_array()
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        d_lat = torch.empty(N, D, device=dev, dtype=torch.float32) if want_boundaries else None
        d_aa = torch.empty(H, Nt, D, device=dev, dtype=torch.float32) if want_boundaries else None
        if tape is not None:
            ws, ws_bytes = tape
        else:
            ws_bytes = L.trajsde_encoder_tape_bytes(C.byref(gc.batch), C.byref(gc.graph))
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        # the backward's own scratch is a separate buffer, allocated only now: between the training forward and this call the
        # step holds the tape alone, and the decoder's / aggregator's workspaces have been released by the time this one is taken
        sc_bytes = L.trajsde_encoder_backward_scratch_bytes(C.byref(gc.batch), C.byref(gc.graph))
        scratch = (torch.zeros if keep_scratch else torch.empty)(sc_bytes, device=dev, dtype=torch.uint8)
        cn = noise.c_noise(noise.z_enc, noise.enc_row_ids)
        dr = noise.c_dropout(m)                                            # the forward's masks, regenerated from the same key
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_backward(
                C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), self.blob().data_ptr(),
                self.blob(_lib.STAGE_ENCODER_BWD).data_ptr(), tab.ctypes.data_as(C.c_void_p), tab_dev.data_ptr(), C.byref(cn),
                d_local.to(torch.float32).contiguous().data_ptr(), float(diff_weight), ws.data_ptr(), ws_bytes, loss.data_ptr(),
                arr, len(names), _ptr(d_lat), _ptr(d_aa), C.byref(dr) if dr is not None else None, 1 if tape is not None else 0,
                scratch.data_ptr(), sc_bytes, _stream()),
                "trajsde_encoder_backward")
        out = {"grads": grads, "diff_loss": loss[0]}
        if want_boundaries:
            out.update(d_latent=d_lat, d_aa_out=d_aa)
        if keep_scratch:
            out["_scratch"] = scratch
        return out

    def _enc_table(self) -> np.ndarray:
        m = self.module
        key = (int(m.historical_steps), float(m.max_past_t), float(m.minimum_step), bool(m.run_backwards))
        if key not in _ENC_TABLES:               # host replay of the solver's float32 time bookkeeping: once per config
            sched = encoder_schedule(*key)
            if sched.n_euler != key[0] or not np.all(sched.out_w1 == 1.0):
                raise _lib.TrajsdeError("encoder schedule is not one Euler step per interval (SURVEY App. D)")
            _ENC_TABLES[key] = np.ascontiguousarray(sched.step_table())
        return _ENC_TABLES[key]

    def encoder_forward_ood(self, data, noise: Optional[NoiseSpec] = None, n_samples: int = 10):
        """LocalEncoderSDESepPara2.forward_ood (ENC:204-370) -> (local_embed, actors_std)."""
        m = self.module
        noise = NoiseSpec.resolve(noise)
        gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), noise, fake_agents=False)
        dev = gc.device
        L = _lib.lib()
        blob = self.blob()
        H, N = gc.batch.H, gc.batch.N
        tab = self._enc_table()
        if noise.z_enc is not None and tuple(noise.z_enc.shape) != (n_samples * H, N, D):
            raise _lib.TrajsdeError(f"z_enc must be [{n_samples * H},{N},{D}] for forward_ood")
        local = torch.empty(N, D, device=dev, dtype=torch.float32)
        stds = torch.empty(N, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_encoder_ood_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), n_samples)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        cn = noise.c_noise(noise.z_enc, noise.enc_row_ids)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_forward_ood(C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), blob.data_ptr(),
                                                     tab.ctypes.data_as(C.c_void_p), C.byref(cn), n_samples, ws.data_ptr(), ws_bytes,
                                                     local.data_ptr(), stds.data_ptr(), _stream()), "trajsde_encoder_forward_ood")
        return local, stds

    # ---------------------------------------------------------------- aggregator
    def aggregator_forward(self, data, local_embed: torch.Tensor, noise: Optional[NoiseSpec] = None,
                           prepared: Optional["RelPrefetch"] = None) -> torch.Tensor:
        """GlobalInteractor.forward (AGG:38-58) -> [K, N, 64].  `noise`: only its dropout key is used, in train mode.
        `prepared`: what `prefetch_rel_embed` returned for this forward -- the relative-pose rows are then already in the
        workspace (or on their way on the side stream: this stream waits for them) and the call skips that kernel."""
        m = self.module
        dr = NoiseSpec.resolve(noise).c_dropout(m) if (m.training and float(getattr(m, "dropout", 0.0) or 0.0) > 0) else None
        _require_gpu(local_embed, "local_embed")
        gc = GraphContext.get(data, None, int(m.historical_steps), None, exact=None)
        dev = gc.device
        L = _lib.lib()
        blob = self.blob()
        K, N = int(m.num_modes), gc.batch.N
        out = torch.empty(K, N, D, device=dev, dtype=torch.float32)
        if prepared is not None and prepared.gc is not gc:
            prepared.join()                                               # (the graph was rebuilt in between: the rows are of another list)
            prepared = None
        if prepared is not None:
            ws, ws_bytes = prepared.ws, prepared.ws_bytes
            prepared.join()
            entry = L.trajsde_aggregator_forward_prepared
        else:
            ws_bytes = L.trajsde_aggregator_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), K)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            entry = L.trajsde_aggregator_forward_heads
        with torch.cuda.device(dev):
            _lib.check(entry(C.byref(gc.batch), C.byref(gc.graph), blob.data_ptr(), int(m.num_layers), K,
                             int(m.num_heads), local_embed.contiguous().data_ptr(), ws.data_ptr(),
                             ws_bytes, out.data_ptr(), C.byref(dr) if dr is not None else None, _stream()),
                       "trajsde_aggregator_forward")
        return out

    def arm_rel_prefetch(self, data, encoder_module) -> Optional["torch.cuda.Stream"]:
        """Called on the AGGREGATOR's runtime before the encoder of an inference forward: names the side stream on which the
        relative-pose embedding (AGG:42-51; it reads the graph stage's output only) will run, and has the encoder call fork it
        at the point where its recurrence starts (trajsde_encoder_fork_stream).  That kernel fills the chip for ~0.2 ms and depends
        on nothing the encoder computes, while the recurrence keeps a third of the CUs idle at one wave per SIMD for about as
        long.  None when switched off (the default; TRAJSDE_OVERLAP_REL=1 turns it on) or where the split entry points do not apply."""
        if not _OVERLAP_REL:
            return None
        m, enc = self.module, encoder_module
        if bool(getattr(enc, "capture_intermediates", False)) or m.training or enc.training:
            return None
        x = data["x"]
        _require_gpu(x, "data['x']")
        side = _side_stream(x.device, torch.cuda.current_stream(x.device))
        _lib.check(_lib.lib().trajsde_encoder_fork_stream(side.cuda_stream), "trajsde_encoder_fork_stream")
        return side

    def launch_rel_prefetch(self, data, side) -> Optional["RelPrefetch"]:
        """after the encoder call returned: the side stream now waits for the fork event; enqueue the embedding on it"""
        if side is None or GraphContext.KEY not in data:
            return None
        m = self.module
        gc = GraphContext.get(data, None, int(m.historical_steps), None, exact=None)
        dev = gc.device
        L = _lib.lib()
        blob = self.blob()                                                # (packed on the current stream before the side kernel reads it:
        ws_bytes = L.trajsde_aggregator_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), int(m.num_modes))     # see pack event below)
        packed = self._blobs[self.stage_id][2]                            # a pack still in flight on another stream: order behind it
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(side):
            # the workspace comes from the SIDE stream's share of the caching allocator: a block of the current stream's could be
            # the encoder workspace that was released a moment ago on the host while the encoder's kernels -- beside which the
            # side kernel is about to run -- still use it
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            if packed is not None:
                side.wait_event(packed)
            _lib.check(L.trajsde_aggregator_prepare(C.byref(gc.batch), C.byref(gc.graph), blob.data_ptr(), ws.data_ptr(), ws_bytes,
                                                    side.cuda_stream), "trajsde_aggregator_prepare")
            done = torch.cuda.Event()
            done.record(side)
        ws.record_stream(cur)                                             # ... and is used on the current stream from the join on
        return RelPrefetch(gc, ws, ws_bytes, done, dev)

    # ---------------------------------------------------------------- vanilla HiVT variant
    def encoder_grid_forward(self, data, noise: Optional[NoiseSpec] = None) -> torch.Tensor:
        """LocalEncoder.forward (enc_hivt_nusargo_grid.py:52-93) -> local_embed [N,64].  In train mode the module's `dropout` is
        applied at the reference's sites with masks keyed by `noise` (csrc/dropout.hpp); the variant draws no other noise."""
        m = self.module
        dr = noise.c_dropout(m) if noise is not None else None
        if dr is None and m.training and float(getattr(m, "dropout", 0.0) or 0.0) > 0:
            raise _lib.TrajsdeError("encoder_grid_forward in train mode needs a NoiseSpec (dropout key)")
        gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), NoiseSpec(seed=0), fake_agents=False)   # no fake agents: the graph only
        dev = gc.device
        L = _lib.lib()
        blob = self.blob()
        N = gc.batch.N
        local = torch.empty(N, D, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_encoder_grid_ws_bytes(C.byref(gc.batch), C.byref(gc.graph))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_grid_forward_train(C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), blob.data_ptr(),
                                                            int(m.num_heads), int(m.num_temporal_layers), ws.data_ptr(), ws_bytes,
                                                            local.data_ptr(), C.byref(dr) if dr is not None else None, _stream()),
                       "trajsde_encoder_grid_forward_train")
        return local

    def encoder_grid_backward(self, data, d_local: torch.Tensor, noise: Optional[NoiseSpec] = None) -> Dict[str, object]:
        """backward of the vanilla LocalEncoder: dL/d local_embed [N,64] -> {"grads": {param name: tensor}}; `noise`: the forward's
        (train mode: its dropout key -- the masks are regenerated)"""
        m = self.module
        dr = noise.c_dropout(m) if noise is not None else None
        if dr is None and m.training and float(getattr(m, "dropout", 0.0) or 0.0) > 0:
            raise _lib.TrajsdeError("encoder_grid_backward in train mode needs the NoiseSpec of the forward pass (dropout key)")
        gc = GraphContext.get(data, float(m.local_radius), int(m.historical_steps), NoiseSpec(seed=0), fake_agents=False)
        dev = gc.device
        L = _lib.lib()
        N, nl = gc.batch.N, int(m.num_temporal_layers)
        if tuple(d_local.shape) != (N, D):
            raise _lib.TrajsdeError(f"d_local must be [{N},{D}]")
        names = self.param_names(_lib.STAGE_ENCODER_GRID_BWD)
        grads = self._grad_buffers(_lib.STAGE_ENCODER_GRID_BWD)
        arr, _keep = grads.pointer_array()
        ws_bytes = L.trajsde_encoder_grid_backward_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), nl)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_encoder_grid_backward_train(
                C.byref(gc.batch), C.byref(gc.graph), gc.rot.data_ptr(), self.blob().data_ptr(),
                self.blob(_lib.STAGE_ENCODER_GRID_BWD).data_ptr(), int(m.num_heads), nl,
                d_local.to(torch.float32).contiguous().data_ptr(), ws.data_ptr(), ws_bytes, arr, len(names),
                C.byref(dr) if dr is not None else None, _stream()), "trajsde_encoder_grid_backward_train")
        return {"grads": grads}

    def mlp_decoder_forward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor) -> Dict[str, torch.Tensor]:
        """MLPDecoder.forward (dec_hivt_nusargo_grid.py:47-63)"""
        m = self.module
        _require_gpu(local_embed, "local_embed")
        dev = local_embed.device
        K, T, N = int(m.num_modes), int(m.future_steps), local_embed.shape[0]
        L = _lib.lib()
        blob = self.blob()
        loc = torch.empty(K, N, T, 4, device=dev, dtype=torch.float32)
        pi = torch.empty(N, K, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_mlp_decoder_ws_bytes(N, K)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_mlp_decoder_forward(N, K, T, blob.data_ptr(), local_embed.contiguous().data_ptr(),
                                                     global_embed.contiguous().data_ptr(), float(m.min_scale), ws.data_ptr(),
                                                     ws_bytes, loc.data_ptr(), pi.data_ptr(), _stream()),
                       "trajsde_mlp_decoder_forward")
        out = {"loc": loc, "pi": pi, "reg_mask": ~data["padding_mask"][:, -T:]}
        if getattr(m, "uncertain", True):                    # (the reference returns the embeddings on this branch only, :56-57)
            out["local_embed"], out["global_embed"] = local_embed, global_embed
        _drop_scale_channels(m, out)
        return out


    def aggregator_forward_train(self, data, local_embed: torch.Tensor, noise: Optional[NoiseSpec] = None):
        """GlobalInteractor.forward for a training step -> (global_embed [K,N,64], tape for aggregator_backward)"""
        m = self.module
        _require_gpu(local_embed, "local_embed")
        gc = GraphContext.get(data, None, int(m.historical_steps), None)
        dev = gc.device
        L = _lib.lib()
        K, N, nl = int(m.num_modes), gc.batch.N, int(m.num_layers)
        out = torch.empty(K, N, D, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_aggregator_backward_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), nl, K)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        dr = noise.c_dropout(m) if noise is not None else None
        if dr is None and m.training and float(getattr(m, "dropout", 0.0) or 0.0) > 0:
            raise _lib.TrajsdeError("aggregator_forward_train in train mode needs a NoiseSpec (dropout key)")
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_aggregator_forward_train(C.byref(gc.batch), C.byref(gc.graph), self.blob().data_ptr(), nl, K, int(m.num_heads),
                                                          local_embed.contiguous().data_ptr(), ws.data_ptr(), ws_bytes, out.data_ptr(),
                                                          C.byref(dr) if dr is not None else None, _stream()), "trajsde_aggregator_forward_train")
        return out, (ws, ws_bytes)

    def aggregator_backward(self, data, local_embed: torch.Tensor, d_global: torch.Tensor,
                            noise: Optional[NoiseSpec] = None, tape=None) -> Dict[str, object]:
        """Backward of GlobalInteractor.forward: dL/d global_embed [K,N,64] -> {"grads": {param name: tensor},
        "d_local_embed": [N,64]} (the aggregator's own contribution; the decoder's d local_embed is added by the
        caller).  The forward is recomputed inside the call."""
        m = self.module
        _require_gpu(local_embed, "local_embed")
        gc = GraphContext.get(data, None, int(m.historical_steps), None)
        dev = gc.device
        L = _lib.lib()
        K, N, nl = int(m.num_modes), gc.batch.N, int(m.num_layers)
        if tuple(d_global.shape) != (K, N, D):
            raise _lib.TrajsdeError(f"d_global must be [{K},{N},{D}]")
        names = self.param_names(_lib.STAGE_AGGREGATOR_BWD)
        grads = self._grad_buffers(_lib.STAGE_AGGREGATOR_BWD)
        arr, _keep = grads.pointer_array()
        d_local = torch.empty(N, D, device=dev, dtype=torch.float32)
        if tape is not None:
            ws, ws_bytes = tape
        else:
            ws_bytes = L.trajsde_aggregator_backward_ws_bytes(C.byref(gc.batch), C.byref(gc.graph), nl, K)
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        dr = noise.c_dropout(m) if noise is not None else None            # the forward's masks
        if dr is None and m.training and float(getattr(m, "dropout", 0.0) or 0.0) > 0:
            raise _lib.TrajsdeError("aggregator_backward in train mode needs the NoiseSpec of the forward pass (dropout key)")
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_aggregator_backward_heads(
                C.byref(gc.batch), C.byref(gc.graph), self.blob().data_ptr(), self.blob(_lib.STAGE_AGGREGATOR_BWD).data_ptr(), nl, K,
                int(m.num_heads), local_embed.contiguous().data_ptr(), d_global.to(torch.float32).contiguous().data_ptr(), ws.data_ptr(), ws_bytes,
                arr, len(names), d_local.data_ptr(), C.byref(dr) if dr is not None else None, 1 if tape is not None else 0, _stream()),
                "trajsde_aggregator_backward")
        return {"grads": grads, "d_local_embed": d_local}


    def mlp_decoder_l2_backward(self, data, local_embed: torch.Tensor, global_embed: torch.Tensor,
                                out: Dict[str, torch.Tensor]) -> Dict[str, object]:
        """L2 (winner takes all) on `out` = mlp_decoder_forward(...) and its gradients, like decoder_l2_backward"""
        m = self.module
        dev = local_embed.device
        K, T, N = int(m.num_modes), int(m.future_steps), local_embed.shape[0]
        y = data["y"]
        if y is None or tuple(y.shape) != (N, T, 2):
            raise _lib.TrajsdeError(f"data['y'] must be [{N},{T},2] (rotated targets)")
        y = y.to(torch.float32).contiguous()
        mask = out["reg_mask"].contiguous().view(torch.uint8)
        L = _lib.lib()
        names = self.param_names(_lib.STAGE_DECODER_MLP_BWD)
        grads = self._grad_buffers(_lib.STAGE_DECODER_MLP_BWD)
        arr, _keep = grads.pointer_array()
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        best = torch.empty(N, device=dev, dtype=torch.int32)
        d_local = torch.empty(N, D, device=dev, dtype=torch.float32)
        d_global = torch.empty(K, N, D, device=dev, dtype=torch.float32)
        ws_bytes = L.trajsde_mlp_decoder_backward_ws_bytes(N)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            _lib.check(L.trajsde_mlp_decoder_l2_backward(
                N, K, T, self.blob(_lib.STAGE_DECODER_MLP_BWD).data_ptr(), local_embed.contiguous().data_ptr(),
                global_embed.contiguous().data_ptr(), _loc4(out).data_ptr(), y.data_ptr(), mask.data_ptr(), ws.data_ptr(),
                ws_bytes, loss.data_ptr(), best.data_ptr(), arr, len(names), d_local.data_ptr(), d_global.data_ptr(), _stream()),
                "trajsde_mlp_decoder_l2_backward")
        return {"loss": loss[0], "best_mode": best, "grads": grads, "d_local_embed": d_local, "d_global_embed": d_global}


class GraphContext:
    """Device-side graph structures of one batch (CSR, compacted edge lists, segment pointers), built once per
    forward by the first stage that needs them and parked on the batch object for the next stage."""

    KEY = "_trajsde_graph"
    DEFAULT_RADIUS = 50.0
    _DEVICE_KEY_TOKENS = itertools.count(1)

    def __init__(self, data, radius: float, H: int, noise: Optional[NoiseSpec], fake_agents: bool = True,
                 exact: bool = True, sync_free_cap: Optional[int] = None) -> None:
        L = _lib.lib()
        self.fake_agents = fake_agents
        x = data["x"]
        _require_gpu(x, "data['x']")
        self.device = dev = x.device
        keep = self._keep = []

        def f32(t):
            t = t.to(torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        def u8(t):
            t = t.contiguous()
            t = t.view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8)
            keep.append(t)
            return t.data_ptr()

        def i64(t):
            t = t.to(torch.int64).contiguous()
            keep.append(t)
            return t.data_ptr() if t.numel() else None

        N, Hx = x.shape[0], x.shape[1]
        if Hx != H:
            raise _lib.TrajsdeError(f"data['x'] has {Hx} history steps, the stage was built for {H}")
        if "rotate_mat" not in data or data["rotate_mat"] is None or data["rotate_mat"] is False:
            raise _lib.TrajsdeError("data['rotate_mat'] is missing: call the model's forward (MODEL:75-85) or runtime.rotate_inputs")
        self.rot = data["rotate_mat"].to(torch.float32).contiguous()
        ei, lai = data["edge_index"], data["lane_actor_index"]
        lp = data["lane_positions"]
        n_agents = data["agent_index"].numel() if fake_agents else 0      # A = 0: no fake rows (forward_ood, ENC:204-370)
        self.batch = _lib.Batch(N, n_agents, ei.shape[1], lp.shape[0], lai.shape[1], H,
                                data["positions"].shape[1], lp.shape[1] if lp.dim() > 1 else 0,
                                f32(x), f32(data["positions"]), u8(data["padding_mask"]), u8(data["bos_mask"]),
                                f32(data["rotate_angles"]), i64(ei), i64(data["agent_index"]) if fake_agents else None, i64(data["batch"]),
                                i64(data["source"]), f32(lp), f32(data["lane_paddings"]), i64(lai),
                                f32(data["lane_actor_vectors"]))
        if data["padding_mask"].shape[1] != data["positions"].shape[1]:
            raise _lib.TrajsdeError("padding_mask and positions must cover the same time slots")
        self.graph = _lib.Graph()
        noise = NoiseSpec.resolve(noise)
        z_fake = noise.z_fake if fake_agents else None
        if z_fake is not None and tuple(z_fake.shape) != (self.batch.A, H, 2):
            raise _lib.TrajsdeError(f"z_fake must be [{self.batch.A},{H},2]")
        cn = noise.c_noise(z_fake, noise.fake_row_ids)
        with torch.cuda.device(dev):
            ws_bytes = L.trajsde_graph_ws_bytes(C.byref(self.batch))
            if ws_bytes < 0:
                raise _lib.TrajsdeError("graph workspace query failed: " + L.trajsde_last_error().decode())
            self.ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
            # sync-free form: list lengths stay on the device, the E_* fields are bounds (int32 positions must hold them)
            b = self.batch
            # ... and the agent-agent list is allocated for the bound 2 H E (24 B per slot): past SYNC_FREE_MAX_BYTES the
            # one synchronisation is the better deal (8 scenes x 1024 agents would reserve 8 GB for 90 MB of edges)
            cap = SYNC_FREE_MAX_BYTES if sync_free_cap is None else int(sync_free_cap)
            if not exact and (2 * b.H * b.E >= 2 ** 31 - 2 or 24 * 2 * b.H * b.E > cap):
                exact = True
            prepare = L.trajsde_graph_prepare if exact else L.trajsde_graph_prepare_async
            _lib.check(prepare(C.byref(self.batch), self.rot.data_ptr(), float(radius), C.byref(cn),
                               self.ws.data_ptr(), ws_bytes, C.byref(self.graph), _stream()), "trajsde_graph_prepare")
            ews_bytes = L.trajsde_graph_edges_ws_bytes(C.byref(self.batch), C.byref(self.graph))
            self.edges_ws = torch.empty(ews_bytes, device=dev, dtype=torch.uint8)
            _lib.check(L.trajsde_graph_compact(C.byref(self.batch), self.rot.data_ptr(), self.ws.data_ptr(), ws_bytes,
                                               self.edges_ws.data_ptr(), ews_bytes, C.byref(self.graph), _stream()),
                       "trajsde_graph_compact")

    def _i32(self, buf: torch.Tensor, ptr: Optional[int], n: int) -> torch.Tensor:
        """int32 view of `n` entries at device address `ptr` inside the workspace tensor `buf`"""
        if not ptr or n <= 0:
            return torch.empty(0, dtype=torch.int32, device=self.device)
        off = ptr - buf.data_ptr()
        assert 0 <= off and off + 4 * n <= buf.numel() and off % 4 == 0
        return buf[off:off + 4 * n].view(torch.int32)

    def true_counts(self) -> Dict[str, int]:
        """the list lengths (a host synchronisation when the graph was built sync-free)"""
        g = self.graph
        if g.exact:
            return {"E_aa": g.E_aa, "E_g": g.E_g, "E_la": g.E_la}
        st = getattr(self, "count_stream", None)
        if st is not None:                    # built on a side stream (prefetch_graph): read there -- a read on the current stream
            with torch.cuda.stream(st):       # would wait for everything queued on it, i.e. for the whole previous training step
                c = self._i32(self.ws, g.counts, 4).tolist()
        else:
            c = self._i32(self.ws, g.counts, 4).tolist()
        return {"E_aa": c[1], "E_g": c[2], "E_la": c[3]}

    def make_exact(self) -> "GraphContext":
        """turn a sync-free graph into an exact one in place: read the counts (synchronises) and store them in the E_* fields.
        The buffers keep their bound-sized allocation; every entry point accepts the graph afterwards."""
        g = self.graph
        if not g.exact:
            c = self.true_counts()
            assert c["E_aa"] <= g.E_aa and c["E_g"] <= g.E_g and c["E_la"] <= g.E_la, "edge-count bound violated"
            g.E_aa, g.E_g, g.E_la, g.exact = c["E_aa"], c["E_g"], c["E_la"], 1
        return self

    def edge_lists(self) -> Dict[str, torch.Tensor]:
        """the compacted lists as tensors (copies): what ENC:107-118, AGG:41 and ENC:198 leave of the input edge lists, in
        this build's canonical order (target-major, senders ascending) -- for edge-for-edge comparison with the oracle"""
        self.make_exact()
        g, b = self.graph, self.batch
        n_aa = b.H * g.Nt
        return {"aa_src": self._i32(self.edges_ws, g.aa_src, g.E_aa).clone(), "aa_dst": self._i32(self.edges_ws, g.aa_dst, g.E_aa).clone(),
                "aa_segptr": self._i32(self.ws, g.aa_segptr, n_aa + 1).clone(),
                "g_src": self._i32(self.edges_ws, g.g_src, g.E_g).clone(), "g_dst": self._i32(self.edges_ws, g.g_dst, g.E_g).clone(),
                "g_segptr": self._i32(self.ws, g.g_segptr, b.N + 1).clone(),
                "la_lane": self._i32(self.edges_ws, g.la_lane, g.E_la).clone(), "la_dst": self._i32(self.edges_ws, g.la_dst, g.E_la).clone(),
                "la_segptr": self._i32(self.ws, g.la_segptr, b.N + 1).clone()}

    @staticmethod
    def _input_stamp(data) -> tuple:
        """identity AND in-place version of every batch tensor the graph is derived from: an edited batch never reuses a
        stale graph, even under a repeated noise seed"""
        out = []
        for k in ("x", "positions", "padding_mask", "bos_mask", "rotate_angles", "edge_index", "agent_index", "batch", "source",
                  "lane_positions", "lane_paddings", "lane_actor_index", "lane_actor_vectors", "rotate_mat"):
            t = data[k] if k in data else None
            if not torch.is_tensor(t):
                out.append(None)
                continue
            # inference tensors (a batch moved to the device under torch.inference_mode(), as Lightning's validate / test loops
            # do) track no version counter -- reading it raises.  They are stamped by identity alone: the encoder rebuilds the
            # graph whenever the noise key changes, i.e. on every forward with fresh noise, and the later stages of the SAME
            # forward reuse it.
            ver = -1 if t.is_inference() else t._version
            out.append((t.data_ptr(), ver, tuple(t.shape)))
        return tuple(out)

    @classmethod
    def get(cls, data, radius: Optional[float], H: int, noise: Optional[NoiseSpec], fake_agents: bool = True,
            want_senders: bool = False, exact: Optional[bool] = True, sync_free_cap: Optional[int] = None) -> "GraphContext":
        """`exact`: True -- the caller's entry point needs the list lengths on the host (training, backward, OOD, vanilla
        variant, captures); False -- build sync-free if a build is needed; None -- take whatever the encoder left."""
        gc = data[cls.KEY] if cls.KEY in data else None
        if gc is not None and want_senders and not gc.graph.aa_src and gc.graph.E_aa > 0:
            gc = None                                                     # built without the sender ids: rebuild
        key = None
        if radius is not None and noise is not None:
            # a device-resident key (NoiseSpec.seed_dev) can change without the host seeing it, and the fake agents' rows are
            # drawn from it: such a graph is never reused across encoder calls (a fresh token per build request)
            dev_key = None if noise.seed_dev is None else next(cls._DEVICE_KEY_TOKENS)
            key = (float(radius), int(noise.seed), dev_key, id(noise.z_fake), id(noise.fake_row_ids), bool(fake_agents),
                   cls._input_stamp(data))
        elif gc is not None and getattr(gc, "build_key", None) is not None and gc.build_key[-1] != cls._input_stamp(data):
            gc = None                                                     # the batch was edited since the encoder built the graph
        if gc is None or (key is not None and getattr(gc, "build_key", None) != key):
            # the encoder (which owns the radius and the fake-agent noise) builds; the aggregator and the backward
            # entry points of the same step (same noise) reuse
            gc = cls(data, cls.DEFAULT_RADIUS if radius is None else radius, H, noise, fake_agents, exact=exact is not False,
                     sync_free_cap=sync_free_cap)
            gc.build_key = key
            data[cls.KEY] = gc
        if exact:
            gc.make_exact()
        return gc
