"""ctypes binding of libtrajsde_hip.so (include/trajsde_hip.h).  There is no fallback: if the HIP library
is missing or fails to load, every stage raises."""
import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRAJSDE_LIB") or os.path.join(HERE, "libtrajsde_hip.so")   # TRAJSDE_LIB: another build of the same ABI (A/B runs)
# the alternative kernel forms kept as cross-checks (TRAJSDE_EDGE_TILE=32, TRAJSDE_EDGE_PIPE, TRAJSDE_FUSED_TILES, TRAJSDE_GATTN_MM):
# a second library of the same ABI that the tests selecting them load through TRAJSDE_LIB (trajsde_amd/build.py)
ALT_LIB_PATH = os.path.join(HERE, "variants", "libtrajsde_alt.so")

STAGE_ENCODER, STAGE_AGGREGATOR, STAGE_DECODER, STAGE_DECODER_BWD, STAGE_AGGREGATOR_BWD, STAGE_ENCODER_BWD = 0, 1, 2, 3, 4, 5
STAGE_ENCODER_GRID, STAGE_DECODER_MLP, STAGE_DECODER_MLP_BWD, STAGE_ENCODER_GRID_BWD = 6, 7, 8, 9
STAGE_DECODER_NLL_BWD = 10


ABI_VERSION = 10         # trajsde_graph grew aa_src / la_lane (2); trajsde_dropout arguments (3); training tapes (4);
                         # device-side list lengths + trajsde_graph_prepare_async (5); encoder tape / scratch split (6);
                         # trajsde_noise.seed_dev: Philox key read on the device (7); trajsde_aggregator_prepare /
                         # _forward_prepared: the relative-pose embedding as a call of its own (8);
                         # trajsde_encoder_grid_forward_train / _backward_train: dropout of the vanilla encoder (9);
                         # trajsde_pack_weights_many, trajsde_grad_gather_add, trajsde_adamw_step: a training step's small launches (10)


class TrajsdeError(RuntimeError):
    pass


class Batch(C.Structure):
    _fields_ = [("N", C.c_int32), ("A", C.c_int32), ("E", C.c_int32), ("L", C.c_int32), ("E_al", C.c_int32),
                ("H", C.c_int32), ("TT", C.c_int32), ("lane_pts", C.c_int32),
                ("x", C.c_void_p), ("positions", C.c_void_p), ("padding_mask", C.c_void_p), ("bos_mask", C.c_void_p),
                ("rotate_angles", C.c_void_p), ("edge_index", C.c_void_p), ("agent_index", C.c_void_p),
                ("batch", C.c_void_p), ("source", C.c_void_p), ("lane_positions", C.c_void_p),
                ("lane_paddings", C.c_void_p), ("lane_actor_index", C.c_void_p), ("lane_actor_vectors", C.c_void_p)]


class Noise(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("z", C.c_void_p), ("row_ids", C.c_void_p), ("seed_dev", C.c_void_p)]


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint64)]


class Graph(C.Structure):
    _fields_ = [("Nt", C.c_int32), ("E_ext", C.c_int32), ("E_aa", C.c_int32), ("E_g", C.c_int32), ("E_la", C.c_int32),
                ("orig", C.c_void_p), ("nus_mask", C.c_void_p), ("eos_idx", C.c_void_p), ("pick_slot", C.c_void_p),
                ("x_fake", C.c_void_p), ("aa_geom", C.c_void_p), ("aa_dst", C.c_void_p), ("aa_segptr", C.c_void_p),
                ("g_geom", C.c_void_p), ("g_src", C.c_void_p), ("g_dst", C.c_void_p), ("g_segptr", C.c_void_p),
                ("la_geom", C.c_void_p), ("la_dst", C.c_void_p), ("la_segptr", C.c_void_p),
                ("aa_src", C.c_void_p), ("la_lane", C.c_void_p), ("counts", C.c_void_p), ("exact", C.c_int32)]


class PackItem(C.Structure):
    _fields_ = [("stage", C.c_int32), ("num_layers", C.c_int32), ("num_modes", C.c_int32), ("n_params", C.c_int32),
                ("params", C.c_void_p), ("blob", C.c_void_p), ("blob_floats", C.c_int64)]


class GatherItem(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("index", C.c_void_p), ("n", C.c_int64), ("mult", C.c_float)]


_lib: Optional[C.CDLL] = None

# every symbol include/trajsde_hip.h declares: (restype, argtypes)
P, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "trajsde_last_error": (C.c_char_p, []),
    "trajsde_split_products": (C.c_int, []),
    "trajsde_abi_version": (C.c_int, []),
    "trajsde_export_senders": (C.c_int, [C.c_int]),
    "trajsde_state_storage": (C.c_int, [C.c_int]),
    "trajsde_range_status": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), P]),
    "trajsde_param_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trajsde_param_name": (C.c_char_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "trajsde_blob_floats": (I64, [C.c_int, C.c_int, C.c_int]),
    "trajsde_pack_weights": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(P), C.c_int, P, I64, P]),
    "trajsde_pack_many_table_bytes": (I64, [C.POINTER(PackItem), C.c_int]),
    "trajsde_pack_weights_many": (C.c_int, [C.POINTER(PackItem), C.c_int, P, P, I64, C.c_int, P]),
    "trajsde_grad_gather_add": (C.c_int, [C.POINTER(GatherItem), C.c_int, P, P]),
    "trajsde_adamw_step": (C.c_int, [P, P, P, P, I64, F32, F32, F32, F32, F32, C.c_int, F32, F32, P]),
    "trajsde_rotate": (C.c_int, [P, I32, P, I32, P, P, P]),
    "trajsde_graph_ws_bytes": (I64, [C.POINTER(Batch)]),
    "trajsde_graph_prepare": (C.c_int, [C.POINTER(Batch), P, F32, C.POINTER(Noise), P, I64, C.POINTER(Graph), P]),
    "trajsde_graph_prepare_async": (C.c_int, [C.POINTER(Batch), P, F32, C.POINTER(Noise), P, I64, C.POINTER(Graph), P]),
    "trajsde_sync_free_supported": (C.c_int, []),
    "trajsde_radius2_threshold": (C.c_float, [C.c_float]),
    "trajsde_graph_edges_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_graph_compact": (C.c_int, [C.POINTER(Batch), P, P, I64, P, I64, C.POINTER(Graph), P]),
    "trajsde_encoder_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_encoder_forward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, C.POINTER(Noise), P, I64, P, P, P, P,
                                          C.POINTER(Dropout), P]),
    "trajsde_encoder_ood_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph), C.c_int]),
    "trajsde_encoder_forward_ood": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, C.POINTER(Noise), C.c_int, P, I64, P, P, P]),
    "trajsde_aggregator_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph), C.c_int]),
    "trajsde_aggregator_forward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, C.c_int, C.c_int, P, P, I64, P, P]),
    "trajsde_decoder_ws_bytes": (I64, [I32, C.c_int]),
    "trajsde_decoder_forward": (C.c_int, [I32, C.c_int, C.c_int, P, P, P, P, C.c_int, P, F32, C.POINTER(Noise), P, I64, P, P, P]),
    "trajsde_decoder_backward_ws_bytes": (I64, [I32, C.c_int, C.c_int, C.c_int]),
    "trajsde_decoder_l2_backward": (C.c_int, [I32, C.c_int, C.c_int, P, P, P, P, P, C.c_int, P, C.POINTER(Noise), P, P, P, P, I64,
                                              P, P, C.POINTER(P), C.c_int, P, P, P]),
    "trajsde_decoder_nll_backward_ws_bytes": (I64, [I32, C.c_int, C.c_int, C.c_int]),
    "trajsde_decoder_nll_backward": (C.c_int, [I32, C.c_int, C.c_int, P, P, P, P, P, C.c_int, P, C.POINTER(Noise), P, P, P, F32, F32, P, I64,
                                               P, P, C.POINTER(P), C.c_int, P, P, P]),
    "trajsde_aggregator_backward_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph), C.c_int, C.c_int]),
    "trajsde_aggregator_backward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, C.c_int, C.c_int, P, P, P, I64,
                                              C.POINTER(P), C.c_int, P, P]),
    "trajsde_aggregator_backward_heads": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, C.c_int, C.c_int, C.c_int, P, P, P, I64,
                                                    C.POINTER(P), C.c_int, P, C.POINTER(Dropout), C.c_int, P]),
    "trajsde_aggregator_forward_train": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, C.c_int, C.c_int, C.c_int, P, P, I64, P,
                                                   C.POINTER(Dropout), P]),
    "trajsde_encoder_backward_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_encoder_tape_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_encoder_backward_scratch_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_encoder_backward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, P, P, C.POINTER(Noise), P, F32, P, I64, P,
                                           C.POINTER(P), C.c_int, P, P, C.POINTER(Dropout), C.c_int, P, I64, P]),
    "trajsde_encoder_forward_train": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, C.POINTER(Noise), P, I64, P, P,
                                                C.POINTER(Dropout), P]),
    "trajsde_aggregator_forward_heads": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, C.c_int, C.c_int, C.c_int, P, P, I64, P,
                                                   C.POINTER(Dropout), P]),
    "trajsde_aggregator_prepare": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, I64, P]),
    "trajsde_encoder_fork_stream": (C.c_int, [P]),
    "trajsde_aggregator_forward_prepared": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, C.c_int, C.c_int, C.c_int, P, P, I64, P,
                                                      C.POINTER(Dropout), P]),
    "trajsde_encoder_grid_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph)]),
    "trajsde_encoder_grid_forward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, C.c_int, C.c_int, P, I64, P, P]),
    "trajsde_encoder_grid_forward_train": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, C.c_int, C.c_int, P, I64, P,
                                                     C.POINTER(Dropout), P]),
    "trajsde_mlp_decoder_ws_bytes": (I64, [I32, C.c_int]),
    "trajsde_mlp_decoder_forward": (C.c_int, [I32, C.c_int, C.c_int, P, P, P, F32, P, I64, P, P, P]),
    "trajsde_mlp_decoder_backward_ws_bytes": (I64, [I32]),
    "trajsde_mlp_decoder_l2_backward": (C.c_int, [I32, C.c_int, C.c_int, P, P, P, P, P, P, P, I64, P, P, C.POINTER(P), C.c_int, P, P, P]),
    "trajsde_encoder_grid_backward_ws_bytes": (I64, [C.POINTER(Batch), C.POINTER(Graph), C.c_int]),
    "trajsde_encoder_grid_backward": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, C.c_int, C.c_int, P, P, I64, C.POINTER(P),
                                                C.c_int, P]),
    "trajsde_encoder_grid_backward_train": (C.c_int, [C.POINTER(Batch), C.POINTER(Graph), P, P, P, C.c_int, C.c_int, P, P, I64, C.POINTER(P),
                                                      C.c_int, C.POINTER(Dropout), P]),
    "trajsde_profile_mode": (C.c_int, [C.c_int]),
    "trajsde_profile_report": (I64, [C.c_char_p, I64]),
    "trajsde_sde_step": (C.c_int, [I32, P, P, P, C.POINTER(F32), C.c_int, C.POINTER(Noise), P]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise TrajsdeError(f"{LIB_PATH} is missing: build it with `python -m trajsde_amd.build` "
                               "(there is no CPU/PyTorch fallback for the hot path)")
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch's first so that this library
        # binds to the runtime that owns torch's device context and streams; loaded the other way round the process
        # ends up with two HIP runtimes and ours reports "no ROCm-capable device".
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if handle.trajsde_abi_version() != ABI_VERSION:
            raise TrajsdeError(f"{LIB_PATH} has ABI version {handle.trajsde_abi_version()}, this binding expects {ABI_VERSION}: "
                               "rebuild with `python -m trajsde_amd.build --force`")
        _lib = handle
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().trajsde_last_error().decode()
        raise TrajsdeError(f"{what or 'trajsde call'} failed ({status}): {msg}")


def check_range(stream: Optional[int] = None, reset: bool = True) -> None:
    """raise TrajsdeError if a launch since the last check fed a magnitude >= 65504 to an fp16x3 product (csrc/range.hpp);
    synchronises `stream` (default: torch's current stream)"""
    import torch
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    check(lib().trajsde_range_status(1 if reset else 0, None, st), "trajsde_range_status")


def profile_report() -> dict:
    """{tag: (launches, total_ms, dominant)} of the events recorded since the last report."""
    L = lib()
    buf = C.create_string_buffer(1 << 16)
    L.trajsde_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        tag, n, ms, dom = line.rsplit(" ", 3)
        out[tag] = (int(n), float(ms), bool(int(dom)))
    return out
