"""Child of tests/test_cabi_cpu.py::test_host_side_under_address_and_ub_sanitizers: drives the HOST side of the C-ABI library --
parameter tables, blob sizes, workspace-size queries, argument checks, job tables and workspace carving of the entry points -- of
a build whose host code runs under AddressSanitizer + UndefinedBehaviorSanitizer (trajsde_amd.build.build_sanitized), on a box
with no GPU.  Launches fail there (no device) and the entry points must turn that into an error status: a sanitizer report or a
crash fails the parent test, an error return does not.  TRAJSDE_LIB names the library; run with LD_PRELOAD=<asan runtime>."""
import ctypes as C
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajsde_amd import _lib  # noqa: E402

L = _lib.lib()
calls = errors = 0


def rc(v):
    """count a call; an error status must come with a message"""
    global calls, errors
    calls += 1
    if isinstance(v, int) and v < 0:
        errors += 1
        assert L.trajsde_last_error() is not None
    return v


assert L.trajsde_abi_version() == _lib.ABI_VERSION
rc(L.trajsde_split_products())
for r in (0.0, 1e-3, 1.0, 50.0, 49.999996, 1e6, 3.0e38):
    t = L.trajsde_radius2_threshold(C.c_float(r))
    assert t == t
# parameter tables and blob sizes of every stage, incl. out-of-range stage ids, indices and sizes
n_names = 0
for stage in range(-1, 13):
    for nl, K in ((0, 0), (1, 1), (3, 6), (3, 10), (4, 20), (-1, -1), (100, 1000)):
        n = rc(L.trajsde_param_count(stage, nl, K))
        rc(L.trajsde_blob_floats(stage, nl, K))
        if 0 < n < 10_000 and nl <= 8:
            for i in (-1, 0, n - 1, n, n + 5):
                s = L.trajsde_param_name(stage, i, nl, K)
                calls += 1
                n_names += s is not None
assert n_names > 100, n_names

FAKE = 0x7F0000000000            # never dereferenced on the host: the entry points only do arithmetic on device addresses


def fake(i):
    return FAKE + (i << 24)


def batch(N, A, E, Ln, Eal, H=21, TT=41, pts=10, null=()):
    ptrs = [None if k in null else fake(k + 1) for k in range(13)]
    return _lib.Batch(N, A, E, Ln, Eal, H, TT, pts, *ptrs)


def graph(b, E_aa, E_g, E_la, exact=1):
    g = _lib.Graph()
    g.Nt, g.E_ext, g.E_aa, g.E_g, g.E_la, g.exact = b.N + b.A, b.E + b.A, E_aa, E_g, E_la, exact
    for k, (name, _t) in enumerate(_lib.Graph._fields_):
        if _t is C.c_void_p:
            setattr(g, name, fake(40 + k))
    return g


shapes = [(0, 0, 0, 0, 0), (1, 1, 0, 0, 0), (16, 2, 16 * 15, 6, 40), (257, 3, 257 * 40, 100, 3000), (8192, 64, 8192 * 127, 4096, 500_000),
          (32 * 1024, 32, 2_000_000_000, 8192, 2_000_000), (2 ** 31 - 1, 1, 2 ** 31 - 1, 2 ** 31 - 1, 2 ** 31 - 1), (-5, -1, -7, -1, -1)]
ws = C.c_void_p(fake(100))
stream = C.c_void_p(0)
nz = _lib.Noise(C.c_uint64(7), None, None, None)
for N, A, E, Ln, Eal in shapes:
    b = batch(N, A, E, Ln, Eal)
    gb = rc(L.trajsde_graph_ws_bytes(C.byref(b)))
    for exact in (1, 0):
        g = graph(b, min(max(E, 0) * 3, 2 ** 31 - 1), max(E, 0), max(Eal, 0), exact)
        rc(L.trajsde_graph_edges_ws_bytes(C.byref(b), C.byref(g)))
        sizes = {
            "enc": rc(L.trajsde_encoder_ws_bytes(C.byref(b), C.byref(g))),
            "tape": rc(L.trajsde_encoder_tape_bytes(C.byref(b), C.byref(g))),
            "ebw": rc(L.trajsde_encoder_backward_ws_bytes(C.byref(b), C.byref(g))),
            "esc": rc(L.trajsde_encoder_backward_scratch_bytes(C.byref(b), C.byref(g))),
            "ood": rc(L.trajsde_encoder_ood_ws_bytes(C.byref(b), C.byref(g), 10)),
            "grid": rc(L.trajsde_encoder_grid_ws_bytes(C.byref(b), C.byref(g))),
            "gridb": rc(L.trajsde_encoder_grid_backward_ws_bytes(C.byref(b), C.byref(g), 4)),
        }
        for K in (1, 6, 20):
            sizes[f"agg{K}"] = rc(L.trajsde_aggregator_ws_bytes(C.byref(b), C.byref(g), K))
            sizes[f"aggb{K}"] = rc(L.trajsde_aggregator_backward_ws_bytes(C.byref(b), C.byref(g), 3, K))
        if not (0 < N <= 8192 and E < 2 ** 30):
            continue
        # the entry points themselves with plausible (fake) device addresses: workspace carving and launch geometry run on the
        # host; with a workspace one byte short they must refuse, with a sufficient one the first launch fails here (no device)
        tab = (C.c_float * (21 * 8))()
        for short in (True, False):
            eb = sizes["enc"] - (1 if short else 0)
            rc(L.trajsde_encoder_forward(C.byref(b), C.byref(g), fake(1), fake(2), C.cast(tab, C.c_void_p), C.byref(nz), ws, eb, fake(3), fake(4),
                                         None, None, None, stream))
            rc(L.trajsde_encoder_forward_train(C.byref(b), C.byref(g), fake(1), fake(2), C.cast(tab, C.c_void_p), C.byref(nz), ws,
                                               sizes["tape"] - (1 if short else 0), fake(3), fake(4), None, stream))
            ab = sizes["agg6"] - (1 if short else 0)
            rc(L.trajsde_aggregator_forward_heads(C.byref(b), C.byref(g), fake(2), 3, 6, 8, fake(3), ws, ab, fake(4), None, stream))
            rc(L.trajsde_aggregator_forward_train(C.byref(b), C.byref(g), fake(2), 3, 6, 8, fake(3), ws, sizes["aggb6"] - (1 if short else 0),
                                                  fake(4), None, stream))
            rc(L.trajsde_encoder_forward_ood(C.byref(b), C.byref(g), fake(1), fake(2), C.cast(tab, C.c_void_p), C.byref(nz), 10, ws,
                                             sizes["ood"] - (1 if short else 0), fake(3), fake(4), stream))
        if exact and gb > 0:
            gout = _lib.Graph()
            rc(L.trajsde_graph_prepare(C.byref(b), fake(1), C.c_float(50.0), C.byref(nz), ws, gb - 1, C.byref(gout), stream))
            rc(L.trajsde_graph_prepare_async(C.byref(b), fake(1), C.c_float(50.0), C.byref(nz), ws, gb, C.byref(gout), stream))
# decoder-side size queries and entry points
for N, K, T, ne in itertools.product((0, 1, 17, 8192, 2 ** 31 - 1, -3), (1, 6, 20, 0), (5, 20, 60), (5, 21, 61)):
    d = rc(L.trajsde_decoder_ws_bytes(N, K))
    rc(L.trajsde_decoder_backward_ws_bytes(N, K, T, ne))
    rc(L.trajsde_decoder_nll_backward_ws_bytes(N, K, T, ne))
    rc(L.trajsde_mlp_decoder_ws_bytes(N, K))
    rc(L.trajsde_mlp_decoder_backward_ws_bytes(N))
    if 0 < N <= 8192 and K > 0 and d > 0:
        for nbytes in (d - 1, d):
            rc(L.trajsde_decoder_forward(N, K, T, fake(2), fake(3), fake(4), fake(5), ne, fake(6), C.c_float(1e-3), C.byref(nz), ws, nbytes,
                                         fake(7), fake(8), stream))
# weight packing: the by-value job tables are built on the host from the parameter table
for stage in range(0, 11):
    for nl, K in ((3, 6), (4, 10)):
        n = L.trajsde_param_count(stage, nl, K)
        nf = L.trajsde_blob_floats(stage, nl, K)
        if n <= 0 or nf <= 0:
            continue
        arr = (C.c_void_p * n)(*[fake(200 + i) for i in range(n)])
        for n_given, floats in ((n, nf), (n - 1, nf), (n, nf - 1), (n + 1, nf), (0, nf)):
            rc(L.trajsde_pack_weights(stage, nl, K, arr, n_given, fake(9), floats, stream))
        arr[n // 2] = None
        rc(L.trajsde_pack_weights(stage, nl, K, arr, n, fake(9), nf, stream))
# the merged table of several stages (host side: plans, re-basing, the table written to the caller's pinned buffer)
sets = [[(0, 3, 6), (5, 3, 6), (1, 3, 6), (4, 3, 6), (2, 3, 6), (3, 3, 6)], [(0, 3, 6), (1, 4, 10), (10, 3, 6)], [(6, 3, 6), (7, 30, 6)], [(99, 1, 1)]]
for entries in sets:
    items = (_lib.PackItem * len(entries))()
    keep = []
    for k, (it, (stage, nl, K)) in enumerate(zip(items, entries)):
        n = max(L.trajsde_param_count(stage, nl, K), 0)
        arr = (C.c_void_p * max(n, 1))(*[fake(300 + 200 * k + i) for i in range(max(n, 1))])
        keep.append(arr)
        it.stage, it.num_layers, it.num_modes, it.n_params = stage, nl, K, n
        it.params, it.blob, it.blob_floats = C.cast(arr, C.c_void_p), fake(40 + k), max(L.trajsde_blob_floats(stage, nl, K), 0)
    need = rc(L.trajsde_pack_many_table_bytes(items, len(entries)))
    host = C.create_string_buffer(max(need, 64))
    for nbytes, fresh in ((max(need, 0), 1), (max(need, 0), 0), (max(need - 1, 0), 0)):
        rc(L.trajsde_pack_weights_many(items, len(entries), host, fake(60), nbytes, fresh, stream))
    if len(entries) > 1:
        items[1].blob = items[0].blob                      # two items on one blob: refused
        rc(L.trajsde_pack_weights_many(items, len(entries), host, fake(60), max(need, 0), 0, stream))
rc(L.trajsde_pack_many_table_bytes(None, 1))
rc(L.trajsde_pack_weights_many(None, 1, None, None, 0, 0, stream))
# the step's closing launches: argument checks
gi = (_lib.GatherItem * 9)()
for k in range(9):
    gi[k].dst, gi[k].src, gi[k].index, gi[k].n, gi[k].mult = fake(70), fake(71), fake(72), 100 * k, 1.0
for n_items in (0, 1, 8, 9):
    rc(L.trajsde_grad_gather_add(gi, n_items, None, stream))
gi[1].n = -1
rc(L.trajsde_grad_gather_add(gi, 2, fake(73), stream))
rc(L.trajsde_grad_gather_add(None, 1, None, stream))
for n_el, b2 in ((0, 31.6), (1000, 31.6), (1001, 0.0), (1001, float('inf')), (-1, 31.6)):
    rc(L.trajsde_adamw_step(fake(80), fake(81), fake(82), fake(83), n_el, 0.99, 0.1, 0.999, 0.001, b2, n_el & 1, 1e-8, -1e-3, stream))
rc(L.trajsde_adamw_step(None, fake(81), fake(82), fake(83), 10, 0.99, 0.1, 0.999, 0.001, 31.6, 0, 1e-8, -1e-3, stream))
rc(L.trajsde_rotate(fake(1), 100, None, 0, fake(2), None, stream))
rc(L.trajsde_rotate(None, 100, fake(3), 20, fake(2), fake(4), stream))
rc(L.trajsde_sde_step(0, fake(1), fake(2), fake(3), None, 0, C.byref(nz), stream))
rc(L.trajsde_profile_mode(2))
buf = C.create_string_buffer(1 << 12)
rc(L.trajsde_profile_report(buf, len(buf)))
rc(L.trajsde_profile_report(buf, 0))
rc(L.trajsde_profile_mode(0))
rc(L.trajsde_state_storage(0))
rc(L.trajsde_export_senders(0))
print(json.dumps({"calls": calls, "errors": errors, "param_names": n_names}))
