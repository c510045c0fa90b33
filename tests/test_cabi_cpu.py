"""CPU suite: the C-ABI library loads here (no GPU) and exports every symbol include/trajsde_hip.h declares;
parameter tables agree with the Python modules' state_dict.  No compute calls."""
import os
import re

import helpers as H


def test_library_exports_every_declared_symbol(repo_root):
    from trajsde_amd import _lib, build
    build.build(verbose=False)
    header = open(os.path.join(repo_root, "include", "trajsde_hip.h")).read()
    declared = set(re.findall(r"\b(trajsde_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.trajsde_abi_version() == _lib.ABI_VERSION == 10
    # the library of the alternative kernel forms (tests and A/B tools load it through TRAJSDE_LIB) speaks the same ABI
    import ctypes
    alt = ctypes.CDLL(_lib.ALT_LIB_PATH)
    for name in declared:
        assert hasattr(alt, name), name
    assert alt.trajsde_abi_version() == _lib.ABI_VERSION
    assert os.path.getsize(_lib.LIB_PATH) < os.path.getsize(_lib.ALT_LIB_PATH)      # the product library carries one form of every kernel


# ---- prototype-level check: names alone would let an argument added on one side only slip through ----------
_STRUCTS = {"trajsde_batch": "Batch", "trajsde_graph": "Graph", "trajsde_noise": "Noise", "trajsde_dropout": "Dropout",
            "trajsde_pack_item": "PackItem", "trajsde_gather_item": "GatherItem"}


def _header_prototypes(text):
    """{name: (return type, [parameter type strings])} of every `trajsde_*` function the header declares (comments stripped)"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 ]*?[ \*]+)\b(trajsde_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?[\* ])([A-Za-z_][A-Za-z_0-9]*)$", a)                 # type, then the parameter's name
                params.append((mm.group(1) if mm else a).strip())
        protos[name] = (ret, params)
    return protos


def _ctypes_class(t):
    """what a ctypes entry says about an argument: ('struct', name) | ('ptr',) | ('scalar', C spelling)"""
    import ctypes as C
    from trajsde_amd import _lib
    scal = {C.c_int: "int", C.c_int32: "int32_t", C.c_int64: "int64_t", C.c_float: "float", C.c_uint64: "uint64_t",
            C.c_uint32: "uint32_t"}
    if C.c_int32 is C.c_int:                       # same object on this platform: `int` and `int32_t` are one C type as well
        scal[C.c_int] = "int"
    if t in (C.c_void_p, C.c_char_p):
        return ("ptr",)
    if t in scal:
        return ("scalar", scal[t])
    if hasattr(t, "_type_"):                       # POINTER(x)
        inner = t._type_
        for cname, pyname in _STRUCTS.items():
            if inner is getattr(_lib, pyname):
                return ("struct", cname)
        return ("ptr",)
    raise AssertionError(f"unmapped ctypes type {t}")


def _check_against_header(signatures, protos):
    """Python-side comparison: argument COUNTS, scalar-vs-pointer classes and which struct a struct pointer points at."""
    problems = []
    for name, (res, args) in signatures.items():
        ret, params = protos[name]
        if len(params) != len(args):
            problems.append(f"{name}: header has {len(params)} parameters, ctypes {len(args)}")
            continue
        for i, (ptype, ctype) in enumerate(zip(params, args)):
            cls = _ctypes_class(ctype)
            is_ptr = "*" in ptype
            base = re.sub(r"\bconst\b|\*", " ", ptype).split()
            base = base[0] if base else ""
            if cls[0] == "scalar":
                same_int = {cls[1], base} <= {"int", "int32_t"}
                if is_ptr or (base != cls[1] and not same_int):
                    problems.append(f"{name} arg {i}: header `{ptype}`, ctypes {cls[1]}")
            elif cls[0] == "struct":
                if not is_ptr or base != cls[1]:
                    problems.append(f"{name} arg {i}: header `{ptype}`, ctypes POINTER({cls[1]})")
            else:
                if not is_ptr or base in _STRUCTS:
                    problems.append(f"{name} arg {i}: header `{ptype}`, ctypes void pointer")
    return problems


def _prototype_tu(signatures, protos):
    """A C translation unit that includes the header and assigns every entry point to a function pointer of the type the ctypes
    table implies: scalars spelled from the ctypes entry, pointers spelled as the header spells them (ctypes cannot say more
    about a pointer than that it is one).  gcc -Werror=incompatible-pointer-types then rejects a differing count or scalar."""
    import ctypes as C
    out = ['#include "trajsde_hip.h"', ""]
    ret_of = {C.c_char_p: "const char*", C.c_int: "int", C.c_int64: "int64_t", C.c_float: "float"}
    for name, (res, args) in signatures.items():
        _, params = protos[name]
        spelled = []
        for i, ctype in enumerate(args):
            cls = _ctypes_class(ctype)
            if cls[0] == "scalar":
                spelled.append(cls[1])
            elif cls[0] == "struct":
                hdr = params[i] if i < len(params) else ""
                spelled.append(("const " if "const" in hdr else "") + cls[1] + "*")
            else:
                spelled.append(params[i] if i < len(params) and "*" in params[i] else "void*")
        out.append(f"static {ret_of[res]} (*const check_{name})({', '.join(spelled) or 'void'}) = {name};")
    out.append("int main(void) { return 0; }")
    return "\n".join(out) + "\n"


def _compile_tu(tu_text, tmp_path, repo_root, tag):
    import subprocess
    src = tmp_path / f"abi_{tag}.c"
    src.write_text(tu_text)
    return subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
                           "-Wno-unused-variable", "-I", os.path.join(repo_root, "include"), str(src)],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_ctypes_signatures_match_the_header_prototypes(repo_root, tmp_path):
    """every `_lib.SIGNATURES` entry against the PROTOTYPE in include/trajsde_hip.h: argument count, scalar types
    (int / int32_t / int64_t / float ...), pointer-ness and pointed-to struct -- compared in Python and by the C compiler"""
    from trajsde_amd import _lib
    header = open(os.path.join(repo_root, "include", "trajsde_hip.h")).read()
    protos = _header_prototypes(header)
    assert set(protos) == set(_lib.SIGNATURES)
    assert _check_against_header(_lib.SIGNATURES, protos) == []
    r = _compile_tu(_prototype_tu(_lib.SIGNATURES, protos), tmp_path, repo_root, "ok")
    assert r.returncode == 0, r.stdout[-3000:]


def test_the_prototype_check_fails_when_one_side_changes(repo_root, tmp_path):
    """teeth of the check above: an argument added, dropped or retyped on the ctypes side only must be caught by BOTH legs"""
    import ctypes as C
    from trajsde_amd import _lib
    header = open(os.path.join(repo_root, "include", "trajsde_hip.h")).read()
    protos = _header_prototypes(header)
    res, args = _lib.SIGNATURES["trajsde_decoder_forward"]
    edits = {"extra argument": list(args) + [C.c_int], "dropped argument": list(args)[:-1],
             "int64 where the header says float": [C.c_int64 if a is C.c_float else a for a in args],
             "int32 where the header says int64": [C.c_int32 if a is C.c_int64 else a for a in args]}
    for what, new_args in edits.items():
        sig = dict(_lib.SIGNATURES)
        sig["trajsde_decoder_forward"] = (res, new_args)
        assert _check_against_header(sig, protos), what
        assert _compile_tu(_prototype_tu(sig, protos), tmp_path, repo_root, "bad").returncode != 0, what
    sig = dict(_lib.SIGNATURES)                    # a struct pointer of the wrong struct (both are pointers: only the class check sees it)
    r2, a2 = sig["trajsde_encoder_ws_bytes"]
    sig["trajsde_encoder_ws_bytes"] = (r2, [a2[1], a2[0]])
    assert _check_against_header(sig, protos)
    assert _compile_tu(_prototype_tu(sig, protos), tmp_path, repo_root, "bad2").returncode != 0


def test_param_tables_match_state_dict():
    from trajsde_amd import _lib
    lib = _lib.lib()
    model, cfg = H.build_model(6, 20, 2.0)
    used = 0
    for stage, mod in ((0, model.encoder), (1, model.aggregator), (2, model.decoder)):
        nl, K = int(getattr(mod, "num_layers", 0)), int(getattr(mod, "num_modes", 0))
        names = [lib.trajsde_param_name(stage, i, nl, K).decode() for i in range(lib.trajsde_param_count(stage, nl, K))]
        sd = dict(mod.named_parameters())
        assert len(set(names)) == len(names)
        for n in names:
            assert n in sd, n
        used += len(names)
        assert lib.trajsde_blob_floats(stage, nl, K) > sum(sd[n].numel() for n in names) * 0.9
    # every parameter the kernels need is named; the unused ones are exactly the reference's dead weights
    all_names = {k for k, _ in model.named_parameters()}
    dead = {"encoder.al_encoder.is_intersection_embed", "encoder.al_encoder.turn_direction_embed",
            "encoder.al_encoder.traffic_control_embed", "decoder.hidden", "encoder.lsde_func.h_func.theta",
            "encoder.lsde_func.h_func.mu", "decoder.lsde_func.h_func.theta", "decoder.lsde_func.h_func.mu"}
    assert used == len(all_names) - len(dead)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest
    from trajsde_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TrajsdeError):
        _lib.lib()


def test_param_tables_of_backward_and_vanilla_stages():
    """every *_BWD stage names a subset of its stage's parameters; the vanilla-variant stages cover theirs"""
    import yaml
    import helpers as H
    from trajsde_amd import _lib
    from trajsde_amd.models.model_base_mix import PredictionModel
    lib = _lib.lib()
    model, _ = H.build_model(6, 20, 2.0)

    def names(stage, nl, K):
        return [lib.trajsde_param_name(stage, i, nl, K).decode() for i in range(lib.trajsde_param_count(stage, nl, K))]

    for stage, mod, nl, K in ((_lib.STAGE_DECODER_BWD, model.decoder, 0, 6), (_lib.STAGE_AGGREGATOR_BWD, model.aggregator, 3, 6),
                              (_lib.STAGE_ENCODER_BWD, model.encoder, 0, 0)):
        sd = dict(mod.named_parameters())
        got = names(stage, nl, K)
        assert len(set(got)) == len(got) and all(n in sd for n in got)
    # the aggregator and encoder losses reach everything the forward uses; the decoder's pi / scale heads are not reached
    assert set(names(_lib.STAGE_AGGREGATOR_BWD, 3, 6)) == set(names(_lib.STAGE_AGGREGATOR, 3, 6))
    assert set(names(_lib.STAGE_ENCODER_BWD, 0, 0)) == set(names(_lib.STAGE_ENCODER, 0, 0))
    missing = set(names(_lib.STAGE_DECODER, 0, 6)) - set(names(_lib.STAGE_DECODER_BWD, 0, 6))
    assert missing and all(n.startswith("pi.") or n.startswith("scale.") for n in missing)

    with open(os.path.join(H.ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        cfg = yaml.safe_load(f)
    van = PredictionModel(**cfg, init_seed=0)
    enc = dict(van.encoder.named_parameters())
    got = names(_lib.STAGE_ENCODER_GRID, 4, 0)
    assert all(n in enc for n in got)
    assert set(enc) - set(got) == {"al_encoder.is_intersection_embed", "al_encoder.turn_direction_embed", "al_encoder.traffic_control_embed"}
    dec = dict(van.decoder.named_parameters())
    assert set(names(_lib.STAGE_DECODER_MLP, 60, 10)) == set(dec)
    assert lib.trajsde_blob_floats(_lib.STAGE_ENCODER_GRID, 4, 0) > lib.trajsde_blob_floats(_lib.STAGE_ENCODER, 0, 0)


def test_build_hook_post_check_passes_on_the_built_library():
    """__graft_entry__.build()'s own check after compiling (it once compared against a stale ABI literal)"""
    import __graft_entry__ as g
    g.check_loaded()


def test_backward_tile_kernels_compile_without_spills_and_with_the_correctness_flag(tmp_path):
    """The builds whose results were not bit-reproducible were built with the SLP vectoriser on, and the kernels
    that misbehaved -- the embedding-backward tile kernels of csrc/node_bwd.hip -- are exactly the ones that then spill registers.  The flag
    must stay in the build, and every kernel of that file must keep compiling without scratch memory."""
    import subprocess
    from trajsde_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS and "-DTSDE_NO_SLP=1" in build.FLAGS
    out = tmp_path / "node_bwd.s"
    src = os.path.join(H.ROOT, "trajsde_amd", "csrc", "node_bwd.hip")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.check_call([build.HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), src], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN4tsde\w+):.*?; ScratchSize: (\d+)", text, flags=re.S | re.M)
    assert len(kernels) >= 10
    spilled = [(name[:60], int(sz)) for name, sz in kernels if int(sz) > 0]
    assert not spilled, spilled


def test_shipped_isa_carries_no_slp_packed_fp32_arithmetic(tmp_path):
    """ISA guard for DESIGN section 5 item 8.  With the SLP vectoriser on, the compiler pairs the fp32 arithmetic around
    the matrix products into v_pk_fma_f32 on register pairs it shuffles together with v_pk_mov_b32, and identical launches of the
    backward tile kernels then disagree in their low-order bits.  The mechanism is not pinned down, so the build is fenced three
    ways: the flag (-fno-slp-vectorize), the macro the sources demand (-DTSDE_NO_SLP=1, csrc/tile.hpp #error), and THIS check of
    what the compiler actually emitted: no translation unit with matrix instructions may contain v_pk_mov_b32 (the register-pair
    shuffles only the vectoriser needs), and the backward units -- where the irreproducible tiles were -- no v_pk_fma_f32 either
    (the f4 arithmetic written out in the sources compiles to v_pk_mul_f32 / v_pk_add_f32 there, which the shipped, bit-reproducible
    build does contain: 200 + 40 in node_bwd.hip, against 489 + 1099 and 850 v_pk_fma_f32 with the vectoriser on; the forward
    units hold a few dozen v_pk_fma_f32 from explicit f4 multiply-adds and are checked for bit-identical repeats on the GPU).  A
    compiler upgrade that packs this arithmetic by another route fails here instead of silently on the GPU."""
    import subprocess
    from trajsde_amd import build
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    offenders, seen_mfma = {}, 0
    procs = []
    for src in build.sources():
        name = os.path.basename(src)
        if name in ("prep.hip", "pack.hip"):                 # no matrix instructions: index work and weight packing
            continue
        out = tmp_path / (name[:-4] + ".s")
        extra = build.PER_FILE_FLAGS.get(name, [])
        procs.append((name, out, subprocess.Popen([build.HIPCC, *flags, *extra, "--cuda-device-only", "-S", "-o", str(out), src],
                                                  stderr=subprocess.DEVNULL)))
    for name, out, p in procs:
        assert p.wait() == 0, name
        text = out.read_text()
        seen_mfma += text.count("v_mfma_")
        ops = ("v_pk_fma_f32", "v_pk_mov_b32") if name.endswith("_bwd.hip") else ("v_pk_mov_b32",)
        bad = {op: text.count(op) for op in ops if text.count(op)}
        if bad:
            offenders[name] = bad
    assert seen_mfma > 3000                                   # the scan really looked at the matrix kernels
    assert not offenders, offenders
    # teeth: the same scan of one backward unit built WITH the vectoriser finds the instructions
    src = os.path.join(H.ROOT, "trajsde_amd", "csrc", "node_bwd.hip")
    out = tmp_path / "node_bwd_slp.s"
    slp_flags = [f for f in flags if f != "-fno-slp-vectorize"]
    subprocess.check_call([build.HIPCC, *slp_flags, "--cuda-device-only", "-S", "-o", str(out), src], stderr=subprocess.DEVNULL)
    assert out.read_text().count("v_pk_fma_f32") > 100


def test_host_side_under_address_and_ub_sanitizers(tmp_path, repo_root):
    """SURVEY section 5 (sanitizers) / VERDICT r4 #10: the HOST side of the library -- parameter tables, blob sizes, the `*_ws_bytes`
    queries, argument checks, the weight packer's job tables, the workspace carving and launch geometry of the 25-argument entry
    points -- built with -fsanitize=address,undefined (device code unsanitised: GPU ASan is not available on this pool) and driven
    by tests/sanitizer_child.py in a process of its own.  No GPU here: launches fail and must surface as error statuses; any
    sanitizer report aborts the child (-fno-sanitize-recover).  This build found, and the sources no longer have: pointer
    arithmetic on the null base the size queries carve from (common.hpp Carver, prep.hip PrepWs) and `E + 1` overflowing int32 at
    the bound the ABI admits (prep.hip EdgeWs)."""
    import json
    import subprocess
    import sys
    from trajsde_amd import build
    lib = build.build_sanitized(str(tmp_path / "libtrajsde_san.so"), str(tmp_path / "obj"))
    env = dict(os.environ, TRAJSDE_LIB=lib, LD_PRELOAD=build.asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(repo_root, "tests", "sanitizer_child.py")], env=env, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["calls"] > 2000 and 0 < res["errors"] < res["calls"] and res["param_names"] > 100
