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
    assert lib.trajsde_abi_version() == _lib.ABI_VERSION == 9


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
    assert "-fno-slp-vectorize" in build.FLAGS
    out = tmp_path / "node_bwd.s"
    src = os.path.join(H.ROOT, "trajsde_amd", "csrc", "node_bwd.hip")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.check_call([build.HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), src], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN4tsde\w+):.*?; ScratchSize: (\d+)", text, flags=re.S | re.M)
    assert len(kernels) >= 10
    spilled = [(name[:60], int(sz)) for name, sz in kernels if int(sz) > 0]
    assert not spilled, spilled
