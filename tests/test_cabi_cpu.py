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
    assert lib.trajsde_abi_version() == 1


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
