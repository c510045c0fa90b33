"""CPU suite: the oracle (oracle/restate.py) against the golden vectors made from the reference, plus
known-answer tests for the third-party semantics it restates."""
import glob
import os

import numpy as np
import pytest
import torch

import helpers as H


@pytest.mark.parametrize("name", H.GOLDEN)
def test_restatement_matches_reference_golden(name):
    batch, meta, out, mid = H.load_fixture(name)
    model, cfg = H.build_model(meta)
    # the fixture stores only the init seed; make sure it regenerates the very weights the reference ran with
    assert abs(H.state_checksum(model.state_dict()) - meta["state_checksum"]) < 1e-6 * meta["state_checksum"]
    mine = H.oracle_forward(model, cfg, batch, meta["noise_seed"])
    tol = 1e-5   # restatement <= 1e-5 vs the reference (observed: 0.0, same torch ops in the same order)
    for key in ("loc", "pi", "diff_in", "diff_out", "label_in", "label_out"):
        assert H.maxdiff(mine[key], out[key]) <= tol, key
    assert torch.equal(mine["reg_mask"], out["reg_mask"])
    assert H.maxdiff(mine["y"], out["y_rot"]) <= tol
    assert H.maxdiff(mine["rotate_mat"], out["rotate_mat"]) <= tol
    for key in ("local_embed", "global_embed", "aa_out", "latent_ys"):
        assert H.maxdiff(mine[key], mid[key]) <= tol, key


def _load_ood(name="ood_k3_t5"):
    import os
    from trajsde_amd.data import TemporalData
    z = np.load(os.path.join(H.ROOT, "tests", "golden_ood", name + ".npz"))
    batch = TemporalData(**{k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")})
    batch["num_nodes"] = batch["x"].shape[0]
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta.")}
    out = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")}
    return batch, meta, out


def test_restatement_matches_reference_golden_ood():
    """MODEL:89-98 with ood=True (forward_ood, ENC:204-370), golden from the reference itself."""
    import restate
    batch, meta, out = _load_ood()
    model, cfg = H.build_model(meta)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    mine = restate.forward(P, cfg, H.clone_batch(batch), restate.PhiloxNoise(int(meta["noise_seed"])), ood=True)
    for key in ("loc", "pi", "stds"):
        assert H.maxdiff(mine[key], out[key]) <= 1e-5, key


GRID = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(H.ROOT, "tests", "golden_grid", "*.npz")))


@pytest.mark.parametrize("name", GRID)
def test_vanilla_restatement_matches_reference_golden(name):
    """oracle/restate_grid.py against the reference's vanilla HiVT model (tests/golden_grid, oracle/make_golden_grid.py).
    torch's fused multi-head attention orders its sums differently, so the pin is 1e-5 here, not bit-exact."""
    import restate_grid
    import yaml
    from trajsde_amd.data import TemporalData
    from trajsde_amd.models.model_base_mix import PredictionModel
    z = np.load(os.path.join(H.ROOT, "tests", "golden_grid", name + ".npz"))
    batch = TemporalData(**{k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")})
    with open(os.path.join(H.ROOT, "trajsde_amd/configs/mi355x_trmenc_mlpdec.yml")) as f:
        cfg = yaml.safe_load(f)
    K, T, heads, layers = (int(z["meta." + k]) for k in ("num_modes", "future_steps", "num_heads", "num_temporal_layers"))
    cfg["model_specific"]["kwargs"].update(num_modes=K, future_steps=T)
    cfg["encoder"]["kwargs"].update(num_heads=heads, num_temporal_layers=layers)
    cfg["aggregator"]["kwargs"].update(num_modes=K, num_heads=heads)
    cfg["decoder"]["kwargs"].update(num_modes=K, future_steps=T)
    if "meta.uncertain" in z.files and int(z["meta.uncertain"]) == 0:      # the decoder without its scale head (GDEC:31, :58-59)
        cfg["decoder"]["kwargs"]["uncertain"] = False
    model = PredictionModel(**cfg, init_seed=int(z["meta.init_seed"]))
    sd = model.state_dict()
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values() if torch.isfinite(v).all()) - float(z["meta.state_checksum"])) < 1e-6 * float(z["meta.state_checksum"])
    out = restate_grid.forward({k: v.detach() for k, v in sd.items()}, cfg, batch, want_intermediates=True)
    for k in ("loc", "pi"):
        assert H.maxdiff(out[k], torch.from_numpy(z["out." + k])) <= 1e-5, k
    assert torch.equal(out["reg_mask"], torch.from_numpy(z["out.reg_mask"]))
    for k in ("local_embed", "global_embed"):
        assert H.maxdiff(out[k], torch.from_numpy(z["mid." + k])) <= 1e-5, k


def test_oracle_detects_a_wrong_radius():
    """negative control: the comparison is not vacuous."""
    batch, meta, out, mid = H.load_fixture("mixed_k6_t20")
    model, cfg = H.build_model(meta)
    cfg["encoder"]["kwargs"]["local_radius"] = 20
    mine = H.oracle_forward(model, cfg, batch, meta["noise_seed"])
    assert H.maxdiff(mine["loc"], out["loc"]) > 1e-4


def test_schedule_table_matches_appendix_d():
    from trajsde_amd.schedule import decoder_schedule, encoder_schedule
    steps = {(5, 0.5): 5, (20, 2.0): 20, (30, 3.0): 31, (50, 5.0): 51, (60, 6.0): 61}
    for (T, mt), n in steps.items():
        s = decoder_schedule(T, mt)
        assert s.n_euler == n and s.n_out == T
        assert s.out_step[-1] == n and np.all(np.diff(s.out_step) >= 0)
        np.testing.assert_allclose(s.out_w0 + s.out_w1, 1.0, atol=1e-6)
    s5 = decoder_schedule(5, 0.5)
    assert np.all(s5.out_w1 == 1.0)
    s60 = decoder_schedule(60, 6.0)
    assert 1e-6 < s60.dt[-1] < 1e-5 and 1e-3 < s60.sqrt_h[-1] < 3e-3      # the noisy micro-step
    e = encoder_schedule()
    assert e.n_euler == 21 and abs(e.dt[0] - 0.01) < 1e-7 and np.all(np.abs(e.dt[1:] - 0.1) < 1e-6)
    assert np.all(e.out_w1 == 1.0) and list(e.out_step) == list(range(1, 22))


def test_philox_known_answers_and_moments():
    from trajsde_amd import philox
    # Random123 kat_vectors: the same three (counter, key) pairs at the library's default 10 rounds and at the 7 rounds this
    # build runs (csrc/philox.hpp PHILOX_ROUNDS; trajsde_amd/philox.py is its twin)
    pi_ctr, pi_key = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)
    kat = {10: [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
                ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
                (pi_ctr, pi_key, (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))],
           7: [((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
               ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
               (pi_ctr, pi_key, (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a))]}
    assert philox.PHILOX_ROUNDS == 7
    for rounds, vectors in kat.items():
        for ctr, key, want in vectors:
            got = philox.philox4x32(np.array([ctr], np.uint32), np.array(key, np.uint32), rounds=rounds)[0]
            assert tuple(int(v) for v in got) == want, (rounds, ctr)
    got = philox.philox4x32(np.array([pi_ctr], np.uint32), np.array(pi_key, np.uint32))[0]        # the default IS the 7-round one
    assert tuple(int(v) for v in got) == kat[7][2][2]
    z = philox.normals(7, philox.STREAM_DECODER, 3, np.arange(4096), 64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # keyed by global row id: a shard sees exactly the rows it owns
    np.testing.assert_array_equal(z[100:200], philox.normals(7, philox.STREAM_DECODER, 3, np.arange(100, 200), 64))


def test_segment_softmax_known_answer():
    import restate
    logits = torch.tensor([[0.0], [1.0], [2.0], [5.0]])
    dst = torch.tensor([0, 0, 2, 2])
    a = restate.segment_softmax(logits, dst, 3)
    e = torch.exp(torch.tensor([-1.0, 0.0, -3.0, 0.0]))
    want = torch.stack([e[0] / (e[0] + e[1]), e[1] / (e[0] + e[1]), e[2] / (e[2] + e[3]), e[3] / (e[2] + e[3])])
    assert H.maxdiff(a.squeeze(1), want) < 1e-7
    agg = restate.attention_aggregate(torch.zeros(3, 64), torch.zeros(4, 64), torch.ones(4, 64), dst, 3)
    assert H.maxdiff(agg[1], torch.zeros(64)) == 0 and H.maxdiff(agg[0], torch.ones(64)) < 1e-6


def test_scene_independence_of_the_oracle():
    """every edge set is intra-scene: a scene's encoder output does not depend on batch mates."""
    import restate
    from trajsde_amd.data import collate
    from trajsde_amd.synth import synth
    model, cfg = H.build_model(2, 5, 0.5, init_seed=5)
    a = synth(S=1, n=5, L=4, F=5, box=60.0, seed=21)
    b = synth(S=1, n=7, L=3, F=5, box=60.0, seed=22)
    both = collate([a, b])
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    c = restate.flat_cfg(cfg)
    zeros = restate.InjectedNoise(torch.zeros(2, 21, 2), torch.zeros(21, 14, 64), torch.zeros(5, 24, 64))
    zeros1 = restate.InjectedNoise(torch.zeros(1, 21, 2), torch.zeros(21, 6, 64), torch.zeros(5, 10, 64))
    o2 = restate.forward(P, cfg, both, zeros, want_intermediates=True)
    o1 = restate.forward(P, cfg, a, zeros1, want_intermediates=True)
    assert H.maxdiff(o2["local_embed"][:5], o1["local_embed"]) < 1e-5
    assert H.maxdiff(o2["loc"][:, :5], o1["loc"]) < 1e-5
