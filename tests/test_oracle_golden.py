"""The oracle (oracle/reference_twin.py) against the golden vectors recorded from the
reference itself by tests/golden/make_golden.py.  CPU only, no /root/reference needed.

The twin was bit-identical to the reference on every vector when the vectors were recorded
(same torch build); the tolerances below only leave room for a different CPU / BLAS.
"""
import numpy as np
import pytest
import torch

from oracle import reference_twin as twin
from oracle import synth

CFGS = ["ala2", "chignolin", "trp_cage", "bba", "villin", "protein_g"]
NORM_STD = {"chignolin": 3.113133430480957, "villin": 6.082900047302246, "ala2": 0.9449278712272644}


def _params(cfg, **kw):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    return twin.to_torch(synth.synth_gnn_params(N, H, L, **kw)), (N, H, L)


def test_param_counts_match_reference_probe():
    # SURVEY.md section 6 "model sizes (params)", probed from the reference's own modules
    expect = dict(ala2=647297, chignolin=600129, trp_cage=1392001, bba=972577, villin=1393921,
                  protein_g=1396609)
    for cfg, n in expect.items():
        _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
        assert synth.count_params(N, H, L) == n


@pytest.mark.parametrize("cfg", CFGS)
def test_score_matches_reference(cfg, golden):
    g = golden(f"score_{cfg}.npz")
    p, (N, H, L) = _params(cfg)
    f, e = twin.score(p, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), L, return_energy=True)
    np.testing.assert_allclose(f.numpy(), g["forces32"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(e.numpy(), g["energy32"], rtol=0, atol=1e-6)
    f1 = twin.score(p, torch.from_numpy(g["x1"]), torch.from_numpy(g["t1"]), L)
    np.testing.assert_allclose(f1.numpy(), g["forces1"], rtol=0, atol=2e-7)
    # forces are mean-free over beads (translation invariance), SURVEY 8b
    assert np.abs(f.numpy().sum(1)).max() < 1e-6


@pytest.mark.parametrize("cfg", ["chignolin", "villin"])
def test_score_float64_matches_reference(cfg, golden):
    g = golden(f"score_{cfg}.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    p = twin.to_torch(synth.synth_gnn_params(N, H, L), torch.float64)
    f = twin.score(p, torch.from_numpy(g["x"]).double(), torch.from_numpy(g["t"]).double(), L)
    np.testing.assert_allclose(f.numpy(), g["forces64"], rtol=0, atol=1e-12)


def test_layer_intermediates(golden):
    g = golden("layers_chignolin.npz")
    p, (N, H, L) = _params("chignolin")
    inter = {}
    xc = twin.center_zero(torch.from_numpy(g["x"]))
    twin.energy(p, xc, torch.from_numpy(g["t"]), L, intermediates=inter)
    for l in range(L):
        for name in ("attn_out", "nodes1", "ff", "nodes2"):
            np.testing.assert_allclose(inter[f"l{l}.{name}"].numpy(), g[f"l{l}.{name}"], rtol=0, atol=2e-6)


def test_schedule_known_answers(golden):
    g = golden("constants.npz")
    s = twin.make_schedule(1000)
    for k in g.files:
        np.testing.assert_array_equal(s[k].numpy(), g[k])
    # SURVEY.md section 8c "Deterministic known answers"
    assert s["betas"][0].item() == pytest.approx(4.128422370e-05, rel=1e-6)
    assert s["betas"][999].item() == pytest.approx(0.999, rel=1e-6)
    assert s["alphas_cumprod"][20].item() == pytest.approx(0.9981142282, rel=1e-7)
    assert s["sqrt_one_minus_alphas_cumprod"][20].item() == pytest.approx(4.342546687e-02, rel=1e-6)
    assert s["posterior_log_variance_clipped"][0].item() == pytest.approx(-46.0517006, rel=1e-6)
    assert s["posterior_mean_coef1"][0].item() == 1.0 and s["posterior_mean_coef2"][0].item() == 0.0


def test_langevin_unit_constants():
    # chignolin t=20, T=340; villin t=5, T=360; ala2 fold1 t=8, T=300
    s = twin.make_schedule(1000)
    c = twin.langevin_constants(3.113133430480957, 20, s, 340, 340, [12.0] * 10, 1.0, None)
    assert c["kb_inv"] == pytest.approx(11.656315268, rel=1e-8)
    assert c["dt"] == pytest.approx(7.758052962e-04, rel=1e-6)
    assert c["vscale"] == pytest.approx(0.999224495563, rel=1e-9)
    assert c["noisescale"] == pytest.approx(3.937521387e-02, rel=1e-6)
    assert 1.0 / (c["kbt_inv"] * c["sigma_t"]) == pytest.approx(671.696578, rel=1e-6)
    c = twin.langevin_constants(6.082900047302246, 5, s, 360, 360, [12.0] * 35, 1.0, None)
    assert c["kb_inv"] == pytest.approx(44.502783505, rel=1e-8)
    assert c["dt"] == pytest.approx(4.755178485e-04, rel=1e-6)
    c = twin.langevin_constants(0.9449278712272644, 8, s, 300, 300, [12.8] * 5, 1.0, None)
    assert c["dt"] == pytest.approx(2.503293691e-05, rel=1e-6)


@pytest.mark.parametrize("cfg", ["chignolin", "ala2"])
def test_p_sample(cfg, golden):
    g = golden(f"psample_{cfg}.npz")
    p, (N, H, L) = _params(cfg)
    s = twin.make_schedule(1000)
    for t in (999, 500, 1, 0):
        y = twin.p_sample(p, s, torch.from_numpy(g[f"x_{t}"]), t, torch.from_numpy(g[f"noise_{t}"]), L)
        np.testing.assert_allclose(y.numpy(), g[f"y_{t}"], rtol=1e-6, atol=1e-6)


def test_p_sample_loop_with_clamp(golden):
    g = golden("ploop_chignolin.npz")
    p, (N, H, L) = _params("chignolin")
    y = twin.p_sample_loop(p, twin.make_schedule(), torch.from_numpy(g["x5"]), torch.from_numpy(g["noises"]), 4, L)
    np.testing.assert_allclose(y.numpy(), g["x0"], rtol=1e-5, atol=1e-5)
    assert np.abs(y.numpy().mean(1)).max() < 1e-3  # assert_center_zero, utils.py:73-86


@pytest.mark.parametrize("name,cfg", [("langevin_chignolin_0", "chignolin"), ("langevin_chignolin_1", "chignolin"),
                                      ("langevin_chignolin_2", "chignolin"), ("langevin_ala2_3", "ala2"),
                                      ("langevin_villin_4", "villin"), ("langevin_chignolin_5", "chignolin"),
                                      ("langevin_chignolin_kcal_r30", "chignolin"), ("langevin_villin_kcal_r31", "villin")])
def test_langevin(name, cfg, golden):
    g = golden(name + ".npz")
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    p = twin.to_torch(synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
    friction = None if g["friction"] < 0 else float(g["friction"])
    norm = float(g["norm"])
    kb = str(g["kb"]) if "kb" in g.files else "consistent"       # round-3 vectors: kb="kcal" (dynamics/langevin.py:141-144)
    c = twin.langevin_constants(norm, int(g["t_level"]), twin.make_schedule(), float(g["temp"]), float(g["temp"]),
                                list(g["masses"]), friction, None if kb == "kcal" else float(g["dt"]), kb=kb)
    if kb == "kcal":
        assert abs(c["dt"] / float(g["dt"]) - 1.0) < 1e-6
    fr, ke, xl, vl = twin.simulate(p, torch.from_numpy(g["init"]) / norm, torch.from_numpy(g["noises"]),
                                   list(g["masses"]), c, L, int(g["save"]))
    np.testing.assert_allclose((fr.reshape(-1, N, 3) * norm).numpy(), g["traj"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xl.numpy(), g["x_last"], rtol=1e-6, atol=1e-6)
    if friction is not None:
        np.testing.assert_allclose(vl.numpy(), g["v_last"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ke.numpy(), g["ke"], rtol=1e-5, atol=1e-7)


def test_num_to_groups():
    assert twin.num_to_groups(1000, 256) == [256, 256, 256, 232]
    assert twin.num_to_groups(512, 256) == [256, 256]
    assert twin.num_to_groups(3, 256) == [3]
