"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle twin and the
golden vectors recorded from the reference.  Run with ``-m gpu`` on an MI355X.

Tolerances (fp32 path): the reference's own float32 run differs from its
float64 run by ~1.1-1.7e-6 relative (printed by tests/golden/make_golden.py).  The HIP path uses
a different (factorised) formulation and MFMA k-order, so it is held to
    ||F_hip - F_ref64|| / ||F_ref64||  <=  GUARD x ||F_ref32 - F_ref64|| / ||F_ref64||     (and <= 1e-5 absolutely)
GUARD = 2.0 for the split engine -- every shipped architecture's default path (SURVEY 8c's "2x, tighten after measuring";
measured 0.5-1.7x over 870 random shapes) -- and 2.5 for the fp32-MFMA engine (DFF_SPLIT_BF16=0, the `gen` input branches,
hidden 256): its k-order inside a 16-block is a fixed permutation of the reference's and it measures up to 2.1x on random
shapes (profiles/r05/fuzz4.txt), 1.5x on trp-cage.
    max|F_hip - F_ref32|               <=  1e-4 * max|F_ref32|
single integrator / reverse steps on identical noise to 2e-5 relative (measured ~1e-7), and K-step fused trajectories to
STEP_TOL x K = 5e-6 K relative (round 3: tightened from 2e-5 K; a 5x regression of the measured error no longer passes).
"""
import os

import numpy as np
import pytest
import torch

from oracle import kernel_model as km
from oracle import reference_twin as twin
from oracle import synth

pytestmark = pytest.mark.gpu

CFGS = ["ala2", "chignolin", "trp_cage", "bba", "villin", "protein_g"]
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
NORM_STD = {"chignolin": 3.113133430480957, "villin": 6.082900047302246, "ala2": 0.9449278712272644,
            "protein_g": 6.354289531707764, "trp_cage": 5.08211088180542, "bba": 6.294918537139893}


@pytest.fixture(scope="module")
def dff():
    import dff_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dff_amd.load_library()  # must exist: no fallback
    return dff_amd


_models = {}


def get_model(dff, cfg, decoder_scale=1.0):
    key = (cfg, decoder_scale)
    if key not in _models:
        from dff_amd.score import GraphTransformer
        _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
        params = synth.synth_gnn_params(N, H, L, decoder_scale=decoder_scale)
        _models[key] = (GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True,
                                         use_abs_coords=False, use_distances=False, conservative=True,
                                         state_dict=params), params)
    return _models[key]


GUARD = 2.0        # rel(hip, ref64) <= GUARD * rel(ref32, ref64): the split engine (every shipped architecture's default path)
GUARD_FP32 = 2.5   # ... the fp32-MFMA engine (DFF_SPLIT_BF16=0, `gen` branches, hidden 256)


def guard_for(kname):
    """The bar that goes with the kernel that ran (its name says which engine multiplied the weights)."""
    return GUARD if "split_" in kname else GUARD_FP32
STEP_TOL = 5e-6    # per fused step, relative to the trajectory's largest entry


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def twin_refs(params, x, t, L):
    """The oracle's float32 and float64 forces for (x, t) (numpy in, numpy out): the twin is bit-identical to the
    reference in float32 (tests/test_oracle_golden.py) and the same code in float64."""
    r32 = twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy()
    r64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x).double(), torch.from_numpy(t).double(), L).numpy()
    return r32, r64


def test_mfma_gemm_stage(dff):
    """The MFMA GEMM routine + weight packing, asymmetric operands (catches transposes)."""
    from dff_amd.binding import debug_gemm
    rng = np.random.default_rng(0)
    for (M, K, Nout) in [(10, 64, 96), (16, 64, 16), (35, 128, 64), (64, 128, 160), (1, 64, 32)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((K, Nout)).astype(np.float32)
        W[0, :] += np.arange(Nout)  # strongly asymmetric
        out = debug_gemm(A, W)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("cfg", CFGS)
def test_score_vs_reference_golden(dff, cfg, golden):
    g = golden(f"score_{cfg}.npz")
    model, _ = get_model(dff, cfg)
    x = torch.from_numpy(g["x"]).cuda()
    t = torch.from_numpy(g["t"]).cuda()
    f, e = model.native.score(x, t, return_energy=True)
    f, e = f.cpu().numpy(), e.cpu().numpy()
    r64 = rel(f, g["forces64"])
    r32 = rel(g["forces32"], g["forces64"])
    print(f"{cfg}: rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e} kernel={model.native.last_launch()}")
    assert r64 <= 1e-5 and r64 <= guard_for(model.native.last_launch()[0]) * r32
    assert np.abs(f - g["forces32"]).max() <= 1e-4 * np.abs(g["forces32"]).max()
    np.testing.assert_allclose(e[..., None], g["energy32"], rtol=0, atol=2e-5)
    assert np.abs(f.sum(1)).max() < 2e-6  # mean-free forces
    f1 = model(torch.from_numpy(g["x1"]).cuda(), torch.eye(model.num_beads), torch.from_numpy(g["t1"]).cuda())
    assert rel(f1.cpu().numpy(), g["forces1"]) <= 1e-5


def test_stash_intermediates_vs_kernel_model(dff, golden):
    """Every stashed forward intermediate of the kernel against the float64 factorised model."""
    g = golden("layers_chignolin.npz")
    model, params = get_model(dff, "chignolin")
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    model.native.score(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda())
    torch.cuda.synchronize()
    fw = km.fold_weights(params, L)
    if H == km.DH and os.environ.get("DFF_FOLD_KV", "1") != "0":   # hidden == head dimension: the library folds W_k / W_v away
        fw = km.fold_kv(fw)
    xc = g["x"].astype(np.float64)
    xc -= xc.mean(1, keepdims=True)
    _, st = km.forward(fw, xc, g["t"])
    worst = {}
    for b in range(2):
        for l in range(L):
            s = st[l]
            # the rows<=16 kernel stashes gelu'(h_pre) in the h_pre slot (backward epilogue = one multiply)
            exp = dict(nodes_in=s["nodes_in"][b], attn_out=s["attn_out"][b], ff=s["ff"][b], h_pre=km.gelu_grad(s["h_pre"][b]),
                       q=s["q"][b].transpose(1, 0, 2).reshape(N, 512), k=s["k"][b].transpose(1, 0, 2).reshape(N, 512),
                       v=s["v"][b].transpose(1, 0, 2).reshape(N, 512), P=s["P"][b])
            for name, ref in exp.items():
                if name in ("k", "v") and "fold_kv" in model.native.last_launch()[0]:
                    continue   # keys and values ARE the LayerNorm rows there: never projected, never stashed
                got = model.native.debug_stash(b, l, name)
                err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
                worst[name] = max(worst.get(name, 0.0), err)
            u = model.native.debug_stash(b, l, "u")[:, :24].reshape(N, 8, 3).transpose(1, 0, 2)
            worst["u"] = max(worst.get("u", 0.0), np.abs(u - s["u"][b]).max() / max(1.0, np.abs(s["u"][b]).max()))
    print("stash worst errors:", {k: f"{v:.2e}" for k, v in worst.items()})
    for name, err in worst.items():
        assert err < 2e-5, (name, err)
    # and against the reference's own recorded layer outputs
    for l in range(L):
        got = np.stack([model.native.debug_stash(b, l, "attn_out") for b in range(2)])
        np.testing.assert_allclose(got, g[f"l{l}.attn_out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("cfg,G", [("ala2", 1), ("ala2", 3), ("chignolin", 1), ("chignolin", 3)])
def test_grouping_and_ragged_batches(dff, cfg, G):
    """Proteins per workgroup (G) and a batch that does not divide by G give the same forces."""
    model, params = get_model(dff, cfg)
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    B = 7
    x = synth.normal((B, N, 3), 11, 3).astype(np.float32)
    t = np.linspace(0.001, 0.9, B).astype(np.float32)
    ref32, ref64 = twin_refs(params, x, t, L)
    model.native.set_group(G)
    try:
        f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        kname = model.native.last_launch()[0]
        print(cfg, G, model.native.last_launch(), rel(f, ref64), rel(ref32, ref64))
    finally:
        model.native.set_group(0)
    assert rel(f, ref64) <= min(1e-5, guard_for(kname) * rel(ref32, ref64))


@pytest.mark.parametrize("cfg", ["ala2", "chignolin"])
def test_generic_kernel_on_small_configs(dff, cfg, golden):
    """rows <= 16 normally take the one-head-per-wave kernel (dff_small.hip); the generic kernel
    must give the same answer on the same inputs."""
    g = golden(f"score_{cfg}.npz")
    model, _ = get_model(dff, cfg)
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    # (hidden 96 -- ala2 -- runs the <= 64-row kernel's one-row-tile split variant by default since round 5: asking for a wave
    # count of the <= 16-row kernel selects that kernel)
    if cfg == "ala2":
        model.native.small_waves(8)
    try:
        f_small = model.native.score(x, t).cpu().numpy()
        k_small = model.native.last_launch()[0]
    finally:
        model.native.small_waves(0)
    model.native.force_generic(True)
    try:
        f_gen = model.native.score(x, t).cpu().numpy()
        k_gen = model.native.last_launch()[0]
    finally:
        model.native.force_generic(False)
    if cfg == "ala2":   # ... and the default choice is the generic variant
        _ = model.native.score(x, t)
        assert model.native.last_launch()[0].startswith("dff_fused_kernel<96,1,4,false,split_f16")
    assert "small" in k_small and "fused" in k_gen
    assert rel(f_gen, g["forces64"]) <= 1e-5 and rel(f_small, g["forces64"]) <= 1e-5
    assert rel(f_small, f_gen) <= 5e-6


def test_score_invariances_full_batch(dff):
    """BASELINE config-2 size (chignolin, batch 256): size-independent properties."""
    model, params = get_model(dff, "chignolin")
    N = 10
    B = 256
    x = torch.from_numpy(synth.normal((B, N, 3), 5, 1).astype(np.float32)).cuda()
    t = torch.full((B,), 0.02, device="cuda")
    f = model.native.score(x, t)
    # translation invariance (the op centres its input), batch-permutation equivariance, determinism
    f_shift = model.native.score(x + torch.tensor([5.0, -3.0, 1.0], device="cuda"), t)
    assert (f - f_shift).abs().max().item() < 5e-6
    perm = torch.randperm(B, device="cuda")
    f_perm = model.native.score(x[perm].contiguous(), t)
    assert torch.equal(f_perm, f[perm])
    assert torch.equal(model.native.score(x, t), f)
    assert f.sum(1).abs().max().item() < 2e-6
    # spot-check 8 of the 256 against the oracle
    ref = twin.score(twin.to_torch(params), x[:8].cpu(), t[:8].cpu(), 3).numpy()
    assert rel(f[:8].cpu().numpy(), ref) <= 1e-5


def test_schedule_tables(dff, golden):
    g = golden("constants.npz")
    model, _ = get_model(dff, "chignolin")
    for name in g.files:
        got = model.native.schedule(name)
        np.testing.assert_allclose(got, g[name], rtol=2e-7, atol=0)


def _diffusion(dff, cfg, decoder_scale=1.0, norm=1.0):
    from dff_amd.ddpm import GaussianDiffusion
    model, params = get_model(dff, cfg, decoder_scale)
    return GaussianDiffusion(model, num_atoms=model.num_beads, timesteps=1000, norm_factor=norm), params


@pytest.mark.parametrize("cfg", ["chignolin", "ala2", "trp_cage", "bba", "villin", "protein_g"])
def test_p_sample_golden(dff, cfg, golden):
    g = golden(f"psample_{cfg}.npz")
    diff, _ = _diffusion(dff, cfg)
    for t in (999, 500, 1, 0):
        x = torch.from_numpy(g[f"x_{t}"]).cuda()
        y = diff.p_sample(x, torch.full((3,), t, dtype=torch.long, device="cuda"), noise=torch.from_numpy(g[f"noise_{t}"]).cuda())
        np.testing.assert_allclose(y.cpu().numpy(), g[f"y_{t}"], rtol=2e-5, atol=2e-5 * np.abs(g[f"y_{t}"]).max())


@pytest.mark.parametrize("cfg", ["chignolin", "protein_g"])
def test_fused_reverse_loop_golden(dff, cfg, golden):
    """5 fused reverse steps incl. the +-1000 clamp path (ddpm.py:248-251) in ONE launch (protein G: the SPILL variant of
    the <= 64-row kernel).  One sample has a coordinate at the clamp, so its entries are O(1000) after centring; every
    sample is held to STEP_TOL x 5 steps relative to ITS OWN largest entry -- the samples that never clamp (entries O(1)) are
    therefore checked to ~2.5e-5 absolute, not to the clamped sample's scale."""
    g = golden(f"ploop_{cfg}.npz")
    diff, _ = _diffusion(dff, cfg)
    with pytest.warns(UserWarning, match="Large molecule"):   # the chain's own end-of-loop check (ddpm.py:249), no extra call needed
        y = diff.p_sample_loop_from(torch.from_numpy(g["x5"]), 4, 0, noises=torch.from_numpy(g["noises"])).cpu().numpy()
    assert diff.last_clamped and not diff.check_clamp()       # ... which also cleared the word
    clamped = 0
    for b in range(y.shape[0]):
        scale = np.abs(g["x0"][b]).max()
        clamped += scale > 100
        np.testing.assert_allclose(y[b], g["x0"][b], rtol=0, atol=STEP_TOL * 5 * scale, err_msg=f"sample {b}")
    assert clamped == 1
    y = torch.from_numpy(y)
    assert y.mean(1).abs().max().item() < 1e-3


def test_fused_single_steps_match_p_sample(dff, golden):
    g = golden("psample_chignolin.npz")
    diff, params = _diffusion(dff, "chignolin")
    sched = twin.make_schedule()
    p = twin.to_torch(params)
    for t in (999, 500, 1, 0):
        x = torch.from_numpy(g[f"x_{t}"])
        nz = torch.from_numpy(g[f"noise_{t}"])
        ref = twin.center_zero(torch.clamp(twin.p_sample(p, sched, x, t, nz, 3), -1000, 1000)).numpy()
        y = diff.p_sample_loop_from(x, t, t, noises=nz[None]).cpu().numpy()
        np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("name,cfg", [("langevin_chignolin_0", "chignolin"), ("langevin_chignolin_1", "chignolin"),
                                      ("langevin_chignolin_2", "chignolin"), ("langevin_ala2_3", "ala2"),
                                      ("langevin_villin_4", "villin"), ("langevin_chignolin_5", "chignolin"),
                                      ("langevin_trp_cage_r20", "trp_cage"), ("langevin_bba_r21", "bba"),
                                      ("langevin_villin_r22", "villin"), ("langevin_protein_g_r23", "protein_g"),
                                      ("langevin_protein_g_r24", "protein_g"),
                                      ("langevin_chignolin_kcal_r30", "chignolin"), ("langevin_villin_kcal_r31", "villin")])
def test_langevin_golden(dff, name, cfg, golden):
    from dff_amd.langevin import LangevinDiffusion
    g = golden(name + ".npz")
    norm = float(g["norm"])
    diff, _ = _diffusion(dff, cfg, decoder_scale=1e-2, norm=norm)
    friction = None if g["friction"] < 0 else float(g["friction"])
    K, save = int(g["K"]), int(g["save"])
    kb = str(g["kb"]) if "kb" in g.files else "consistent"       # round-3 vectors: kb="kcal" (dynamics/langevin.py:141-144)
    ld = LangevinDiffusion(diff, torch.from_numpy(g["init"]), K, save_interval=save, t=int(g["t_level"]),
                           diffusion_steps=1000, temp_data=float(g["temp"]), temp_sim=float(g["temp"]),
                           dt=None if kb == "kcal" else float(g["dt"]),     # (kcal: the derived dt of :161-167 is under test too)
                           masses=[float(m) for m in g["masses"]], friction=friction, kb=kb, verbose=False)
    if kb == "kcal":
        assert abs(ld.dt / float(g["dt"]) - 1.0) < 1e-6
    traj = ld.sample(noises=torch.from_numpy(g["noises"]))
    assert traj.shape == g["traj"].shape and traj.dtype == torch.float32 and traj.device.type == "cpu"
    tol = STEP_TOL * K
    np.testing.assert_allclose(traj.numpy(), g["traj"], rtol=tol, atol=tol * np.abs(g["traj"]).max())
    np.testing.assert_allclose(ld.x.cpu().numpy(), g["x_last"], rtol=tol, atol=tol * np.abs(g["x_last"]).max())
    if friction is not None:
        np.testing.assert_allclose(ld.v.cpu().numpy(), g["v_last"], rtol=tol, atol=tol * np.abs(g["v_last"]).max())
        np.testing.assert_allclose(ld.kinetic_energies, g["ke"], rtol=10 * tol, atol=1e-6)


def test_langevin_chunked_equals_single_launch(dff):
    """n_steps split over several launches (state carried in x, v; Philox keyed by absolute step)
    reproduces the single-launch trajectory bit for bit."""
    from dff_amd.langevin import LangevinDiffusion
    diff, _ = _diffusion(dff, "chignolin", decoder_scale=1e-2, norm=NORM_STD["chignolin"])
    init = torch.from_numpy(synth.normal((5, 10, 3), 3, 9).astype(np.float32)) * 3.0
    kw = dict(n_timesteps=40, save_interval=10, t=20, temp_data=340, temp_sim=340, dt=None, masses=[12.0] * 10,
              friction=1.0, verbose=False, seed=77)
    a = LangevinDiffusion(diff, init, **kw).sample()
    b = LangevinDiffusion(diff, init, chunk=10, **kw).sample()
    assert torch.equal(a, b)
    assert torch.isfinite(a).all()


def test_iid_sample_properties(dff):
    """Full 1000-step chain, fused: centred output, finite, deterministic in (seed, offset)."""
    diff, _ = _diffusion(dff, "ala2", decoder_scale=1e-2, norm=NORM_STD["ala2"])
    diff.seed(5)
    a = diff.sample(6)
    diff.seed(5)
    b = diff.sample(6)
    assert a.shape == (6, 5, 3) and torch.isfinite(a).all() and torch.equal(a, b)
    assert (a / NORM_STD["ala2"]).mean(1).abs().max().item() < 1e-3
    assert a.std().item() > 1e-3


def _write_model_dir(path, cfg, decoder_scale=1e-2):
    """A saved_models/<mol>-style directory in the reference's format: args.pickle (argparse
    Namespace that also pickles an nn.Module, as the shipped ones do) + model-best.pt whose
    ["ema"] entry is an EMA(GaussianDiffusion) state-dict (trainer.py:181-206, sample.py:154-167)."""
    import argparse
    import pickle
    mol, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    ns = argparse.Namespace(mol=mol, mean0=True, fold=1, shuffle_data_before_splitting=True, scale_data=True,
                            backbone_network="graph-transformer", hidden_features_gnn=H, num_layers_gnn=L,
                            use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True,
                            diffusion_steps=1000, loss_weights="higheruntil_100", activation=torch.nn.Tanh())
    with open(path / "args.pickle", "wb") as f:
        pickle.dump(ns, f)
    params = synth.synth_gnn_params(N, H, L, decoder_scale=decoder_scale)
    gd = {k: v.clone() for k, v in twin.make_schedule().items()}
    gd["p2_loss_weight"] = torch.ones(1000)
    gd.update({"model." + k: torch.from_numpy(v) for k, v in params.items()})
    ema = {"initted": torch.tensor([True]), "step": torch.tensor([123])}
    ema.update({"ema_model." + k: v for k, v in gd.items()})
    ema.update({"online_model." + k: torch.zeros_like(v) for k, v in gd.items()})
    torch.save({"step": 123, "model": {k: torch.zeros_like(v) for k, v in gd.items()}, "ema": ema}, path / "model-best.pt")
    return params, (N, H, L)


def test_cli_end_to_end(dff, tmp_path):
    """sample.py flags, input files and output files of the reference (sample.py:101-249)."""
    from dff_amd import cli
    params, (N, H, L) = _write_model_dir(tmp_path, "chignolin")
    out = cli.main(["--model_path", str(tmp_path), "--gen_mode", "iid", "--num_samples_eval", "10",
                    "--batch_size_gen", "4", "--seed", "3"])
    f = tmp_path / "main_eval_output_iid" / "sample-iid.pt"
    saved = torch.load(f)
    assert saved.shape == (10, N, 3) and saved.dtype == torch.float32 and torch.equal(saved, out)
    assert torch.isfinite(saved).all() and (saved / 3.113133430480957).mean(1).abs().max() < 1e-3
    pdb = (tmp_path / "main_eval_output_iid" / "sample-iid.pdb").read_text()
    assert pdb.count("MODEL") == 10 and pdb.count("ATOM") == 10 * N
    # batches of 4,4,2 draw from one global Philox stream: same seed, other batching, same samples
    out2 = cli.main(["--model_path", str(tmp_path), "--gen_mode", "iid", "--num_samples_eval", "10",
                     "--batch_size_gen", "10", "--seed", "3", "--append_exp_name", "b10"])
    assert (tmp_path / "main_eval_output_iid_b10" / "sample-iid.pt").exists()
    assert torch.equal(out, out2)
    # langevin: P * n_timesteps / save_interval frames, simulation-major
    out3 = cli.main(["--model_path", str(tmp_path), "--gen_mode", "langevin", "--parallel_sim", "3",
                     "--n_timesteps", "20", "--save_interval", "10", "--batch_size_gen", "3", "--masses", "[12.0]*10"])
    assert out3.shape == (3 * 2, N, 3) and torch.isfinite(out3).all()
    assert (tmp_path / "main_eval_output_langevin" / "sample-langevin.pt").exists()
    with pytest.raises(Exception):
        cli.main(["--model_path", str(tmp_path), "--gen_mode", "bogus"])
    with pytest.raises(ValueError):  # save_interval must divide n_timesteps (langevin_cgnet.py:305-309)
        cli.main(["--model_path", str(tmp_path), "--gen_mode", "langevin", "--parallel_sim", "2", "--n_timesteps", "25",
                  "--save_interval", "10", "--batch_size_gen", "2"])


def test_checkpoint_loading_matches_direct_params(dff, tmp_path):
    from dff_amd import cli
    params, (N, H, L) = _write_model_dir(tmp_path, "ala2", decoder_scale=1.0)
    args = cli.load_training_args(str(tmp_path))
    ddpm, mol = cli.build_diffusion(args, str(tmp_path), "best", torch.device("cuda", 0))
    assert ddpm.norm_factor == 0.9449278712272644 and mol.n_beads == 5
    x = torch.from_numpy(synth.normal((4, N, 3), 8, 8).astype(np.float32))
    t = torch.tensor([0.1, 0.2, 0.3, 0.4])
    f = ddpm.model(x.cuda(), torch.eye(N), t.cuda()).cpu().numpy()
    ref = twin.score(twin.to_torch(params), x, t, L).numpy()
    assert rel(f, ref) <= 1e-5
    e = ddpm.model(x.cuda(), torch.eye(N), t.cuda(), return_energy=True)
    assert e.shape == (4, N, 1)


@pytest.mark.parametrize("cfg", CFGS)
def test_many_workgroups_identical_copies(dff, cfg, golden):
    """The golden inputs replicated over many workgroups, repeatedly: every copy must reproduce the
    same (correct) forces bit for bit -- catches cross-workgroup scratch overlap and races, which
    a 3-sample launch cannot see (the per-workgroup stash slots are adjacent in memory)."""
    g = golden(f"score_{cfg}.npz")
    model, _ = get_model(dff, cfg)
    N = model.num_beads
    copies = 48
    x = torch.from_numpy(np.tile(g["x"], (copies, 1, 1))).cuda()
    t = torch.from_numpy(np.tile(g["t"], copies)).cuda()
    # (ala2: two or three proteins share a workgroup's 16-row tile -- since round 5 also at this batch, as groups of two on two
    # workgroups each -- and a protein's k-steps depend on where in the tile it sits: copies are bit-identical when they sit at the
    # same place, i.e. every 6th copy whatever the group size; with one protein per workgroup, all of them)
    for group, period in ((0, 6), (1, 1)):
        model.native.set_group(group)
        for rep in range(4):
            f = model.native.score(x, t).reshape(copies, 3, N, 3)
            assert torch.equal(f[period:], f[:-period]), f"copies differ (group {group}, rep {rep})"
            assert max(rel(f[c].cpu().numpy(), g["forces64"]) for c in range(period)) <= 1e-5
    model.native.set_group(0)


def test_small_kernel_wave_variants(dff, golden):
    """chignolin runs the 8-wave (two waves per SIMD, one head per wave) variant by default; the
    4-wave variant must agree, in score mode and over a fused multi-step Langevin launch."""
    from dff_amd.langevin import LangevinDiffusion
    g = golden("score_chignolin.npz")
    model, _ = get_model(dff, "chignolin")
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    out = {}
    for w in (8, 4):
        model.native.small_waves(w)
        out[w] = (model.native.score(x, t).cpu().numpy(), model.native.last_launch()[0])
    model.native.small_waves(0)
    assert "64,8" in out[8][1] and "64,4" in out[4][1], (out[8][1], out[4][1])
    for w in (8, 4):
        assert rel(out[w][0], g["forces64"]) <= 1e-5
    assert rel(out[8][0], out[4][0]) <= 5e-6
    diff, _ = _diffusion(dff, "chignolin", decoder_scale=1e-2, norm=NORM_STD["chignolin"])
    init = torch.from_numpy(synth.normal((6, 10, 3), 3, 9).astype(np.float32)) * 3.0
    kw = dict(n_timesteps=30, save_interval=10, t=20, temp_data=340, temp_sim=340, dt=None, masses=[12.0] * 10,
              friction=1.0, verbose=False, seed=11)
    tr = {}
    for w in (8, 4):
        diff.model.native.small_waves(w)
        tr[w] = LangevinDiffusion(diff, init, **kw).sample().numpy()
    diff.model.native.small_waves(0)
    np.testing.assert_allclose(tr[8], tr[4], rtol=2e-4, atol=2e-4 * np.abs(tr[4]).max())


@pytest.mark.parametrize("cfg,G", [("chignolin", 0), ("ala2", 3), ("trp_cage", 0), ("villin", 0), ("ala2", 5)])
def test_layer0_table_is_bit_identical(dff, cfg, G):
    """The sampling loops read layer 0's x-independent inputs from a table precomputed per noise level
    ; switching the table off recomputes them every step.  Same arithmetic either way:
    trajectories and samples must agree bit for bit -- Langevin (one entry), DDPM (one entry per t),
    several proteins per workgroup, chunked launches, and a change of noise level between runs."""
    from dff_amd.langevin import LangevinDiffusion
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    diff, _ = _diffusion(dff, cfg, decoder_scale=1e-2, norm=NORM_STD.get(cfg, 5.0))
    nat = diff.model.native
    nat.set_group(G)                      # ("ala2", 5): 25 rows -> the generic kernel with 5 proteins per workgroup
    if (cfg, G) == ("ala2", 3):
        nat.small_waves(4)                # (15 rows of a hidden-96 model: keep the <= 16-row kernel's table path covered)
    init = torch.from_numpy(synth.normal((7, N, 3), 3, 19).astype(np.float32)) * 2.0
    out = {}
    try:
        for on in (True, False):
            nat.l0_table(on)
            res = []
            for t_level in (20, 8):
                kw = dict(n_timesteps=24, save_interval=6, t=t_level, temp_data=300, temp_sim=300, dt=None,
                          masses=[12.0] * N, friction=1.0, verbose=False, seed=5)
                res.append(LangevinDiffusion(diff, init, **kw).sample().numpy())
            x = torch.from_numpy(synth.normal((7, N, 3), 4, 23).astype(np.float32)).cuda()
            x = x - x.mean(1, keepdim=True)
            res.append(diff.p_sample_loop_from(x.clone(), 40).cpu().numpy())
            res.append(diff.p_sample_loop_from(x.clone(), 999, t_end=990).cpu().numpy())
            out[on] = res
            assert ("small" in nat.last_launch()[0]) == (max(G, 1) * N <= 16)
    finally:
        nat.l0_table(True)
        nat.set_group(0)
        nat.small_waves(0)
    for a, b in zip(out[True], out[False]):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cfg", ["chignolin", "trp_cage"])
def test_batch_split_over_launches_is_invisible(dff, cfg):
    """Proteins are independent: a batch that needs more than `max_workgroups` workgroups runs as consecutive
    launches over one bounded stash.  Forces, energies, trajectories (frames, x, v, KE) and samples must not
    depend on the limit."""
    from dff_amd.langevin import LangevinDiffusion
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    diff, _ = _diffusion(dff, cfg, decoder_scale=1e-2, norm=NORM_STD.get(cfg, 5.0))
    nat = diff.model.native
    B = 11
    x = torch.from_numpy(synth.normal((B, N, 3), 7, 31).astype(np.float32)).cuda()
    t = torch.linspace(0.01, 0.9, B).cuda()
    init = torch.from_numpy(synth.normal((B, N, 3), 8, 32).astype(np.float32)) * 2.0
    kw = dict(n_timesteps=12, save_interval=4, t=20, temp_data=300, temp_sim=300, dt=None, masses=[12.0] * N,
              friction=1.0, verbose=False, seed=5)
    out = {}
    nat.pair(False)    # (two workgroups per protein is a different variant with its own summation order: tested on its own)
    try:
        for lim in (2048, 3):
            nat.max_workgroups(lim)
            f, e = nat.score(x, t, return_energy=True)
            ld = LangevinDiffusion(diff, init, **kw)
            traj = ld.sample().numpy()
            xc = x - x.mean(1, keepdim=True)
            s = diff.p_sample_loop_from(xc.clone(), 30, t_end=26).cpu().numpy()
            out[lim] = [f.cpu().numpy(), e.cpu().numpy(), traj, ld.x.cpu().numpy(), ld.v.cpu().numpy(),
                        np.asarray(ld.kinetic_energies), s]
            assert nat.last_launch()[1] == (B if lim == 2048 else B - 3 * ((B - 1) // 3))   # grid of the last launch
    finally:
        nat.max_workgroups(2048)
        nat.pair(True)
    for a, b in zip(out[2048], out[3]):
        assert np.isfinite(a).all() and np.array_equal(a, b)


@pytest.mark.gpu
def test_langevin_equipartition_at_full_size(dff):
    """BASELINE config 2 as shipped (256 parallel simulations, 10 000 steps, save interval 250, in-kernel Philox
    noise): a size-independent property of the BAOA(F)B thermostat (langevin_cgnet.py:447-479).  The O step is
    v <- vscale v + sqrt(1 - vscale^2) sqrt(1 / (beta m)) xi, so once the velocities have relaxed (1 / friction =
    1290 steps here) every degree of freedom carries 1 / (2 beta) of kinetic energy whatever the force field is:
    <KE> = 3 N / (2 beta).  Checks the fused loop, the RNG stream (mean / variance / independence across
    trajectories) and the unit bookkeeping at the size the benchmark runs."""
    from dff_amd.langevin import LangevinDiffusion
    diff, _ = _diffusion(dff, "chignolin", decoder_scale=1e-2, norm=NORM_STD["chignolin"])
    P, N = 256, 10
    init = torch.from_numpy(synth.normal((P, N, 3), 11, 4).astype(np.float32)) * NORM_STD["chignolin"]
    init = init - init.mean(1, keepdim=True)
    ld = LangevinDiffusion(diff, init, n_timesteps=10000, save_interval=250, t=20, temp_data=340, temp_sim=340, dt=None,
                           masses=[12.0] * N, friction=1.0, verbose=False, seed=2024)
    traj = ld.sample()
    assert traj.shape == (P * 40, N, 3) and torch.isfinite(traj).all()
    c = twin.langevin_constants(NORM_STD["chignolin"], 20, twin.make_schedule(), 340, 340, [12.0] * N, 1.0, None)
    ke = np.asarray(ld.kinetic_energies)                      # (n_sims, n_frames)
    assert ke.shape == (P, 40)
    expect = 1.5 * N / c["beta"]
    late = ke[:, 30:]                                         # steps 7750 .. 10000: six relaxation times in
    assert abs(late.mean() / expect - 1.0) < 0.04, (late.mean(), expect)
    # per-trajectory KE is chi-square with 3N degrees of freedom: relative variance 2 / (3 N)
    rel_var = late.var() / late.mean() ** 2
    assert 0.6 * 2 / (3 * N) < rel_var < 1.5 * 2 / (3 * N), rel_var
    # early frames are still heating up from v0 = 0: <KE(t)> = KE_eq (1 - exp(-2 friction t))
    t1 = 250 * c["dt"]
    assert abs(ke[:, 0].mean() / (expect * (1 - np.exp(-2 * t1))) - 1.0) < 0.12


@pytest.mark.gpu
def test_ddpm_chain_variance_at_full_size(dff):
    """BASELINE config 3's batch (4096 chignolin samples per launch), in-kernel Philox noise: with the energy head
    zeroed the score is exactly 0, the reverse chain (ddpm.py:195-254) is the linear recursion
        x_{t-1} = (c1_t / sqrt(ac_t) + c2_t) x_t + sigma_t center_zero(z_t),
    and the per-coordinate variance after the steps 300 .. 0 follows from the schedule alone: a statistical check of
    the fused DDPM loop, its schedule tables and its RNG stream (variance, independence across samples and steps)."""
    diff, _ = _diffusion(dff, "chignolin", decoder_scale=0.0)
    B, N, T0 = 4096, 10, 300
    x = torch.from_numpy(synth.normal((B, N, 3), 5, 6).astype(np.float32))
    x = x - x.mean(1, keepdim=True)
    v0 = float(x.var())
    diff.seed(9)
    y = diff.p_sample_loop_from(x, T0, 0)
    assert torch.isfinite(y).all() and y.mean(1).abs().max().item() < 1e-4
    s = {k: v.double().numpy() for k, v in twin.make_schedule().items()}
    var = v0
    for t in range(T0, -1, -1):
        a = s["posterior_mean_coef1"][t] * s["sqrt_recip_alphas_cumprod"][t] + s["posterior_mean_coef2"][t]
        var = a * a * var + (np.exp(s["posterior_log_variance_clipped"][t]) * (1 - 1 / N) if t > 0 else 0.0)
    got = float(y.double().var())
    assert abs(got / var - 1.0) < 0.025, (got, var)
    # neighbouring samples draw from disjoint Philox counters
    yc = y.reshape(B, -1).double()
    yc = yc - yc.mean(1, keepdim=True)
    corr = (yc[:-1] * yc[1:]).sum(1) / (yc[:-1].norm(dim=1) * yc[1:].norm(dim=1))
    assert abs(float(corr.mean())) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,dec,xs", [("chignolin", 1e-6, 1.0), ("chignolin", 1e-2, 0.05), ("chignolin", 1.0, 3.0), ("chignolin", 1e2, 1.0),
                                        ("chignolin", 1e4, 10.0), ("villin", 1e-6, 1.0), ("villin", 1e2, 1.0), ("villin", 1e4, 10.0),
                                        ("protein_g", 1e-6, 1.0), ("protein_g", 1e4, 10.0), ("trp_cage", 1e-6, 0.05), ("bba", 1e4, 3.0),
                                        ("chignolin-unfolded", 1e-6, 1.0), ("chignolin-unfolded", 1.0, 0.05), ("chignolin-unfolded", 1e4, 10.0)])
def test_fp16_engine_over_gradient_magnitudes(dff, cfg, dec, xs, monkeypatch):
    """Round 5: the split variants (chignolin: dff_small_kernel<64,8,split_f16,fold_kv>; trp-cage ... protein G:
    dff_fused_kernel<...,split_f16[,pair]>) run their weight GEMMs on a TWO-piece fp16 split (2^-22 per operand, three MFMAs
    per product).  fp16 has five exponent bits, so the backward's GEMM inputs -- gradients, whose size follows the energy
    head's weights -- are scaled by powers of two before they are split (row-wise: dff_small.hip a_store_row, dff_kernels.hip
    row_pow2_scale; dQ / dK / dV of the <= 64-row kernels and of the unfolded <= 16-row variants -- DFF_FOLD_KV=0 here, models with
    absolute coordinates or distances otherwise -- by one scale per workgroup and layer: block_pow2_scale, stall_run).  Energy-head scales from 1e-6 to 1e4 (forces from 1e-7 to 1e3) and coordinates from 0.05 to
    10 sigma: forces against the reference twin's float64 run stay within the usual bar -- 2.5 x the distance of the twin's
    own float32 run on the same inputs -- and within 5e-6, the SAME relative error at every magnitude (a missing or wrong
    row scale shows up as 1e-3 .. 1e-2 at the small end, profiles/r05/f16_engine); the fp32-MFMA engine (DFF_SPLIT_BF16=0)
    runs next to it."""
    from dff_amd.score import GraphTransformer
    if cfg.endswith("-unfolded"):
        cfg = cfg[:-len("-unfolded")]
        monkeypatch.setenv("DFF_FOLD_KV", "0")
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L, seed=4321, decoder_scale=dec)
    x = (synth.normal((7, N, 3), 17, 5) * xs).astype(np.float32)
    t = np.array([0.005, 0.02, 0.1, 0.5, 0.9, 0.999, 0.0], np.float32)
    f32ref, f64 = twin_refs(params, x, t, L)
    r32 = rel(f32ref, f64)
    out = {}
    for split in (True, False):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        kname = model.native.last_launch()[0]
        assert ("split_f16" in kname) == split, kname
        assert ("fold_kv" in kname) == (split and cfg == "chignolin" and os.environ.get("DFF_FOLD_KV") != "0"), kname
        assert np.isfinite(f).all()
        out[split] = rel(f, f64)
    print(f"{cfg} {kname}: decoder x{dec:g}, x x{xs:g}: rel(fp16 engine, f64)={out[True]:.3e} rel(fp32 engine, f64)={out[False]:.3e} "
          f"rel(ref32, ref64)={r32:.3e} |F|max={np.abs(f64).max():.3e}")
    assert out[True] <= 5e-6 and out[True] <= GUARD * max(r32, 4e-7)
    assert out[False] <= 5e-6 and out[False] <= GUARD_FP32 * max(r32, 4e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["chignolin", "villin"])
def test_fp16_engine_steps_aside_for_models_out_of_its_range(dff, cfg):
    """The forward GEMM inputs are split into fp16 pieces as they are; their worst-case magnitudes follow from the weights
    (LayerNorm gains, the L1 norms of W_v's and W1's rows).  A model that could take them past 1.6e4 -- here: a LayerNorm gain
    of 5000 -- is given the fp32-MFMA engine at dff_model_create (one line on stderr), and its forces are still the
    reference's (twin, float64)."""
    from dff_amd.score import GraphTransformer
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = dict(synth.synth_gnn_params(N, H, L, seed=99, decoder_scale=1e-2))
    k = "graphtransformer.layers.1.1.0.norm.weight"            # LN2 of layer 1: the FFN's input would reach ~ 5000 sqrt(H)
    params[k] = (params[k] * 5000.0).astype(np.float32)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=params)
    x = synth.normal((5, N, 3), 3, 3).astype(np.float32)
    t = np.array([0.01, 0.02, 0.3, 0.7, 0.99], np.float32)
    f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
    kname = model.native.last_launch()[0]
    assert "split_" not in kname, kname
    ref32, ref64 = twin_refs(params, x, t, L)
    assert np.isfinite(f).all() and rel(f, ref64) <= max(1e-5, GUARD_FP32 * rel(ref32, ref64)), (rel(f, ref64), rel(ref32, ref64))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["chignolin", "trp_cage", "bba", "villin", "protein_g"])
def test_split_bf16_weight_gemms_are_fp32_exact(dff, cfg, golden, monkeypatch):
    """Default variants where they exist (chignolin: all eight weight GEMMs of the <= 16-row kernel; trp-cage, BBA, villin
    and -- round 4 -- protein G: those of the generic kernel): the weight GEMMs run on v_mfma_f32_16x16x32_bf16 with every fp32 operand split
    exactly into three bf16 pieces (six products kept).  DFF_SPLIT_BF16=0 when the model is created selects the pure
    v_mfma_f32_16x16x4_f32 variants.  Both are held to the SAME tolerances against the reference's float64 forces, and a
    fused Langevin run of one is compared with the other."""
    from dff_amd.score import GraphTransformer
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    g = golden(f"score_{cfg}.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]

    def make(split, scale):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        return GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                use_distances=False, conservative=True,
                                state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=scale))

    for split in (True, False):
        model = make(split, 1.0)
        f, e = model.native.score(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda(), return_energy=True)
        assert ("split_" in model.native.last_launch()[0]) == split, model.native.last_launch()   # split_bf16, or split_f16 (chignolin, round 5)
        f, e = f.cpu().numpy(), e.cpu().numpy()
        r64, r32 = rel(f, g["forces64"]), rel(g["forces32"], g["forces64"])
        print(f"{cfg}: split={split} rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
        assert r64 <= 1e-5 and r64 <= (GUARD if split else GUARD_FP32) * r32
        assert np.abs(f - g["forces32"]).max() <= 1e-4 * np.abs(g["forces32"]).max()
        np.testing.assert_allclose(e[..., None], g["energy32"], rtol=0, atol=2e-5)
    # 20 fused Langevin steps on supplied noise: split variant vs fp32-MFMA variant
    init = torch.from_numpy(synth.normal((6, N, 3), 2, 8).astype(np.float32)) * 3.0
    noises = torch.from_numpy(synth.normal((20, 6, N, 3), 4, 2).astype(np.float32))
    out = []
    for split in (False, True):
        mdl = make(split, 1e-2)
        diff = GaussianDiffusion(mdl, num_atoms=N, timesteps=1000, norm_factor=3.0)
        ld = LangevinDiffusion(diff, init, n_timesteps=20, save_interval=5, t=20, temp_data=340, temp_sim=340, dt=None,
                               masses=[12.0] * N, friction=1.0, verbose=False)
        out.append(ld.simulate(noises=noises))
        assert ("split_" in mdl.native.last_launch()[0]) == split
    assert rel(out[1], out[0]) <= 2e-5


@pytest.mark.gpu
def test_kv_fold_matches_the_unfolded_network(dff, golden, monkeypatch):
    """hidden == head dimension (chignolin): by default the library folds W_k into W_q and W_v into W_o (keys = values =
    LayerNorm rows; the <= 16-row FOLD kernel never projects or stashes them).  DFF_FOLD_KV=0 at model creation keeps the
    checkpoint's own four projections.  Both against the reference's float64 forces at the same tolerances, on the <= 16-row
    kernels (8 and 4 waves) and on the generic kernel (which runs the identity blocks), plus a fused Langevin run of one
    against the other."""
    from dff_amd.score import GraphTransformer
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    g = golden("score_chignolin.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]

    def make(fold, scale):
        monkeypatch.setenv("DFF_FOLD_KV", "1" if fold else "0")
        return GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                use_distances=False, conservative=True,
                                state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=scale))

    r32 = rel(g["forces32"], g["forces64"])
    for fold in (True, False):
        model = make(fold, 1.0)
        for path in ("small8", "small4", "generic"):
            model.native.force_generic(path == "generic")
            model.native.small_waves(4 if path == "small4" else 0)
            f, e = model.native.score(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda(), return_energy=True)
            name = model.native.last_launch()[0]
            assert ("fold_kv" in name) == (fold and path == "small8"), (fold, path, name)
            r64 = rel(f.cpu().numpy(), g["forces64"])
            print(f"fold={fold} {path}: {name} rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
            assert r64 <= 1e-5 and r64 <= guard_for(name) * r32
            np.testing.assert_allclose(e.cpu().numpy()[..., None], g["energy32"], rtol=0, atol=2e-5)
        model.native.force_generic(False)
        model.native.small_waves(0)
    init = torch.from_numpy(synth.normal((6, N, 3), 2, 8).astype(np.float32)) * 3.0
    noises = torch.from_numpy(synth.normal((20, 6, N, 3), 4, 2).astype(np.float32))
    out = []
    for fold in (False, True):
        mdl = make(fold, 1e-2)
        diff = GaussianDiffusion(mdl, num_atoms=N, timesteps=1000, norm_factor=3.0)
        ld = LangevinDiffusion(diff, init, n_timesteps=20, save_interval=5, t=20, temp_data=340, temp_sim=340, dt=None,
                               masses=[12.0] * N, friction=1.0, verbose=False)
        out.append(ld.simulate(noises=noises))
    assert rel(out[1], out[0]) <= 2e-5


# ---------------------------------------------------------------------------------------------
# Full-size parity with a LIVE network at the BASELINE per-GPU batches: the fused loops run K = 8 steps on supplied noise
# and a handful of trajectories / samples -- the first, the last, and the two either side of a launch boundary (a batch
# larger than max_workgroups runs as consecutive launches over one bounded stash) -- are compared with the oracle twin run
# on just those (trajectories are independent, so the twin needs only their rows of x0 and of the noise).
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,P,wgs", [("villin", 256, 100), ("protein_g", 128, 50), ("protein_g", 128, 0), ("chignolin", 256, 0)])
def test_full_size_langevin_subset_vs_oracle(dff, cfg, P, wgs):
    """(protein G at 128 per GPU runs as TWO workgroups per protein -- the PAIR variant, 256 blocks in one launch; the run
    with a workgroup limit turns that off and exercises the one-workgroup SPILL variant over three launches.)"""
    from dff_amd.langevin import LangevinDiffusion
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    # noise level: BASELINE configs 2 / 4 say noise_level=20 (chignolin, villin: sample.py's default); protein G keeps the
    # paper's 5 (SURVEY 8d config 5 names none).  One scalar of the same kernel, but it is the configured one.
    K, tlev, temp = 8, {"villin": 20, "protein_g": 5, "chignolin": 20}[cfg], {"villin": 360, "protein_g": 350, "chignolin": 340}[cfg]
    diff, params = _diffusion(dff, cfg, decoder_scale=1e-2, norm=NORM_STD[cfg])
    x0 = synth.normal((P, N, 3), 31, 3).astype(np.float32)
    x0 = (x0 - x0.mean(1, keepdims=True)) * NORM_STD[cfg]
    noises = synth.normal((K, P, N, 3), 32, 4).astype(np.float32)
    masses = [12.0] * N
    if wgs:
        diff.model.native.max_workgroups(wgs)    # P = 256 in launches of 100, 100, 56 workgroups
        diff.model.native.pair(False)
    try:
        ld = LangevinDiffusion(diff, torch.from_numpy(x0), K, save_interval=1, t=tlev, diffusion_steps=1000, temp_data=temp,
                               temp_sim=temp, dt=None, masses=masses, friction=1.0, kb="consistent", verbose=False)
        traj = ld.sample(noises=torch.from_numpy(noises)).numpy().reshape(P, K, N, 3)   # simulation-major (langevin.py:205-212)
        kname = diff.model.native.last_launch()[0]
        assert ("pair" in kname) == (cfg == "protein_g" and not wgs), kname
        assert diff.model.native.pair_status() == 0
    finally:
        diff.model.native.max_workgroups(2048)
        diff.model.native.pair(True)
    idx = sorted({0, 1, P // 2, P - 1} | ({wgs - 1, wgs, 2 * wgs - 1, 2 * wgs} if wgs else set()))
    c = twin.langevin_constants(NORM_STD[cfg], tlev, twin.make_schedule(), temp, temp, masses, 1.0, None)
    fr, ke, xl, vl = twin.simulate(twin.to_torch(params), torch.from_numpy(x0[idx]) / NORM_STD[cfg],
                                   torch.from_numpy(noises[:, idx]), masses, c, L, 1)
    ref = (fr * NORM_STD[cfg]).numpy()                                # (sims, frames, N, 3)
    tol = STEP_TOL * K
    np.testing.assert_allclose(traj[idx], ref, rtol=tol, atol=tol * np.abs(ref).max())
    np.testing.assert_allclose(ld.v.cpu().numpy()[idx], vl.numpy(), rtol=tol, atol=tol * np.abs(vl.numpy()).max())
    assert np.isfinite(traj).all()


@pytest.mark.gpu
@pytest.mark.parametrize("H,N", [(128, 32), (128, 45), (128, 49), (96, 26), (256, 20)])
def test_fused_langevin_at_odd_bead_counts(dff, H, N):
    """The fused Langevin loop (layer-0 table, stash reloads, integrator) away from the shipped sizes: a hidden-128 model at 32
    rows on the three-row-tile shape (stash rows sized by the shape, not the rows), 45 rows (fp32 engine: the split operands do
    not fit), 49 rows on the tight four-row-tile split variant with two workgroups per protein, BBA's shape at 26, hidden 256:
    6 steps on supplied noise against the oracle twin, all trajectories."""
    from dff_amd.score import GraphTransformer
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    L, K, P, norm, tlev, temp = 2, 6, 5, 3.0, 20, 340
    params = synth.synth_gnn_params(N, H, L, seed=7000 + N + H, decoder_scale=1e-2)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=params)
    diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=norm)
    x0 = synth.normal((P, N, 3), 43, N).astype(np.float32)
    x0 = (x0 - x0.mean(1, keepdims=True)) * norm
    noises = synth.normal((K, P, N, 3), 44, N).astype(np.float32)
    masses = [12.0] * N
    ld = LangevinDiffusion(diff, torch.from_numpy(x0), K, save_interval=2, t=tlev, diffusion_steps=1000, temp_data=temp,
                           temp_sim=temp, dt=None, masses=masses, friction=1.0, kb="consistent", verbose=False)
    traj = ld.sample(noises=torch.from_numpy(noises)).numpy().reshape(P, K // 2, N, 3)
    kname = model.native.last_launch()[0]
    c = twin.langevin_constants(norm, tlev, twin.make_schedule(), temp, temp, masses, 1.0, None)
    fr, ke, xl, vl = twin.simulate(twin.to_torch(params), torch.from_numpy(x0) / norm, torch.from_numpy(noises), masses, c, L, 2)
    ref = (fr * norm).numpy()
    err = np.abs(traj - ref).max() / np.abs(ref).max()
    print(f"H={H} N={N} {kname}: {K}-step Langevin rel err {err:.3e}")
    assert err <= STEP_TOL * K, (H, N, kname, err)
    assert model.native.status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("H,N", [(128, 32), (128, 49), (96, 26), (256, 20), (64, 13)])
def test_fused_reverse_loop_at_odd_bead_counts(dff, H, N):
    """The fused reverse-DDPM loop (one table entry per noise level, posterior update, clamp / centre flag) away from the shipped
    sizes -- see test_fused_langevin_at_odd_bead_counts for the shapes: 6 reverse steps t = 5 .. 0 on supplied noise against the
    oracle's p_sample_loop, all samples."""
    from dff_amd.score import GraphTransformer
    from dff_amd.ddpm import GaussianDiffusion
    L, K, B = 2, 6, 5
    params = synth.synth_gnn_params(N, H, L, seed=8000 + N + H)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=params)
    diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=3.0)
    x = synth.normal((B, N, 3), 45, N).astype(np.float32)
    x = (x - x.mean(1, keepdims=True)) * 0.6
    noises = synth.normal((K, B, N, 3), 46, N).astype(np.float32)
    y = diff.p_sample_loop_from(torch.from_numpy(x), K - 1, 0, noises=torch.from_numpy(noises)).cpu().numpy()
    kname = model.native.last_launch()[0]
    ref = twin.p_sample_loop(twin.to_torch(params), twin.make_schedule(), torch.from_numpy(x), torch.from_numpy(noises), K - 1, L).numpy()
    err = np.abs(y - ref).max() / np.abs(ref).max()
    print(f"H={H} N={N} {kname}: {K} reverse steps rel err {err:.3e}")
    assert err <= STEP_TOL * K, (H, N, kname, err)
    assert np.abs(y.mean(1)).max() < 1e-4 and model.native.status() == 0


@pytest.mark.gpu
def test_full_size_ddpm_subset_vs_oracle(dff):
    """BASELINE config 3's batch (4096 chignolin samples per launch call = two launches of 2048 workgroups) with the
    network live: 8 fused reverse steps t = 7 .. 0 on supplied noise vs the oracle's p_sample_loop on six samples."""
    cfg, B, K = "chignolin", 4096, 8
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    diff, params = _diffusion(dff, cfg)
    x = synth.normal((B, N, 3), 41, 5).astype(np.float32)
    x = (x - x.mean(1, keepdims=True)) * 0.6
    noises = synth.normal((K, B, N, 3), 42, 6).astype(np.float32)
    y = diff.p_sample_loop_from(torch.from_numpy(x), K - 1, 0, noises=torch.from_numpy(noises)).cpu().numpy()
    assert diff.model.native.last_launch()[1] == 2048                  # the batch was split at 2048 workgroups
    idx = [0, 1, 2047, 2048, 3000, 4095]
    ref = twin.p_sample_loop(twin.to_torch(params), twin.make_schedule(), torch.from_numpy(x[idx]),
                             torch.from_numpy(noises[:, idx]), K - 1, L).numpy()
    tol = STEP_TOL * K
    np.testing.assert_allclose(y[idx], ref, rtol=tol, atol=tol * np.abs(ref).max())
    assert np.isfinite(y).all() and np.abs(y.mean(1)).max() < 1e-4


# ---------------------------------------------------------------------------------------------
# The N > 1 product path on real kernels (no multi-GPU node needed): two ranks share the one GPU (DFF_DEVICE=0),
# rendezvous over gloo (DFF_DIST_BACKEND), and run the sample.py-compatible CLI end to end.  Philox counters are the
# GLOBAL sample / trajectory indices, so the gathered result must equal the one-rank run bit for bit, in
# simulation-major order (sample.py:180-214, dynamics/langevin.py:205-212).
# ---------------------------------------------------------------------------------------------
_TWO_RANK_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DFF_REPO"])
import dff_amd
from dff_amd import cli
out = cli.main(sys.argv[1:])
if int(os.environ["RANK"]) == 0:
    torch.save(out, os.environ["DFF_OUT"])
"""


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["iid", "langevin"])
def test_two_ranks_on_one_gpu_equal_one_rank(dff, tmp_path, mode, monkeypatch):
    import socket
    import subprocess
    import sys
    from dff_amd import cli
    params, (N, H, L) = _write_model_dir(tmp_path, "chignolin")
    # (ranks that share a GPU run the one-workgroup kernels -- sampling.dist_env, DFF_PAIR=0 -- and so does the one-rank reference
    # here: the two-workgroups variants sum a protein's heads in another order)
    monkeypatch.setenv("DFF_PAIR", "0")
    argv = ["--model_path", str(tmp_path), "--gen_mode", mode, "--seed", "5", "--batch_size_gen", "4"]
    argv += (["--num_samples_eval", "12"] if mode == "iid" else
             ["--parallel_sim", "6", "--n_timesteps", "20", "--save_interval", "5", "--masses", "[12.0]*10"])
    one = cli.main(argv + ["--append_exp_name", "w1"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DFF_TEST_KNOBS="1", DFF_DEVICE="0", DFF_DIST_BACKEND="gloo", DFF_REPO=ROOT, DFF_OUT=str(tmp_path / "w2.pt"))
        procs.append(subprocess.Popen([sys.executable, str(script)] + argv + ["--append_exp_name", "w2"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        assert pr.returncode == 0, out.decode()[-2000:]
    two = torch.load(tmp_path / "w2.pt")
    assert two.shape == one.shape == ((12, N, 3) if mode == "iid" else (6 * 4, N, 3))
    assert torch.equal(two, one), float((two - one).abs().max())
    saved = torch.load(tmp_path / f"main_eval_output_{mode}_w2" / f"sample-{mode}.pt")
    assert torch.equal(saved, one)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(dff):
    """bench.py's N > 1 path (process group, barriers, max-over-ranks timing, the frame all_gather, rank-0 JSON line)
    executed for real: two ranks share the one GPU (test knobs of sampling.dist_env, gloo rendezvous).  The ranks are
    time-sliced, so the whole-job value is about the one-rank value (twice the work in twice the time), not 2x."""
    import json
    import socket
    import subprocess
    import sys
    args = ["--steps", "500", "--warmup", "250", "--no-cpu", "--no-extras"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True,
                         timeout=600, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-1500:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DFF_TEST_KNOBS="1", DFF_DEVICE="0", DFF_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [pr.communicate(timeout=900) for pr in procs]
    for pr, (o, e) in zip(procs, outs):
        assert pr.returncode == 0, o[-1500:] + e[-1500:]
    json_lines = [[ln for ln in o.splitlines() if ln.startswith("{")] for o, _ in outs]
    assert len(json_lines[0]) == 1 and len(json_lines[1]) == 0         # only rank 0 prints the line (gloo chatters on stdout)
    r2 = json.loads(json_lines[0][0])
    assert r2["n_gpus"] == 2 and r2["steps"] == r1["steps"] == 2000 and r2["scaling"] == "weak" and r2["finite"]
    assert r2["metric"] == r1["metric"] and r2["config"]["kernel"] == r1["config"]["kernel"]
    assert r2["gather_ms"] > 0.0 and r1["gather_ms"] == 0.0
    ratio = r2["value"] / r1["value"]
    print(f"bench 2 ranks on 1 GPU: value {r2['value']:.1f} vs 1 rank {r1['value']:.1f} (ratio {ratio:.2f})")
    assert 0.45 < ratio < 1.35, ratio


@pytest.mark.gpu
@pytest.mark.parametrize("N", [49, 53, 56, 57, 60, 61])
def test_four_row_tiles_at_odd_bead_counts(dff, N, monkeypatch):
    """The four-row-tile shape beyond protein G's 56 beads (round 4: its split-engine variant keeps P / dS tile arrays of the
    real rows x 60 columns -- `LdsLayout`, TIGHT -- whose guards only matter when the row count is not a multiple of 4 or 16,
    and it fits the LDS up to 56 rows: 57..64 beads must fall back to the fp32 engine of the same shape by themselves).
    Forces against the oracle twin in float64, one workgroup and two workgroups per protein, split and fp32 engines."""
    from dff_amd.score import GraphTransformer
    H, L = 128, 2
    if N == 61:
        with pytest.raises(ValueError, match="does not fit the LDS"):
            GraphTransformer(62, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=synth.synth_gnn_params(62, H, L))
    params = synth.synth_gnn_params(N, H, L, seed=4000 + N)
    x = synth.normal((3, N, 3), 40, N).astype(np.float32) * 1.5
    t = np.array([0.01, 0.3, 0.7], np.float32)
    ref64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x).double(), torch.from_numpy(t).double(), L).numpy()
    ref32 = twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy()
    r32 = rel(ref32, ref64)
    for split in (True, False):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        try:
            for pair in (True, False):
                model.native.pair(pair)
                f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
                kname = model.native.last_launch()[0]
                assert kname.startswith("dff_fused_kernel<128,4,1,true") and ("pair" in kname) == pair, kname
                assert ("split_" in kname) == (split and N <= 56), (N, kname)      # 57+ rows: the split A operand does not fit
                r64 = rel(f, ref64)
                print(f"N={N} {kname}: rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
                assert r64 <= 1e-5 and r64 <= guard_for(kname) * r32, (N, kname, r64, r32)
                assert model.native.status() == 0
        finally:
            model.native.pair(True)


@pytest.mark.gpu
@pytest.mark.parametrize("H,N", [(128, 17), (128, 23), (128, 32), (128, 33), (128, 41), (128, 45), (128, 48), (96, 17), (96, 26), (96, 32),
                                 (64, 17), (64, 31)])
def test_two_and_three_row_tiles_at_odd_bead_counts(dff, H, N, monkeypatch):
    """Every shape of the <= 64-row kernel away from the shipped bead counts (row counts that are not multiples of 4 / 16, full
    tiles): forces against the oracle twin in float64, split and fp32 engines, one and two workgroups per protein.  (Round 4
    moved the split A operand, dQ and dV into bf16 pieces written by their producers; the edge-size test of the four-row-tile
    shape found a pad-row bug that the shipped sizes could not show.)"""
    from dff_amd.score import GraphTransformer
    L = 2
    params = synth.synth_gnn_params(N, H, L, seed=5000 + N + H)
    x = synth.normal((3, N, 3), 41, N).astype(np.float32) * 1.5
    t = np.array([0.01, 0.3, 0.7], np.float32)
    ref64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x).double(), torch.from_numpy(t).double(), L).numpy()
    r32 = rel(twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy(), ref64)
    mt = (N + 15) // 16
    seen = set()
    for split in (True, False):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        try:
            for pair in (True, False):
                model.native.pair(pair)
                f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
                kname = model.native.last_launch()[0]
                # (hidden 128 at 32 rows does not fit the LDS in two row tiles: the three-row-tile shape takes it)
                assert kname.startswith(f"dff_fused_kernel<{H},{3 if (H, N) == (128, 32) else mt},"), kname
                seen.add(kname)
                r64 = rel(f, ref64)
                print(f"H={H} N={N} {kname}: rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
                assert r64 <= 1e-5 and r64 <= guard_for(kname) * r32, (N, kname, r64, r32)
                assert model.native.status() == 0
        finally:
            model.native.pair(True)
    # the split engine wherever its operands fit the LDS next to the head buffers (the smallest size of every shape does);
    # beyond that the fp32 engine of the same shape takes over by itself
    assert any("split_" not in k for k in seen), seen
    if (H, N) in ((128, 17), (128, 33), (96, 17)):
        assert any("split_" in k for k in seen), seen


@pytest.mark.gpu
@pytest.mark.parametrize("H,N,G", [(64, 2, 1), (64, 3, 5), (64, 7, 2), (64, 11, 1), (64, 13, 1), (64, 16, 1), (96, 2, 8), (96, 9, 1), (96, 16, 1),
                                   (128, 4, 4), (128, 10, 1), (128, 15, 1), (256, 2, 1), (256, 16, 1), (256, 17, 1), (256, 32, 1)])
def test_small_models_at_odd_sizes(dff, H, N, G, monkeypatch):
    """Bead counts and hidden sizes between the shipped ones on the <= 16-row kernel (all its wave / engine variants: 8 waves up to
    10 rows, 4 above; split and fp32; G proteins per workgroup, ragged batch) and at hidden 256's limits (<= 64-row kernel only:
    16 | 17 rows = one | two row tiles, 32 = the largest it takes): forces against the oracle twin in float64."""
    from dff_amd.score import GraphTransformer
    L = 2
    params = synth.synth_gnn_params(N, H, L, seed=6000 + N + H)
    B = 7
    x = synth.normal((B, N, 3), 42, N).astype(np.float32) * 1.5
    t = np.linspace(0.001, 0.9, B).astype(np.float32)
    ref64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x).double(), torch.from_numpy(t).double(), L).numpy()
    r32 = rel(twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy(), ref64)
    for split in (True, False):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        model.native.set_group(G)
        f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        kname = model.native.last_launch()[0]
        # (hidden 96 / 128 with the split engine on: the one-row-tile split_f16 variant of the <= 64-row kernel, round 5)
        assert ("small" in kname) == (H != 256 and G * N <= 16 and not (H in (96, 128) and split)), kname
        r64 = rel(f, ref64)
        print(f"H={H} N={N} G={G} {kname}: rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
        assert r64 <= 1e-5 and r64 <= guard_for(kname) * max(r32, 4e-7), (H, N, G, kname, r64, r32)


@pytest.mark.gpu
def test_random_shapes(dff):
    """tests/fuzz_shapes.py, 30 cases: random hidden size / bead count / depth / input branch / head / magnitudes / batch / packing /
    engine against the oracle twin in float64 -- forces on every case, four fused Langevin and reverse-DDPM steps on the shipped
    branch -- at the bars of the fixed-shape tests.  (Round 5: a 120-case sweep is what showed the truncated fp16 pieces 2 % past
    the force bar on one shape; profiles/r05/fuzz*.txt hold the long sweeps.)"""
    import fuzz_shapes
    lines = []
    bad = fuzz_shapes.run(30, 2025, lines.append)
    print("\n".join(lines))
    assert bad == 0, [ln for ln in lines if ln.startswith("FAIL")]


@pytest.mark.gpu
def test_pair_failure_word_is_sticky_reported_once_and_never_syncs_the_launch_path(dff, golden):
    """The two-workgroups-per-protein variants' failure word (round 4): the launch path does not read it (stays
    asynchronous); a launch queued on top of a failure leaves at kernel entry with its OUTPUTS set to NaN (round 5, ADVICE
    r04: Model.score() hands out torch.empty buffers and never checks -- no uninitialised forces), the host's next status
    check raises ONCE and clears, after which the same model works again; once the host has seen the word, further launches run
    on the one-workgroup kernels without touching the device (round 6; they used to be refused) until it is cleared; Model.pair(False)
    clears it too."""
    cfg = "protein_g"
    g = golden(f"score_{cfg}.npz")
    model, _ = get_model(dff, cfg)
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    try:
        model.native.pair(True)
        good = model.native.score(x, t).cpu().numpy()
        assert "pair" in model.native.last_launch()[0] and model.native.status() == 0
        model.native.poke_status(1)                       # as a kernel that lost its partner would
        sent = torch.full_like(x, 123.0)
        import dff_amd.binding as B
        dead, dead_e = model.native.score(x, t, return_energy=True)   # accepted (the host has not looked), leaves at entry
        assert bool(torch.isnan(dead).all()) and bool(torch.isnan(dead_e).all()), "forces of a launch that did not run must be NaN"
        # ... through the raw ABI into a sentinel buffer: nothing of it survives either
        rc = model.native.lib.dff_score(model.native.handle, B._ptr(x), B._ptr(t), x.shape[0], B._ptr(sent), None, model.native._stream())
        assert rc == 0
        torch.cuda.synchronize()
        assert bool(torch.isnan(sent).all()), "a two-workgroups launch ran on top of a reported failure"
        with pytest.raises(RuntimeError, match="partner workgroup"):
            model.native.check()
        model.native.check()                              # reported once, cleared
        again = model.native.score(x, t).cpu().numpy()
        assert np.array_equal(again, good)
        model.native.poke_status(1)
        assert model.native.status() == 1                 # the host has seen it now: the two-workgroups variants are out of the choice
        one = model.native.score(x, t).cpu().numpy()      # (round 6) ... and the model keeps working on the one-workgroup kernels
        assert "pair" not in model.native.last_launch()[0] and rel(one, good) <= 5e-6
        model.native.status_clear()                       # re-arms
        assert np.array_equal(model.native.score(x, t).cpu().numpy(), good) and "pair" in model.native.last_launch()[0]
        model.native.poke_status(1)
        model.native.pair(False)                          # selects the one-workgroup kernels AND clears the word
        assert model.native.status() == 0
        one = model.native.score(x, t).cpu().numpy()
        assert rel(one, good) <= 5e-6
    finally:
        model.native.status_clear()
        model.native.pair(True)


@pytest.mark.gpu
def test_bench_rccl_world_one(dff):
    """The RCCL code path itself, executed before the first multi-GPU run does: bench.py under torchrun with ONE rank and
    DFF_FORCE_DIST=1 -> init_process_group("nccl", device_id=...) (RCCL is loaded), the barriers, the max-over-ranks
    all_reduce on a device tensor and the frame all_gather (sample.py:180-190's gather) all run on the one GPU."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DFF_TEST_KNOBS", "DFF_DIST_BACKEND", "DFF_DEVICE")}
    env.update(DFF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "500", "--warmup", "250",
                        "--no-cpu", "--no-extras"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["finite"] and j["config"]["collective_backend"] == "nccl"
    assert j["gather_ms"] > 0.0 and j["steps"] == 2000 and j["value"] > 1000.0
    print(f"bench over RCCL at world 1: {j['value']:.0f} MD-steps/s, gather {j['gather_ms']:.2f} ms")


@pytest.mark.gpu
def test_device_flag_word_persists_and_reports_centre(dff):
    """One flag word per GaussianDiffusion: a clamp in an EARLIER batch is still reported after later clean batches
    (the reference warns per step, ddpm.py:248-250), and a chain that ends off-centre -- here: NaNs, which slip through the
    reference's own entry check just the same -- raises like assert_center_zero (ddpm.py:252); reading clears the word."""
    diff, _ = _diffusion(dff, "chignolin")
    diff.defer_checks = True          # as the CLI / bench.py run it: many batches, ONE check at the end
    N = 10
    big = torch.zeros(2, N, 3)
    big[0, 0, 0], big[0, 1, 0] = 4000.0, -4000.0                      # centred, far outside +-1000
    diff.p_sample_loop_from(big, 999, 999)                             # batch 1: clamps
    clean = torch.from_numpy(synth.normal((2, N, 3), 5, 5).astype(np.float32))
    diff.p_sample_loop_from(clean - clean.mean(1, keepdim=True), 3, 0)  # batch 2: does not
    with pytest.warns(UserWarning, match="Large molecule"):
        assert diff.check_clamp()
    assert not diff.check_clamp()                                      # cleared
    bad = clean - clean.mean(1, keepdim=True)
    bad[1, 3, 2] = float("nan")
    diff.p_sample_loop_from(bad, 1, 0)
    with pytest.raises(AssertionError, match="Center not at zero"):
        diff.check_clamp()
    assert not diff.check_clamp()
    # default (defer_checks=False): every chain entry point checks at its own end, like ddpm.py:249-252
    diff.defer_checks = False
    with pytest.warns(UserWarning, match="Large molecule"):
        diff.p_sample_loop_from(big, 999, 999)
    with pytest.raises(AssertionError, match="Center not at zero"):
        diff.p_sample_loop_from(bad, 1, 0)


@pytest.mark.gpu
def test_pairs_of_groups_for_proteins_that_share_a_row_tile(dff):
    """ala2 (5 beads: up to three proteins in a 16-row tile).  Round 5: the two-workgroups variant takes a GROUP of proteins per
    pair -- the smallest group size whose pairs fit the CUs: 256 trajectories run as 128 groups of two on 256 workgroups (64 vs
    80 us per step), 300 and 384 as groups of three, 512 no longer fits and runs one workgroup per group of three.  Dispatch,
    forces against the twin (float64) on a ragged batch whose last group is short, agreement with the one-workgroup path, the
    exchange status, run-to-run bit-identity, and the fused Langevin loop on supplied noise."""
    from dff_amd.score import GraphTransformer
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    cfg = "ala2"
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L, seed=515, decoder_scale=1e-2)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=params)
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("needs the full 256 CUs")
    try:
        for B, grid, paired in ((128, 256, True), (256, 256, True), (299, 208, True), (384, 256, True), (512, 171, False)):
            x = (synth.normal((B, N, 3), 77, B) * 1.2).astype(np.float32)
            t = (0.001 + 0.9 * synth.uniform((B,), 78, B, 0.0, 1.0)).astype(np.float32)
            xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
            model.native.pair(True)
            f = model.native.score(xd, td).cpu().numpy()
            name, g_, _ = model.native.last_launch()
            assert ("pair" in name) == paired and g_ == grid, (B, name, g_)
            assert model.native.pair_status() == 0
            assert np.array_equal(f, model.native.score(xd, td).cpu().numpy())
            sub = np.r_[0:4, B - 5:B]
            f64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x[sub]).double(), torch.from_numpy(t[sub]).double(), L).numpy()
            r32 = rel(twin.score(twin.to_torch(params), torch.from_numpy(x[sub]), torch.from_numpy(t[sub]), L).numpy(), f64)
            r64 = rel(f[sub], f64)
            model.native.pair(False)
            f1 = model.native.score(xd, td).cpu().numpy()
            assert "pair" not in model.native.last_launch()[0]
            print(f"ala2 B={B}: {name} grid {g_}: rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e} rel(pair, one workgroup)={rel(f, f1):.3e}")
            assert r64 <= 1e-5 and r64 <= guard_for(name) * max(r32, 4e-7)
            assert rel(f, f1) <= 5e-6
        # the fused loop: 6 steps, 256 trajectories (groups of two), against the twin on the first and last three
        model.native.pair(True)
        K, P, norm, tlev, temp = 6, 256, 3.0, 20, 300
        diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=norm)
        x0 = synth.normal((P, N, 3), 43, 5).astype(np.float32)
        x0 = (x0 - x0.mean(1, keepdims=True)) * norm
        noises = synth.normal((K, P, N, 3), 44, 5).astype(np.float32)
        masses = [12.0] * N
        ld = LangevinDiffusion(diff, torch.from_numpy(x0), K, save_interval=2, t=tlev, diffusion_steps=1000, temp_data=temp,
                               temp_sim=temp, dt=None, masses=masses, friction=1.0, kb="consistent", verbose=False)
        traj = ld.sample(noises=torch.from_numpy(noises)).numpy().reshape(P, K // 2, N, 3)
        assert "pair" in model.native.last_launch()[0] and model.native.pair_status() == 0
        sub = np.r_[0:3, P - 3:P]
        c = twin.langevin_constants(norm, tlev, twin.make_schedule(), temp, temp, masses, 1.0, None)
        fr, _, _, _ = twin.simulate(twin.to_torch(params), torch.from_numpy(x0[sub]) / norm, torch.from_numpy(noises[:, sub]), masses, c, L, 2)
        ref = (fr * norm).numpy()
        err = np.abs(traj[sub] - ref).max() / np.abs(ref).max()
        print(f"ala2 P={P} {model.native.last_launch()[0]}: {K}-step Langevin rel err {err:.3e}")
        assert err <= STEP_TOL * K
    finally:
        model.native.pair(True)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["protein_g", "villin", "trp_cage", "bba"])
def test_pair_variant_equals_one_workgroup_variant(dff, cfg, golden):
    """Two workgroups per protein (PAIR: heads / FFN chunks split, partial tiles exchanged through L2 with agent-scope
    release / acquire) against the one-workgroup variant of the same kernel: forces vs the reference's float64 run within
    the usual tolerance for both, each other within 5e-6, every exchange found its partner (status word), repeated launches
    bit-identical (a lost or stale exchange would show up as run-to-run differences), and a batch that is not a multiple
    of the 8-protein block groups."""
    g = golden(f"score_{cfg}.npz")
    model, _ = get_model(dff, cfg)
    N = synth.SHIPPED_CONFIGS[cfg][1]
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    out = {}
    try:
        # 2: the exchange protocol of a pair whose blocks sit on different XCDs, forced; 3 (round 5): partners on
        # adjacent blocks, i.e. REALLY on different XCDs under the round-robin placement -- the kernel's own XCC-ID check picks
        # the protocol there
        for on in (True, 2, 3, False):
            model.native.pair(on)
            f = model.native.score(x, t).cpu().numpy()
            assert ("pair" in model.native.last_launch()[0]) == bool(on)
            assert model.native.pair_status() == 0
            r64, r32 = rel(f, g["forces64"]), rel(g["forces32"], g["forces64"])
            print(f"{cfg}: pair={on} rel(hip,ref64)={r64:.3e} rel(ref32,ref64)={r32:.3e}")
            assert r64 <= 1e-5 and r64 <= guard_for(model.native.last_launch()[0]) * r32
            out[on] = f
        assert rel(out[True], out[False]) <= 5e-6
        assert np.array_equal(out[True], out[2])     # same sums in the same order, whichever way the tiles travel
        assert np.array_equal(out[True], out[3])     # ... and whichever XCDs the two blocks run on
        model.native.pair(3)                         # repeated launches across XCDs: a stale or lost exchange shows up as a difference
        for _ in range(3):
            assert np.array_equal(model.native.score(x, t).cpu().numpy(), out[3])
        assert model.native.pair_status() == 0
        model.native.pair(True)
        xb = torch.from_numpy(synth.normal((13, N, 3), 6, 6).astype(np.float32)).cuda()      # 13 proteins: 2 block groups
        tb = torch.full((13,), 0.02).cuda()
        runs = [model.native.score(xb, tb).cpu().numpy() for _ in range(4)]
        assert model.native.last_launch()[1] == 32 and model.native.pair_status() == 0
        for r in runs[1:]:
            assert np.array_equal(r, runs[0])
        model.native.pair(False)
        assert rel(runs[0], model.native.score(xb, tb).cpu().numpy()) <= 5e-6
    finally:
        model.native.pair(True)


@pytest.mark.gpu
def test_small_kernel_two_workgroups_per_protein(dff, monkeypatch):
    """Round 6: the <= 16-row FOLD kernel as TWO workgroups per protein for per-GPU batches that leave half
    the CUs idle (the reference's published protocol is --parallel_sim 100, evaluate/sampling_commands.md:13).  Blocks b and b + 8
    own heads / FFN slices 0-3 and 4-7 on four waves each and exchange the row stages' partial sums + the final dE/dx.  Fused
    Langevin and reverse-DDPM runs at 100 per GPU against the one-workgroup kernel and against the twin (float64) on a subset;
    the exchange protocols (same XCD; agent scope forced; partners really on different XCDs) give the same bits; repeated
    launches are identical; past 128 per GPU the one-workgroup kernel runs by itself.  Measured: no faster than one workgroup per
    protein (45.8 vs 44.4 us per step at 128 per GPU, profiles/r06/pair_small/) -- the variant is opt-in, DFF_SMALL_PAIR=1."""
    from dff_amd.langevin import LangevinDiffusion
    monkeypatch.setenv("DFF_SMALL_PAIR", "1")
    diff, params = _diffusion(dff, "chignolin", decoder_scale=1e-2, norm=NORM_STD["chignolin"])
    nat = diff.model.native
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    P, K = 100, 8
    x0 = synth.normal((P, N, 3), 31, 5).astype(np.float32)
    x0 = (x0 - x0.mean(1, keepdims=True)) * NORM_STD["chignolin"]
    noises = synth.normal((K, P, N, 3), 32, 5).astype(np.float32)
    masses = [12.0] * N

    def run(pair):
        nat.pair(pair)
        ld = LangevinDiffusion(diff, torch.from_numpy(x0), K, save_interval=2, t=20, temp_data=340, temp_sim=340, dt=None,
                               masses=masses, friction=1.0, verbose=False)
        out = ld.sample(noises=torch.from_numpy(noises)).numpy().reshape(P, K // 2, N, 3)
        return out, nat.last_launch()
    try:
        out = {}
        for on in (True, 2, 3, False):
            out[on], (kname, grid, _) = run(on)
            assert ("pair" in kname) == bool(on) and "fold_kv" in kname, kname
            assert grid == (2 * 8 * 13 if on else P) and nat.pair_status() == 0, (on, grid)
            assert np.isfinite(out[on]).all()
        assert np.array_equal(out[True], out[2]) and np.array_equal(out[True], out[3])
        assert np.abs(out[True] - out[False]).max() <= STEP_TOL * K * np.abs(out[False]).max()
        again, _ = run(True)
        assert np.array_equal(again, out[True])
        # against the twin in float64 on the first, a middle and the last trajectories
        sub = [0, 1, 49, 98, 99]
        c = twin.langevin_constants(NORM_STD["chignolin"], 20, twin.make_schedule(), 340, 340, masses, 1.0, None)
        fr, _, _, _ = twin.simulate(twin.to_torch(params, torch.float64), torch.from_numpy(x0[sub]).double() / NORM_STD["chignolin"],
                                    torch.from_numpy(noises[:, sub]).double(), masses, c, L, 2)
        ref = (fr * NORM_STD["chignolin"]).numpy()
        err = np.abs(out[True][sub] - ref).max() / np.abs(ref).max()
        print(f"two workgroups per protein, {P} trajectories x {K} steps: rel err vs twin (float64) {err:.3e}")
        assert err <= STEP_TOL * K
        # in-kernel Philox noise + the reverse-DDPM loop (the layer-0 table has 1000 entries there)
        nat.pair(True)
        diff.seed(3)
        y1 = diff.p_sample_loop_from(torch.from_numpy(x0[:37] / NORM_STD["chignolin"]), 11, 0).cpu().numpy()
        assert "pair" in nat.last_launch()[0]
        nat.pair(False)
        diff.seed(3)
        y0 = diff.p_sample_loop_from(torch.from_numpy(x0[:37] / NORM_STD["chignolin"]), 11, 0).cpu().numpy()
        assert "pair" not in nat.last_launch()[0] and np.abs(y1 - y0).max() <= STEP_TOL * 12 * np.abs(y0).max()
        # 129 .. : one workgroup per protein by itself
        nat.pair(True)
        ld = LangevinDiffusion(diff, torch.from_numpy(np.concatenate([x0, x0[:40]])), 2, save_interval=2, t=20, temp_data=340, temp_sim=340,
                               dt=None, masses=masses, friction=1.0, verbose=False, seed=1)
        ld.sample()
        assert "pair" not in nat.last_launch()[0]
    finally:
        nat.pair(True)
