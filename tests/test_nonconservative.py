"""The non-conservative branch (first part): conservative=False makes
node_decoder a Linear(H, 3) force head and `forces = output` (models/graph_transformer.py:62-65,112-113):
a forward-only score op.  Golden vectors: tests/golden/make_golden_nc.py (reference classes, seeded
synthetic weights with a (3, H) decoder)."""
import numpy as np
import pytest
import torch

from oracle import reference_twin as twin
from oracle import synth

CFGS = ["ala2", "chignolin", "trp_cage", "villin"]


def params_for(cfg, decoder_scale=1.0):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    return synth.synth_gnn_params(N, H, L, seed=4321, decoder_scale=decoder_scale, decoder_out=3), (N, H, L)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# ------------------------------------------------------------------ CPU
@pytest.mark.parametrize("cfg", CFGS)
def test_twin_matches_reference(cfg, golden):
    g = golden(f"score_nc_{cfg}.npz")
    p, (N, H, L) = params_for(cfg)
    f = twin.score(twin.to_torch(p), torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), L, conservative=False).numpy()
    assert np.array_equal(f, g["forces32"])
    f64 = twin.score(twin.to_torch(p, torch.float64), torch.from_numpy(g["x"]).double(), torch.from_numpy(g["t"]).double(),
                     L, conservative=False).numpy()
    assert np.abs(f64 - g["forces64"]).max() < 1e-12


def test_weight_layout_of_the_force_head():
    import dff_amd
    from dff_amd import binding, weights
    p, (N, H, L) = params_for("chignolin")
    flat = weights.flatten_gnn_params(p, N, H, L, conservative=False)
    cfg = binding.DffConfig(N, H, L, 1000, 1, 0, 0, 0)
    assert flat.size == binding.load_library().dff_weight_count(cfg)
    cfg1 = binding.DffConfig(N, H, L, 1000, 1, 0, 0, 1)
    assert flat.size == binding.load_library().dff_weight_count(cfg1) + 2 * H + 2
    with pytest.raises(ValueError):
        weights.flatten_gnn_params(p, N, H, L, conservative=True)      # (3, H) decoder in a conservative model
    assert dff_amd is not None


# ------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _model(cfg, decoder_scale=1.0):
    from dff_amd.score import GraphTransformer
    p, (N, H, L) = params_for(cfg, decoder_scale)
    return GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                            use_distances=False, conservative=False, state_dict=p), (N, H, L)


@gpu
@pytest.mark.parametrize("cfg", CFGS)
def test_forces_vs_reference(cfg, golden):
    g = golden(f"score_nc_{cfg}.npz")
    model, (N, H, L) = _model(cfg)
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    variants = [("default", lambda: None)]
    if N <= 16:
        variants += [("4 waves", lambda: model.native.small_waves(4)), ("generic", lambda: model.native.force_generic(True)),
                     ("3 per workgroup", lambda: model.native.set_group(3 if N == 5 else 1))]
    for name, setup in variants:
        model.native.small_waves(0); model.native.force_generic(False); model.native.set_group(0)
        setup()
        f = model(x, None, t).cpu().numpy()
        r64, a32 = rel(f, g["forces64"]), np.abs(f - g["forces32"]).max() / np.abs(g["forces32"]).max()
        print(f"{cfg} [{name}] {model.native.last_launch()[0]}: rel64 {r64:.2e} abs32 {a32:.2e}")
        assert r64 <= 1e-5 and a32 <= 1e-4, (cfg, name)
    model.native.small_waves(0); model.native.force_generic(False); model.native.set_group(0)
    with pytest.raises(ValueError):
        model(x, None, t, return_energy=True)


@gpu
def test_p_sample_and_langevin_with_the_force_head(golden):
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    model, (N, H, L) = _model("chignolin")
    diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=3.0)
    g = golden("psample_nc_chignolin.npz")
    for t in (500, 0):
        y = diff.p_sample(torch.from_numpy(g[f"x_{t}"]).cuda(), torch.full((3,), t, dtype=torch.long, device="cuda"),
                          noise=torch.from_numpy(g[f"noise_{t}"]).cuda())
        np.testing.assert_allclose(y.cpu().numpy(), g[f"y_{t}"], rtol=2e-5, atol=2e-5 * np.abs(g[f"y_{t}"]).max())
    # the fused reverse loop agrees with stepping p_sample by hand (same noise)
    x = torch.from_numpy(g["x_500"]).cuda()
    nz = torch.from_numpy(synth.normal((3, 3, N, 3), 5, 5).astype(np.float32)).cuda()
    fused = diff.p_sample_loop_from(x.clone(), 500, t_end=498, noises=nz).cpu().numpy()
    xs = x.clone()
    for k, t in enumerate((500, 499, 498)):
        xs = diff.p_sample(xs, torch.full((3,), t, dtype=torch.long, device="cuda"), noise=nz[k])
        xs = xs - xs.mean(1, keepdim=True)
    np.testing.assert_allclose(fused, xs.cpu().numpy(), rtol=1e-4, atol=1e-4 * np.abs(fused).max())

    g = golden("langevin_nc_chignolin.npz")
    model2, _ = _model("chignolin", decoder_scale=1e-2)
    diff2 = GaussianDiffusion(model2, num_atoms=N, timesteps=1000, norm_factor=3.0)
    ld = LangevinDiffusion(diff2, torch.from_numpy(g["x0"]), int(g["K"]), save_interval=int(g["save_interval"]), t=20,
                           diffusion_steps=1000, temp_data=340, temp_sim=340, dt=None, masses=[12.0] * N, friction=1.0,
                           kb="consistent", verbose=False)
    traj = ld.sample(noises=torch.from_numpy(g["noise"])).numpy()
    assert traj.shape == g["frames"].shape
    np.testing.assert_allclose(traj, g["frames"], rtol=2e-4, atol=2e-4 * np.abs(g["frames"]).max())
