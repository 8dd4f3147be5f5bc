"""The two-piece fp16 split engine (split_f16 kernels) against operands chosen to hurt it, and against the fp32-MFMA
engine as a sampler.  `-m gpu`; every call goes through the C ABI.

An fp32 operand of a weight GEMM is carried as h + l'/2048 (two fp16 pieces rounded to nearest: 2^-22 per operand where
fp32 is exact) and a product keeps three of the four piece products.  On i.i.d. operands the piece roundings cancel along
a dot product (tests/test_gpu_parity.py, tests/fuzz_shapes.py: synthetic uniform weights).  Here:

* `test_fp16_engine_on_structured_operands`: weights on a bf16 grid, on an int8-dequantised grid, with same-sign
  residuals (w = fp16(w) (1 + 2^-12): every low piece has the sign of its high piece), LayerNorm rows with one entry of
  1e2 among entries of 1e-3 (the pieces are cut of the activations "as they are": 2^-35 absolute), energy heads of 1e-6
  and 1e4, and coordinates at the +-1000 clamp of models/ddpm.py:248-250 -- both engines against the twin in float64 at
  the bars of tests/test_gpu_parity.py, five architectures.
* `test_engines_agree_statistically`: 20 000 Langevin steps x 256 trajectories with the in-kernel Philox noise on either
  engine; the pairwise-distance histograms of the two ensembles (the library's own dff_pwd_hist) are no
  further apart -- Jensen-Shannon, evaluate/evaluators.py:251-270 -- than two seeds of ONE engine, and both thermostats
  hold equipartition (dynamics/langevin_cgnet.py:538-542).
* `test_engines_agree_statistically_iid`: the same comparison for the other sampler of the hot path -- reverse-DDPM chains
  (models/ddpm.py:195-254) over the last 300 noise levels, 4096 chains per engine and seed.
"""
import numpy as np
import pytest
import torch

from oracle import reference_twin as twin
from oracle import synth

pytestmark = pytest.mark.gpu

GUARD, GUARD_FP32 = 2.0, 2.5   # tests/test_gpu_parity.py
NORM_STD = {"chignolin": 3.113133430480957, "villin": 6.082900047302246}
TEMP = {"chignolin": 340, "villin": 360}


@pytest.fixture(scope="module")
def dff():
    import dff_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dff_amd.load_library()
    return dff_amd


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def bf16_grid(w):
    """float32 -> nearest bfloat16 (ties to even), as float32"""
    u = w.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def structured(params, kind):
    """The synthetic state dict with its Linear weights / LayerNorm affine parameters moved onto a structure."""
    out, ln = {}, 0
    for k, v in params.items():
        v = v.copy()
        linear = v.ndim == 2 and "proj" not in k          # (the gates' 1 x 3H projections are row-stage VALU work, not GEMMs)
        if kind == "bf16_grid" and linear:
            v = bf16_grid(v)
        elif kind == "int8_grid" and linear:              # symmetric per-tensor int8, dequantised
            s = np.abs(v).max() / 127.0
            v = (np.round(v / s) * s).astype(np.float32)
        elif kind == "same_sign_residual" and linear:     # h (1 + 2^-12): exact in fp32; w - fp16(w) has the sign of w everywhere
            h = v.astype(np.float16).astype(np.float32)
            v = (h * np.float32(1.0 + 2.0 ** -12)).astype(np.float32)
        elif kind == "ln_outlier" and k.endswith("norm.weight"):
            g = np.full_like(v, 1e-3)
            g[1::2] *= -1
            g[(7 * ln + 3) % v.size] = 100.0              # (1e3 would put sqrt(H) |gain| |W1 row|_1 past the engine's range guard:
            ln += 1                                       #  test_fp16_engine_steps_aside_for_models_out_of_its_range covers that side)
            v = g
        elif kind == "ln_outlier" and k.endswith("norm.bias"):
            v = (v * 1e-2).astype(np.float32)
        out[k] = v
    return out


STRUCTURES = ["bf16_grid", "int8_grid", "same_sign_residual", "ln_outlier", "decoder_1e-6", "decoder_1e4", "x_at_clamp"]


@pytest.mark.parametrize("cfg", ["chignolin", "trp_cage", "bba", "villin", "protein_g"])
@pytest.mark.parametrize("kind", STRUCTURES)
def test_fp16_engine_on_structured_operands(dff, cfg, kind, monkeypatch):
    from dff_amd.score import GraphTransformer
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    dec = {"decoder_1e-6": 1e-6, "decoder_1e4": 1e4}.get(kind, 1e-2)
    params = structured(synth.synth_gnn_params(N, H, L, seed=777, decoder_scale=dec), kind)
    x = synth.normal((5, N, 3), 21, 3)
    if kind == "x_at_clamp":                               # every coordinate on the clamp; one sample spread inside it
        x = np.sign(x) * 1000.0
        x[0] = np.clip(synth.normal((N, 3), 22, 3) * 300.0, -1000.0, 1000.0)
    x = (x - x.mean(1, keepdims=True)).astype(np.float32)
    t = np.array([0.0, 0.02, 0.3, 0.7, 0.999], np.float32)
    f32ref = twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy()
    f64 = twin.score(twin.to_torch(params, torch.float64), torch.from_numpy(x).double(), torch.from_numpy(t).double(), L).numpy()
    r32 = rel(f32ref, f64)
    # the two ill-conditioned structures (logits of 1e2 .. 1e3: the reference's own float32 run is 1e-5 .. 2e-3 from its
    # float64 one) are held to the relative bar only
    absbar = 5e-6 if kind not in ("ln_outlier", "x_at_clamp") else np.inf
    got = {}
    for split in (True, False):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        kname = model.native.last_launch()[0]
        assert ("split_f16" in kname) == split, kname
        assert np.isfinite(f).all() and model.native.status() == 0
        got[split] = rel(f, f64)
    print(f"{cfg} {kind}: rel(split_f16, f64)={got[True]:.3e} rel(fp32 MFMA, f64)={got[False]:.3e} rel(ref32, ref64)={r32:.3e} "
          f"|F|max={np.abs(f64).max():.3e}")
    assert got[True] <= absbar and got[True] <= GUARD * max(r32, 4e-7)
    assert got[False] <= absbar and got[False] <= GUARD_FP32 * max(r32, 4e-7)


@pytest.mark.parametrize("cfg,steps", [("chignolin", 20000), ("villin", 20000)])
def test_engines_agree_statistically(dff, cfg, steps, monkeypatch, tmp_path):
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.evaluate import PwdEvaluator
    from dff_amd.langevin import LangevinDiffusion
    from dff_amd.score import GraphTransformer
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    P, save, burn = 256, 250, 20
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
    init = torch.from_numpy(synth.normal((P, N, 3), 11, 4).astype(np.float32)) * NORM_STD[cfg]
    init = init - init.mean(1, keepdim=True)
    c = twin.langevin_constants(NORM_STD[cfg], 20, twin.make_schedule(), TEMP[cfg], TEMP[cfg], [12.0] * N, 1.0, None)
    expect_ke = 1.5 * N / c["beta"]
    runs = {}
    for split, seed in ((True, 101), (True, 202), (False, 101), (False, 202)):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=NORM_STD[cfg])
        ld = LangevinDiffusion(diff, init, n_timesteps=steps, save_interval=save, t=20, temp_data=TEMP[cfg], temp_sim=TEMP[cfg],
                               dt=None, masses=[12.0] * N, friction=1.0, verbose=False, seed=seed)
        traj = ld.sample()
        kname = model.native.last_launch()[0]
        assert ("split_f16" in kname) == split, kname
        nf = steps // save
        assert traj.shape == (P * nf, N, 3) and torch.isfinite(traj).all() and model.native.status() == 0
        ke = np.asarray(ld.kinetic_energies)                  # (P, nf)
        late = ke[:, burn:]
        # equipartition of the BAOAB thermostat: <KE> = 3 N / (2 beta) whatever the forces are (P x 60 frames: +-0.4 % at 1 sigma)
        assert abs(late.mean() / expect_ke - 1.0) < 0.03, (split, seed, late.mean(), expect_ke)
        runs[(split, seed)] = (traj.reshape(P, nf, N, 3)[:, burn:].reshape(-1, N, 3).contiguous(), late.mean() / expect_ke)
    js = {}
    for a, b in (((True, 101), (True, 202)), ((False, 101), (False, 202)), ((True, 101), (False, 101)), ((True, 202), (False, 202)),
                 ((True, 101), (False, 202))):
        ev = PwdEvaluator(runs[a][0], mol_name=cfg, offset=3, saved_ref=str(tmp_path / f"ref_{a[0]}_{a[1]}.pickle"))
        js[(a, b)] = float(ev.eval(runs[b][0]))
    same = [js[((True, 101), (True, 202))], js[((False, 101), (False, 202))]]
    cross = [js[((True, 101), (False, 101))], js[((True, 202), (False, 202))], js[((True, 101), (False, 202))]]
    print(f"{cfg}: JS(two seeds, split_f16)={same[0]:.3e} JS(two seeds, fp32 MFMA)={same[1]:.3e} JS(split_f16 vs fp32 MFMA)="
          + " ".join(f"{v:.3e}" for v in cross) + " KE/expected=" + " ".join(f"{runs[k][1]:.4f}" for k in runs))
    # two engines are two samples of one ensemble: no further apart than two seeds of one engine (each JS is an estimate from
    # 256 x 60 correlated frames: a quarter of slack)
    assert max(cross) <= 1.25 * max(same), (cross, same)


@pytest.mark.parametrize("cfg", ["chignolin", "villin"])
def test_engines_agree_statistically_iid(dff, cfg, monkeypatch, tmp_path):
    """The reverse-DDPM sampler (models/ddpm.py:195-254, in-kernel Philox noise) on either engine: 4096 chains per run over the last
    300 noise levels (a random-weight network is no denoiser: from t = T its chains run into the +-1000 clamp, from t = 300 they stay
    O(1)); the pairwise-distance histograms are no further apart between the engines than between two seeds of one."""
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.evaluate import PwdEvaluator
    from dff_amd.score import GraphTransformer
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    B, rounds, t0 = 256, 16, 300
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
    inits = []
    for r in range(rounds):
        x = torch.from_numpy(synth.normal((B, N, 3), 300 + r, 5).astype(np.float32))
        inits.append(x - x.mean(1, keepdim=True))
    runs = {}
    for split, seed in ((True, 11), (True, 22), (False, 11), (False, 22)):
        monkeypatch.setenv("DFF_SPLIT_BF16", "1" if split else "0")
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True, state_dict=params)
        diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=NORM_STD[cfg])
        outs = []
        for r in range(rounds):
            diff.seed(1000 * seed + r)
            outs.append(diff.p_sample_loop_from(inits[r], t0, 0).cpu() * NORM_STD[cfg])
        out = torch.cat(outs)
        kname = model.native.last_launch()[0]
        assert ("split_f16" in kname) == split, kname
        assert out.shape == (B * rounds, N, 3) and torch.isfinite(out).all() and model.native.status() == 0
        assert float(out.abs().max()) < 50.0 * NORM_STD[cfg] and not diff.last_clamped
        assert float(out.mean(1).abs().max()) < 1e-3 * NORM_STD[cfg]          # centred chains (models/ddpm.py:251-252)
        runs[(split, seed)] = out.contiguous()
    assert not torch.equal(runs[(True, 11)], runs[(True, 22)])                 # (the seed reaches the kernel's Philox key)
    js = {}
    for a, b in (((True, 11), (True, 22)), ((False, 11), (False, 22)), ((True, 11), (False, 11)), ((True, 22), (False, 22)),
                 ((True, 11), (False, 22))):
        ev = PwdEvaluator(runs[a], mol_name=cfg, offset=3, saved_ref=str(tmp_path / f"iid_ref_{a[0]}_{a[1]}.pickle"))
        js[(a, b)] = float(ev.eval(runs[b]))
    same = [js[((True, 11), (True, 22))], js[((False, 11), (False, 22))]]
    cross = [js[((True, 11), (False, 11))], js[((True, 22), (False, 22))], js[((True, 11), (False, 22))]]
    print(f"{cfg} iid: JS(two seeds, split_f16)={same[0]:.3e} JS(two seeds, fp32 MFMA)={same[1]:.3e} JS(split_f16 vs fp32 MFMA)="
          + " ".join(f"{v:.3e}" for v in cross))
    # same seed, two engines: the same draws, so the ensembles nearly coincide; different seeds: the sampling noise of 4096 chains
    # either way -- no pair of engines further apart than two seeds of one (a quarter of slack)
    assert max(cross) <= 1.25 * max(same), (cross, same)
