#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the REFERENCE.

Run once in the build container (the only place /root/reference exists):

    cd /tmp && python /root/repo/tests/golden/make_golden.py

It imports the reference's own modules (models/graph_transformer.py, models/ddpm.py,
dynamics/langevin.py, dynamics/langevin_cgnet.py, utils.py) with ``mdtraj`` stubbed
(utils.py:5 imports it at module top; nothing on the hot path uses it), loads the
deterministic synthetic weights of oracle/synth.py into the reference's module classes, and
records inputs + outputs as small .npz files.  Only DATA is written: no reference source
travels.  It also prints how far oracle/reference_twin.py is from the reference on every
vector (expected: bit-identical or ~1e-7).

Vectors:
  G1 score_<cfg>.npz   x, t_norm -> energy, forces (float32 run and float64 run)
  G2 layers_chignolin.npz  per-layer intermediates of the float32 run
  G3 psample_<cfg>.npz (x_t, t, noise) -> x_{t-1} for t in {999, 500, 1, 0}
  G4 ploop_chignolin.npz   last 5 reverse steps with clamp + centring
  G5 langevin_<cfg>_<variant>.npz  x0, noise[0:K] -> frames, KE, x_K, v_K
  G6 constants.npz     schedule known answers and Langevin unit constants
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.modules["mdtraj"] = types.ModuleType("mdtraj")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

from models.graph_transformer import GraphTransformer  # noqa: E402  (reference)
from models.ddpm import GaussianDiffusion  # noqa: E402  (reference)
from dynamics.langevin import LangevinDiffusion, temp_dict  # noqa: E402  (reference)
from utils import center_zero  # noqa: E402  (reference)

from oracle import reference_twin as twin  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)

NORM_STD = {"chignolin": 3.113133430480957, "villin": 6.082900047302246,
            "protein_g": 6.354289531707764, "ala2": 0.9449278712272644,
            "trp_cage": 5.08211088180542, "bba": 6.294918537139893}


def build_reference(cfg, seed=1234, decoder_scale=1.0, dtype=torch.float32):
    mol, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    gnn = GraphTransformer(N, hidden_nf=H, device="cpu", n_layers=L, use_intrinsic_coords=True,
                           use_abs_coords=False, use_distances=False, conservative=True)
    params = synth.synth_gnn_params(N, H, L, seed=seed, decoder_scale=decoder_scale)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    missing = gnn.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ddpm = GaussianDiffusion(model=gnn, features=torch.eye(N), num_atoms=N, timesteps=1000,
                             norm_factor=NORM_STD[cfg], loss_weights="higheruntil_100")
    ddpm.eval()
    if dtype == torch.float64:
        ddpm = ddpm.double()
    return ddpm, params, (N, H, L)


def inputs(cfg, B, stream):
    N = synth.SHIPPED_CONFIGS[cfg][1]
    x = synth.normal((B, N, 3), 777, stream).astype(np.float32)
    return x


def report(tag, a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    print(f"  twin-vs-ref {tag:28s} max|d|={np.abs(a - b).max():.3e}  max|ref|={np.abs(b).max():.3e}")


def g1_score():
    for cfg in ("ala2", "chignolin", "trp_cage", "bba", "villin", "protein_g"):
        ddpm, params, (N, H, L) = build_reference(cfg)
        ddpm64, _, _ = build_reference(cfg, dtype=torch.float64)
        x = inputs(cfg, 3, 1)  # NOT centred: the op centres internally
        x = x + np.array([0.3, -0.2, 0.1], np.float32)
        t = np.array([0.005, 0.02, 0.5], np.float32)
        xt, tt = torch.from_numpy(x), torch.from_numpy(t)
        h = torch.eye(N)
        f32 = ddpm.model(xt, h, tt)
        e32 = ddpm.model(xt, h, tt, return_energy=True).detach()
        f64 = ddpm64.model(xt.double(), h.double(), tt.double())
        e64 = ddpm64.model(xt.double(), h.double(), tt.double(), return_energy=True).detach()
        x1 = inputs(cfg, 1, 2)
        t1 = np.array([0.02], np.float32)
        f1 = ddpm.model(torch.from_numpy(x1), h, torch.from_numpy(t1))
        np.savez(os.path.join(OUT, f"score_{cfg}.npz"), x=x, t=t, forces32=f32.numpy(),
                 energy32=e32.numpy(), forces64=f64.numpy(), energy64=e64.numpy(),
                 x1=x1, t1=t1, forces1=f1.numpy(), seed=1234, n_params=synth.count_params(N, H, L))
        p = twin.to_torch(params)
        tf, te = twin.score(p, xt, tt, L, return_energy=True)
        print(cfg, "params", synth.count_params(N, H, L), "ref params",
              sum(v.numel() for v in ddpm.model.parameters()))
        report("forces32", tf, f32)
        report("energy32", te, e32)
        p64 = twin.to_torch(params, torch.float64)
        report("forces64", twin.score(p64, xt.double(), tt.double(), L), f64)
        print(f"  ref32-vs-ref64 rel = {np.linalg.norm(f32.numpy() - f64.numpy()) / np.linalg.norm(f64.numpy()):.3e}")


def g2_layers():
    cfg = "chignolin"
    ddpm, params, (N, H, L) = build_reference(cfg)
    x = inputs(cfg, 2, 3)
    t = np.array([0.02, 0.5], np.float32)
    caps = {}
    hooks = []
    for l, (attn_block, ff_block) in enumerate(ddpm.model.graphtransformer.layers):
        hooks.append(attn_block[0].register_forward_hook(
            lambda m, i, o, l=l: caps.__setitem__(f"l{l}.attn_out", o.detach().numpy())))
        hooks.append(attn_block[1].register_forward_hook(
            lambda m, i, o, l=l: caps.__setitem__(f"l{l}.nodes1", o.detach().numpy())))
        hooks.append(ff_block[0].register_forward_hook(
            lambda m, i, o, l=l: caps.__setitem__(f"l{l}.ff", o.detach().numpy())))
        hooks.append(ff_block[1].register_forward_hook(
            lambda m, i, o, l=l: caps.__setitem__(f"l{l}.nodes2", o.detach().numpy())))
    f = ddpm.model(torch.from_numpy(x), torch.eye(N), torch.from_numpy(t))
    for hk in hooks:
        hk.remove()
    np.savez(os.path.join(OUT, "layers_chignolin.npz"), x=x, t=t, forces=f.numpy(), **caps)
    inter = {}
    xc = twin.center_zero(torch.from_numpy(x))
    twin.energy(twin.to_torch(params), xc, torch.from_numpy(t), L, intermediates=inter)
    for k, v in caps.items():
        report(k, inter[k], v)


def g3_psample():
    for cfg in ("chignolin", "ala2"):
        ddpm, params, (N, H, L) = build_reference(cfg)
        sched = twin.make_schedule(1000)
        rec = {}
        p = twin.to_torch(params)
        for t in (999, 500, 1, 0):
            x = center_zero(torch.from_numpy(inputs(cfg, 3, 10 + t)))
            if t < 999:  # plausible magnitude for a partly denoised sample
                x = x * 0.7
            tt = torch.full((3,), t, dtype=torch.long)
            torch.manual_seed(4000 + t)
            y = ddpm.p_sample(x, tt)
            torch.manual_seed(4000 + t)
            noise = torch.randn_like(x)  # the score net draws nothing: same bits
            rec[f"x_{t}"] = x.numpy()
            rec[f"noise_{t}"] = noise.numpy()
            rec[f"y_{t}"] = y.numpy()
            report(f"p_sample t={t}", twin.p_sample(p, sched, x, t, noise, L), y)
        np.savez(os.path.join(OUT, f"psample_{cfg}.npz"), **rec)


def g4_ploop():
    cfg = "chignolin"
    ddpm, params, (N, H, L) = build_reference(cfg)
    x = center_zero(torch.from_numpy(inputs(cfg, 4, 30))) * 0.5
    x[1, 2, 0] = 2500.0  # force the +-1000 clamp (ddpm.py:248-250) on the first step
    x = center_zero(x)
    noises = torch.from_numpy(synth.normal((5, 4, N, 3), 999, 31).astype(np.float32))
    mol = x.clone()
    import warnings
    for k, i in enumerate(range(4, -1, -1)):
        # body of GaussianDiffusion.p_sample_loop (ddpm.py:244-251) driven by the reference's
        # own p_sample; the randn_like draw is replaced by seeding so that noise[k] is known.
        torch.manual_seed(5000 + k)
        noises[k] = torch.randn_like(mol)
        torch.manual_seed(5000 + k)
        mol = ddpm.p_sample(mol, torch.full((4,), i, dtype=torch.long))
        if (mol.max() > 1000) or (mol.min() < -1000):
            warnings.warn("Large molecule encountered in sampling")
            mol = torch.clamp(mol, min=-1000, max=1000)
        mol = center_zero(mol)
    np.savez(os.path.join(OUT, "ploop_chignolin.npz"), x5=x.numpy(), noises=noises.numpy(),
             x0=mol.numpy())
    report("p_sample_loop 5 steps", twin.p_sample_loop(twin.to_torch(params), twin.make_schedule(), x,
                                                       noises, 4, L), mol)


def g5_langevin():
    import contextlib
    import io
    variants = [
        ("chignolin", 20, 10, 5, 1.0, None),
        ("chignolin", 20, 1, 1, 1.0, None),
        ("chignolin", 20, 20, 1, None, None),
        ("ala2", 8, 10, 5, 1.0, None),
        ("villin", 5, 4, 2, 1.0, None),
        ("chignolin", 20, 10, 10, 1.0, 2e-3),
    ]
    for vi, (cfg, tlev, K, save, friction, dt) in enumerate(variants):
        mol, N, H, L = synth.SHIPPED_CONFIGS[cfg]
        ddpm, params, _ = build_reference(cfg, decoder_scale=1e-2)
        P = 3
        init = center_zero(torch.from_numpy(inputs(cfg, P, 50 + vi))) * NORM_STD[cfg]
        masses = [12.8] * N if "alanine" in mol else [12.0] * N
        temp = temp_dict[mol.upper()]
        with contextlib.redirect_stdout(io.StringIO()):
            ld = LangevinDiffusion(ddpm, init, K, save_interval=save, t=tlev, diffusion_steps=1000,
                                   temp_data=temp, temp_sim=temp, dt=dt, masses=masses,
                                   friction=friction, kb="consistent")
            torch.manual_seed(6000 + vi)
            traj = ld.sample()
        torch.manual_seed(6000 + vi)
        noises = torch.stack([torch.randn(P, N, 3) for _ in range(K)], 0)
        sim = ld.sim
        rec = dict(init=init.numpy(), noises=noises.numpy(), traj=traj.numpy(),
                   x_last=sim.x_old.detach().numpy(), t_level=tlev, K=K, save=save,
                   friction=-1.0 if friction is None else friction, dt=sim.dt,
                   temp=temp, masses=np.array(masses, np.float32), norm=NORM_STD[cfg])
        if friction is not None:
            rec["v_last"] = sim.v_old.detach().numpy()
            rec["ke"] = np.asarray(sim.kinetic_energies)
        name = f"langevin_{cfg}_{vi}.npz"
        np.savez(os.path.join(OUT, name), **rec)
        sched = twin.make_schedule()
        c = twin.langevin_constants(NORM_STD[cfg], tlev, sched, temp, temp, masses, friction, dt)
        fr, ke, xl, vl = twin.simulate(twin.to_torch(params), init / NORM_STD[cfg], noises, masses, c, L, save)
        report(name, fr.reshape(-1, N, 3) * NORM_STD[cfg], traj)
        report(name + " x_last", xl, sim.x_old.detach())


def g6_constants():
    ddpm, _, _ = build_reference("chignolin")
    rec = {k: getattr(ddpm, k).numpy() for k in (
        "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
        "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
        "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
        "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")}
    sched = twin.make_schedule()
    for k, v in rec.items():
        report("sched " + k, sched[k], v)
    np.savez(os.path.join(OUT, "constants.npz"), **rec)


if __name__ == "__main__":
    with torch.no_grad():
        g6_constants()
    g1_score()
    g2_layers()
    g3_psample()
    g4_ploop()
    g5_langevin()
    print("golden vectors written to", OUT)
