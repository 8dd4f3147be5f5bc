#!/usr/bin/env python3
"""Random model shapes / batch shapes / magnitudes against the oracle twin (float64): forces (dff_score; all input branches, energy
and force heads, batches of 1 .. 900) and a few fused Langevin and reverse-DDPM steps on supplied noise (shipped branch, energy
head).  The bars are the tests': GUARD x the twin's own float32 distance for forces, STEP_TOL per step for the loops.
test_gpu_parity.py::test_random_shapes runs a short sweep; `python tests/fuzz_shapes.py [n_cases] [seed]` a long one (GPU box)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dff_amd  # noqa: F401,E402
from oracle import synth, reference_twin as twin  # noqa: E402
from dff_amd.score import GraphTransformer  # noqa: E402
from dff_amd.ddpm import GaussianDiffusion  # noqa: E402
from dff_amd.langevin import LangevinDiffusion  # noqa: E402

GUARD, GUARD_FP32, STEP_TOL = 2.0, 2.5, 5e-6   # tests/test_gpu_parity.py: the split engine's bar, the fp32-MFMA engine's, the loops'
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))  # noqa: E731


def run(n_cases, seed, log=print, only=None, hook=None):
    """Returns the number of cases outside the bars (each case is logged as one line).  `only`: run these case numbers of the
    sequence (the others just consume their random draws); `hook(case, tag, model, params, x, t, sub, f, flags, cons, L)`: called
    with a case's inputs and forces (tests/gen_outliers.py)."""
    rng = np.random.default_rng(seed)
    env = {k: os.environ.get(k) for k in ("DFF_SPLIT_BF16", "DFF_FOLD_KV")}
    bad = 0
    try:
        for case in range(n_cases):
            bad += not _case(case, rng, log, only, hook)
    finally:
        for k, v in env.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return bad


def _case(case, rng, log, only=None, hook=None):
    H = int(rng.choice([64, 96, 128]))
    N = int(rng.integers(2, (61 if H == 128 else 32) + 1))
    L = int(rng.integers(1, 5))
    B = int(rng.integers(1, 12)) if rng.integers(0, 4) else int(rng.integers(200, 900))   # (one in four: several launch rounds / groups)
    cons = bool(rng.integers(0, 6) > 0)
    dec = float(10.0 ** rng.uniform(-6, 4)); xs = float(10.0 ** rng.uniform(-1.3, 1.0))
    G = int(rng.choice([0, 0, 1, 2, 3]))
    split = bool(rng.integers(0, 4) > 0)
    os.environ["DFF_SPLIT_BF16"] = "1" if split else "0"
    os.environ["DFF_FOLD_KV"] = "1" if rng.integers(0, 3) > 0 else "0"
    flags = [(1, 0, 0), (1, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 0)][int(rng.integers(0, 8))]
    intr, dist, ab = flags
    shipped = flags == (1, 0, 0)
    if not shipped: dec, cons = 1.0, True
    params = synth.synth_gnn_params(N, H, L, seed=int(rng.integers(1, 1 << 30)), decoder_scale=dec, decoder_out=1 if cons else 3, node_in=N + 1 + 3 * ab, edge_in=(3 * intr + dist) or 1)
    tag = dict(case=case, H=H, N=N, L=L, B=B, dec=float("%.2g" % dec), xs=float("%.2g" % xs), G=G, split=int(split), fold=int(os.environ["DFF_FOLD_KV"]), flags="%d%d%d" % flags, cons=int(cons))
    fl = tuple(bool(v) for v in flags)
    if only is not None and case not in only:   # (the same draws as a case that runs)
        rng.uniform(0.0, 1.0, B)
        if B > 12: rng.choice(B, 8, replace=False)
        if shipped and cons: rng.integers(1, 60)
        return True
    try:
        model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=fl[0], use_abs_coords=fl[2], use_distances=fl[1],
                                 conservative=cons, state_dict=params)
        model.native.set_group(G)
        x = (synth.normal((B, N, 3), 11 + case, N) * xs).astype(np.float32)
        t = rng.uniform(0.0, 1.0, B).astype(np.float32)
        f = model.native.score(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        kn = model.native.last_launch()[0]
        sub = np.arange(B) if B <= 12 else np.sort(rng.choice(B, 8, replace=False))   # (a sample's forces depend on that sample only)
        xs_, ts_ = torch.from_numpy(x[sub]), torch.from_numpy(t[sub])
        r64ref = twin.score(twin.to_torch(params, torch.float64), xs_.double(), ts_.double(), L, conservative=cons, flags=fl).numpy()
        r32 = rel(twin.score(twin.to_torch(params), xs_, ts_, L, conservative=cons, flags=fl).numpy(), r64ref)
        r = rel(f[sub], r64ref)
        if not shipped and xs > 1.5 and r > 2.5 * max(r32, 4e-7):
            # Distance features far from the origin: single trajectories there are float32-UNSTABLE in the reference itself -- its own
            # float32 run lands 3e-3 from its float64 one on one host CPU and 1.1e-1 on another (profiles/r06/gen_outliers*.txt; a
            # 2^-22 relative change of x moves the float64 forces by 3e-3) -- so one float32 run is no yardstick.  Take the
            # reference's float32 band instead: the worst of its runs at x and at the neighbouring float32 inputs (one ulp either way,
            # forces compared with the float64 run at x).
            for direction in (np.inf, -np.inf):
                xn = torch.from_numpy(np.nextafter(x[sub], np.float32(direction)).astype(np.float32))
                r32 = max(r32, rel(twin.score(twin.to_torch(params), xn, ts_, L, conservative=cons, flags=fl).numpy(), r64ref))
            tag.update(band=1)
        if hook is not None: hook(case, tag, model, params, x, t, sub, f, fl, cons, L)
        # (other branches: the tests' 2e-5 at unit coordinates; distance features at |x| ~ 10 sigma are ill-conditioned in float32 --
        # the reference's own float32 run is then 1e-4 from its float64 one -- so the bar follows that distance there)
        # other input branches: the hot path's bar up to 1.5 sigma; beyond, distance features make the INPUT ill-conditioned -- the
        # reference's own float32 run is 1e-5 .. 7e-4 from its float64 one there and the same algebra in another summation order
        # (numpy float32) up to 2.4 x that (profiles/r06/gen_conditioning.txt): 6 x its distance
        gbar = (GUARD_FP32 if xs <= 1.5 else 6.0) * max(r32, 4e-7)
        ok = np.isfinite(f).all() and (r <= 1e-5 and r <= (GUARD if "split_" in kn else GUARD_FP32) * max(r32, 4e-7) if shipped else r <= gbar)
        tag.update(kernel=kn, rel=float("%.3g" % r), r32=float("%.3g" % r32))
        if not shipped or not cons:   # (the twin's integrator runs the shipped branch only: tests/test_input_branches.py covers the loops there)
            log(("ok   " if ok else "FAIL ") + json.dumps(tag))
            return bool(ok)
        # a few fused Langevin steps
        K, P, norm, tlev, temp = 4, min(B, 4), 3.0, int(rng.integers(1, 60)), 300
        diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=norm)
        x0 = synth.normal((P, N, 3), 43 + case, N).astype(np.float32)
        x0 = (x0 - x0.mean(1, keepdims=True)) * norm
        noises = synth.normal((K, P, N, 3), 44 + case, N).astype(np.float32)
        masses = [12.0] * N
        ld = LangevinDiffusion(diff, torch.from_numpy(x0), K, save_interval=2, t=tlev, diffusion_steps=1000, temp_data=temp, temp_sim=temp,
                               dt=None, masses=masses, friction=1.0, kb="consistent", verbose=False)
        traj = ld.sample(noises=torch.from_numpy(noises)).numpy().reshape(P, K // 2, N, 3)
        c = twin.langevin_constants(norm, tlev, twin.make_schedule(), temp, temp, masses, 1.0, None)
        fr, ke, xl, vl = twin.simulate(twin.to_torch(params, torch.float64), torch.from_numpy(x0).double() / norm, torch.from_numpy(noises).double(), masses, c, L, 2)
        ref = (fr * norm).numpy()
        el = float(np.abs(traj - ref).max() / max(np.abs(ref).max(), 1e-30))
        kl = model.native.last_launch()[0]
        tag.update(langevin_err=float("%.3g" % el), status=int(model.native.status()))
        if kl != kn: tag.update(kernel_l=kl)
        ok = ok and el <= STEP_TOL * K and model.native.status() == 0
        # ... and of the reverse-DDPM loop, t = K - 1 .. 0
        xd = synth.normal((P, N, 3), 45 + case, N).astype(np.float32)
        xd = (xd - xd.mean(1, keepdims=True)) * 0.6
        y = diff.p_sample_loop_from(torch.from_numpy(xd), K - 1, 0, noises=torch.from_numpy(noises)).cpu().numpy()
        refd = twin.p_sample_loop(twin.to_torch(params, torch.float64), {k: v.double() for k, v in twin.make_schedule().items()},
                                  torch.from_numpy(xd).double(), torch.from_numpy(noises).double(), K - 1, L).numpy()
        ed = float(np.abs(y - refd).max() / max(np.abs(refd).max(), 1e-30))
        tag.update(ddpm_err=float("%.3g" % ed))
        ok = ok and ed <= STEP_TOL * K and model.native.status() == 0
    except Exception as e:  # noqa: BLE001
        tag.update(error=repr(e)[:300]); ok = False
    log(("ok   " if ok else "FAIL ") + json.dumps(tag))
    return bool(ok)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    nbad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1, lambda m: print(m, flush=True))
    print(f"{n - nbad} / {n} ok")
