#!/usr/bin/env python3
"""Golden vectors of the NON-conservative branch (conservative=False: node_decoder = Linear(H, 3),
forces = its output, models/graph_transformer.py:62-65,112-113), by running the REFERENCE classes.

    cd /tmp && python /root/repo/tests/golden/make_golden_nc.py

Writes score_nc_<cfg>.npz (x, t -> forces, float32 and float64 runs), psample_nc_chignolin.npz
(one reverse step at t = 500 and t = 0 on recorded noise) and langevin_nc_chignolin.npz (10 BAOAB steps
on recorded noise).  Data only.  Also checks oracle/reference_twin.py against the reference.
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
from dynamics.langevin import LangevinDiffusion  # noqa: E402  (reference)

from oracle import reference_twin as twin  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def build(cfg, dtype=torch.float32, decoder_scale=1.0):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    gnn = GraphTransformer(N, hidden_nf=H, device="cpu", n_layers=L, use_intrinsic_coords=True,
                           use_abs_coords=False, use_distances=False, conservative=False)
    params = synth.synth_gnn_params(N, H, L, seed=4321, decoder_scale=decoder_scale, decoder_out=3)
    res = gnn.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ddpm = GaussianDiffusion(model=gnn, features=torch.eye(N), num_atoms=N, timesteps=1000, norm_factor=3.0,
                             loss_weights="higheruntil_100")
    ddpm.eval()
    if dtype == torch.float64:
        ddpm = ddpm.double()
    return ddpm, params, (N, H, L)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def main():
    for cfg in ("ala2", "chignolin", "trp_cage", "villin"):
        ddpm, params, (N, H, L) = build(cfg)
        ddpm64, _, _ = build(cfg, torch.float64)
        x = synth.normal((3, N, 3), 888, 1).astype(np.float32) + np.array([0.3, -0.2, 0.1], np.float32)
        t = np.array([0.005, 0.02, 0.5], np.float32)
        h = torch.eye(N)
        with torch.no_grad():
            f32 = ddpm.model(torch.from_numpy(x), h, torch.from_numpy(t)).numpy()
            f64 = ddpm64.model(torch.from_numpy(x).double(), h.double(), torch.from_numpy(t).double()).numpy()
        tw = twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L, conservative=False).numpy()
        print(f"{cfg}: twin-vs-ref max|d| = {np.abs(tw - f32).max():.3e}   ref32-vs-ref64 rel = {rel(f32, f64):.3e}")
        np.savez(os.path.join(OUT, f"score_nc_{cfg}.npz"), x=x, t=t, forces32=f32, forces64=f64)

    # one reverse step on recorded noise (the score net consumes no RNG in eval)
    ddpm, params, (N, H, L) = build("chignolin")
    out = {}
    for tt in (500, 0):
        xt = torch.from_numpy(synth.normal((3, N, 3), 888, 10 + tt).astype(np.float32))
        xt = xt - xt.mean(1, keepdim=True)
        tv = torch.full((3,), tt, dtype=torch.long)
        torch.manual_seed(7 + tt)
        with torch.no_grad():
            y = ddpm.p_sample(xt, tv)
        torch.manual_seed(7 + tt)
        noise = torch.randn_like(xt)
        out[f"x_{tt}"], out[f"noise_{tt}"], out[f"y_{tt}"] = xt.numpy(), noise.numpy(), y.numpy()
    np.savez(os.path.join(OUT, "psample_nc_chignolin.npz"), **out)

    # 10 BAOAB steps driven by the force head, noise recorded by replaying the CPU generator
    ddpm, params, (N, H, L) = build("chignolin", decoder_scale=1e-2)
    x0 = synth.normal((4, N, 3), 888, 50).astype(np.float32) * 3.0
    K = 10
    torch.manual_seed(99)
    ld = LangevinDiffusion(ddpm, torch.from_numpy(x0), K, save_interval=5, t=20, diffusion_steps=1000, temp_data=340,
                           temp_sim=340, dt=None, masses=[12.0] * N, friction=1.0, kb="consistent")
    frames = ld.sample().numpy()
    torch.manual_seed(99)
    noise = np.stack([torch.randn(4, N, 3).numpy() for _ in range(K)])
    np.savez(os.path.join(OUT, "langevin_nc_chignolin.npz"), x0=x0, noise=noise, frames=frames, K=K, save_interval=5)
    print("done")


if __name__ == "__main__":
    main()
