#!/usr/bin/env python3
"""Round-4 golden vectors, recorded by running the REFERENCE classes (mdtraj stubbed):

    cd /tmp && python /root/repo/tests/golden/make_golden_r04.py

hidden_nf = 256 -- no shipped checkpoint uses it, but it is the size of the reference's own smoke test
(models/graph_transformer.py:332-359: num_beads 10, hidden_nf 256, n_layers 5, conservative False):
  score_h256_smoke.npz   that exact architecture (force head), x, t -> forces (float32 and float64 runs)
  score_h256_cons.npz    a conservative model at 20 beads, 2 layers (two row tiles): forces32 / 64 and energy32
  langevin_h256_cons.npz 6 BAOAB steps of the conservative model on recorded noise
Data only.  Also checks oracle/reference_twin.py against the reference (bit-identical).
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
SHAPES = {"smoke": (10, 256, 5, False), "cons": (20, 256, 2, True)}   # N, H, L, conservative


def params_for(name, decoder_scale=1.0):
    N, H, L, cons = SHAPES[name]
    return synth.synth_gnn_params(N, H, L, seed=2560 + N, decoder_scale=decoder_scale, decoder_out=1 if cons else 3)


def build(name, dtype=torch.float32, decoder_scale=1.0):
    N, H, L, cons = SHAPES[name]
    gnn = GraphTransformer(N, hidden_nf=H, device="cpu", n_layers=L, use_intrinsic_coords=True,
                           use_abs_coords=False, use_distances=False, conservative=cons)
    params = params_for(name, decoder_scale)
    res = gnn.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ddpm = GaussianDiffusion(model=gnn, features=torch.eye(N), num_atoms=N, timesteps=1000, norm_factor=3.0,
                             loss_weights="higheruntil_100")
    ddpm.eval()
    if dtype == torch.float64:
        ddpm = ddpm.double()
    return ddpm, params


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def main():
    for name, (N, H, L, cons) in SHAPES.items():
        ddpm, params = build(name)
        ddpm64, _ = build(name, torch.float64)
        x = synth.normal((3, N, 3), 256, N).astype(np.float32) + np.array([0.3, -0.2, 0.1], np.float32)
        t = np.array([0.005, 0.02, 0.5], np.float32)
        h = torch.eye(N)
        xt, tt = torch.from_numpy(x), torch.from_numpy(t)
        out = dict(x=x, t=t)
        if cons:
            out["forces32"] = ddpm.model(xt, h, tt).detach().numpy()
            out["forces64"] = ddpm64.model(xt.double(), h.double(), tt.double()).detach().numpy()
            out["energy32"] = ddpm.model(xt, h, tt, return_energy=True).detach().numpy()
        else:
            with torch.no_grad():
                out["forces32"] = ddpm.model(xt, h, tt).numpy()
                out["forces64"] = ddpm64.model(xt.double(), h.double(), tt.double()).numpy()
        tw = twin.score(twin.to_torch(params), xt, tt, L, conservative=cons).numpy()
        print(f"{name}: twin-vs-ref max|d| = {np.abs(tw - out['forces32']).max():.3e}   ref32-vs-ref64 rel = {rel(out['forces32'], out['forces64']):.3e}")
        assert np.array_equal(tw, out["forces32"])
        np.savez(os.path.join(OUT, f"score_h256_{name}.npz"), **out)

    N, H, L, cons = SHAPES["cons"]
    ddpm, params = build("cons", decoder_scale=1e-2)
    x0 = synth.normal((4, N, 3), 256, 50).astype(np.float32) * 3.0
    K = 6
    torch.manual_seed(256)
    ld = LangevinDiffusion(ddpm, torch.from_numpy(x0), K, save_interval=3, t=20, diffusion_steps=1000, temp_data=340,
                           temp_sim=340, dt=None, masses=[12.0] * N, friction=1.0, kb="consistent")
    frames = ld.sample().numpy()
    torch.manual_seed(256)
    noise = np.stack([torch.randn(4, N, 3).numpy() for _ in range(K)])
    np.savez(os.path.join(OUT, "langevin_h256_cons.npz"), x0=x0, noise=noise, frames=frames, K=K, save_interval=3)
    print("done")


if __name__ == "__main__":
    main()
