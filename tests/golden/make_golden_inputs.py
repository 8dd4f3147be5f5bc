#!/usr/bin/env python3
"""Golden vectors of the OTHER input branches of the score net (models/graph_transformer.py:53-58,
99-102,116-140): use_intrinsic_coords / use_distances / use_abs_coords in the combinations
main_train.py can produce (its defaults are intrinsic=False, distances=True, abs=True), by running the
REFERENCE classes on seeded synthetic weights.

    cd /tmp && python /root/repo/tests/golden/make_golden_inputs.py

Writes score_in_<cfg>_<i><d><a>.npz: x, t -> forces (float32 and float64 runs), energy.  Data only.
Also checks oracle/kernel_model_gen.py (float64 factorised model + hand-written VJP) against the
reference's float64 run.
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

from oracle import kernel_model_gen as kg  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)
COMBOS = [(0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 0)]
SEED = 2468


def build(cfg, intr, dist, ab, dtype):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    nf = 3 * intr + dist
    p = synth.synth_gnn_params(N, H, L, seed=SEED, node_in=N + 1 + 3 * ab, edge_in=nf)
    g = GraphTransformer(N, hidden_nf=H, device="cpu", n_layers=L, use_intrinsic_coords=bool(intr),
                         use_abs_coords=bool(ab), use_distances=bool(dist), conservative=True)
    res = g.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    g.eval()
    return (g.double() if dtype == torch.float64 else g), p, (N, H, L)


def main():
    for cfg in ("chignolin", "trp_cage"):
        for (intr, dist, ab) in COMBOS:
            g32, p, (N, H, L) = build(cfg, intr, dist, ab, torch.float32)
            g64, _, _ = build(cfg, intr, dist, ab, torch.float64)
            x = (synth.normal((3, N, 3), 999, 7) * 1.2).astype(np.float32) + np.array([0.3, -0.2, 0.1], np.float32)
            t = np.array([0.005, 0.02, 0.5], np.float32)
            h = torch.eye(N)
            f32 = g32(torch.from_numpy(x), h, torch.from_numpy(t)).detach().numpy()
            e32 = g32(torch.from_numpy(x), h, torch.from_numpy(t), return_energy=True).detach().numpy()
            f64 = g64(torch.from_numpy(x).double(), h.double(), torch.from_numpy(t).double()).detach().numpy()
            fk, ek = kg.score(p, x, t, L, bool(intr), bool(dist), bool(ab))
            rel32 = np.linalg.norm(f32 - f64) / np.linalg.norm(f64)
            print(f"{cfg} intr={intr} dist={dist} abs={ab}: model-vs-ref64 max|d| = {np.abs(fk - f64).max():.2e}  "
                  f"ref32-vs-ref64 rel = {rel32:.2e}  max|F| = {np.abs(f64).max():.3f}")
            assert np.abs(fk - f64).max() < 1e-11 * max(1.0, np.abs(f64).max())
            np.savez(os.path.join(OUT, f"score_in_{cfg}_{intr}{dist}{ab}.npz"), x=x, t=t, forces32=f32, forces64=f64,
                     energy32=e32, flags=np.array([intr, dist, ab]), seed=SEED)
    print("done")


if __name__ == "__main__":
    main()
