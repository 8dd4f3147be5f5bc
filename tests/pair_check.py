#!/usr/bin/env python3
"""Two workgroups per protein (PAIR variants) against one: step time at a batch that leaves half the CUs idle, and
agreement of the forces with the oracle.  usage: tests/pair_check.py [cfg[:P] ...]  (P: trajectories, default 128)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
import synth_weights as synth
from oracle import reference_twin as twin
for cfg in (sys.argv[1:] or ["protein_g", "villin", "trp_cage"]):
    cfg, _, P = cfg.partition(":")
    P = int(P or 128)
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1.0)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=params)
    x = synth.normal((5, N, 3), 1, 1).astype(np.float32); t = np.array([0.005, 0.02, 0.5, 0.1, 0.3], np.float32)
    ref = twin.score(twin.to_torch(params), torch.from_numpy(x), torch.from_numpy(t), L).numpy()
    for on in (True, False):
        model.native.pair(on)
        f = model(torch.from_numpy(x).cuda(), torch.eye(N), torch.from_numpy(t).cuda()).cpu().numpy()
        print(cfg, "pair", on, model.native.last_launch(), "status", model.native.pair_status(), "rel err vs oracle %.2e" % (np.linalg.norm(f - ref) / np.linalg.norm(ref)))
    p2 = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
    m2 = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=p2)
    diff = GaussianDiffusion(m2, num_atoms=N, norm_factor=3.0)
    x0 = torch.randn(P, N, 3); x0 = (x0 - x0.mean(1, keepdim=True)) * 3.0
    for on in (True, False):
        m2.native.pair(on)
        for rep in range(2):
            ld = LangevinDiffusion(diff, x0, 400, save_interval=400, t=5, temp_data=350, temp_sim=350, dt=None, masses=[12.0] * N, friction=1.0, verbose=False, seed=3)
            torch.cuda.synchronize(); t0 = time.perf_counter(); tr = ld.simulate(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(cfg, f"P={P} pair", on, m2.native.last_launch()[:2], "status", m2.native.pair_status(), f"{1e6*dt/400:.1f} us/step")
