#!/usr/bin/env python3
"""Per-step timing of the fused kernel on the other BASELINE configs (not the bench.py headline):
Langevin step and DDPM reverse step, for each shipped architecture at a given batch."""
import argparse, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
import synth_weights as synth
MF = {"ala2": 11.27, "chignolin": 22.00, "trp_cage": 102.97, "bba": 107.19, "villin": 190.40, "protein_g": 327.47}
ap = argparse.ArgumentParser(); ap.add_argument("--cfgs", default="ala2,chignolin,trp_cage,bba,villin,protein_g")
ap.add_argument("--P", type=int, default=256); ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--flags", default="100", help="use_intrinsic_coords, use_distances, use_abs_coords as three digits (shipped: 100; main_train.py default: 011)")
ap.add_argument("--group", type=int, default=0, help="proteins per workgroup (0 = the library's choice)")
a = ap.parse_args()
out = []
for cfg in a.cfgs.split(","):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    P = a.P if cfg != "protein_g" else min(a.P, 128)
    intr, dist, ab = (int(ch) for ch in a.flags)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=bool(intr), use_abs_coords=bool(ab),
                             use_distances=bool(dist), conservative=True,
                             state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=1e-2, node_in=N + 1 + 3 * ab,
                                                               edge_in=(3 * intr + dist) or 1))
    if a.group:
        model.native.set_group(a.group)
    diff = GaussianDiffusion(model, num_atoms=N, norm_factor=3.0)
    x0 = torch.randn(P, N, 3); x0 = (x0 - x0.mean(1, keepdim=True)) * 3.0
    def lang():
        ld = LangevinDiffusion(diff, x0, a.steps, save_interval=a.steps, t=20, temp_data=340, temp_sim=340, dt=None,
                               masses=[12.0] * N, friction=1.0, verbose=False)
        ld.simulate()
    def ddpm():
        diff.p_sample_loop_from(x0 / 3.0, 500, 500 - a.steps + 1)
    res = {"cfg": cfg, "group": a.group, "flags": a.flags, "N": N, "H": H, "L": L, "P": P}
    for name, fn in (("langevin", lang), ("ddpm", ddpm)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        res[name + "_us_per_step"] = 1e6 * dt
        res[name + "_frac_fp32_roof"] = MF[cfg] * 1e6 * P / dt / 157.3e12
    res["kernel"] = model.native.last_launch()[0]
    res["iid_samples_per_s"] = P / (1000 * res["ddpm_us_per_step"] * 1e-6)
    print(json.dumps(res), flush=True)
