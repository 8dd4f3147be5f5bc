#!/usr/bin/env python3
"""Race screen at the BASELINE sizes: every shipped architecture, 200 Langevin steps at batch 256 (protein G 128),
four times from the same state and seed -- the saved frames must be bit-identical (the kernels have no atomics; the
PAIR variants exchange partial sums between two workgroups in a fixed order: any difference is a missing barrier, an
LDS / stash overrun or a hand-off that let a stale tile through)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
import synth_weights as synth
for cfg, P in (("villin", 256), ("protein_g", 128), ("trp_cage", 256), ("bba", 256), ("chignolin", 256),
               ("villin", 128), ("trp_cage", 128), ("bba", 128), ("protein_g", 60),   # 128 / 60: the two-workgroups-per-protein variants
               ("ala2", 256), ("ala2", 383), ("ala2", 128), ("ala2", 768)):          # ... on groups of two / three / one protein; three per workgroup
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
    diff = GaussianDiffusion(model, num_atoms=N, norm_factor=3.0)
    x0 = torch.randn(P, N, 3, generator=torch.Generator().manual_seed(1)); x0 = (x0 - x0.mean(1, keepdim=True)) * 3.0
    outs = []
    for rep in range(4):
        ld = LangevinDiffusion(diff, x0, 200, save_interval=50, t=20, temp_data=340, temp_sim=340, dt=None,
                               masses=[12.0] * N, friction=1.0, verbose=False, seed=3)
        outs.append(torch.from_numpy(ld.simulate()).clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(cfg, P, model.native.last_launch()[0], "pair status", model.native.pair_status(), "4 repeats bit-identical:", same, "finite:", bool(torch.isfinite(outs[0]).all()), flush=True)
    assert same
