#!/usr/bin/env python3
"""Per-stage cycle breakdown of the fused kernel (workgroup 0), via dff_debug_profile.

Needs a library built with the stage ticks compiled in (-DDFF_PROF=1: `DFF_EXTRA_FLAGS=-DDFF_PROF=1 ./build.sh`, or
`./tools_exp.sh build prof -DDFF_PROF=1` and DFF_LIB_PATH=build/exp/prof/libdff_amd.so); the product build leaves them out."""
import argparse, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
import synth_weights as synth
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", default="chignolin"); ap.add_argument("--P", type=int, default=256)
ap.add_argument("--steps", type=int, default=250); ap.add_argument("--group", type=int, default=0)
ap.add_argument("--waves", default="0", help="comma list: whose view of the stages (<= 16-row kernel; wave 0 otherwise)")
a = ap.parse_args()
_, N, H, L = synth.SHIPPED_CONFIGS[a.cfg]
model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                         use_distances=False, conservative=True, state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
if a.group: model.native.set_group(a.group)
diff = GaussianDiffusion(model, num_atoms=N, norm_factor=3.0)
x0 = torch.randn(a.P, N, 3); x0 = (x0 - x0.mean(1, keepdim=True)) * 3.0
ld = LangevinDiffusion(diff, x0, a.steps, save_interval=a.steps, t=20, temp_data=340, temp_sim=340, dt=None,
                       masses=[12.0] * N, friction=1.0, verbose=False)
ld.simulate(); torch.cuda.synchronize()
for wv in [int(w) for w in a.waves.split(",")]:
    model.native.profile(1 + wv)
    ld2 = LangevinDiffusion(diff, x0, a.steps, save_interval=a.steps, t=20, temp_data=340, temp_sim=340, dt=None,
                            masses=[12.0] * N, friction=1.0, verbose=False)
    t0 = time.perf_counter(); ld2.simulate(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = model.native.profile_read()
    if "small" in model.native.last_launch()[0]:
        names = ["centre", "rowA ln1(l0)", "attn fwd (wave-private)", "rowB gate1+ln2", "ffn fwd", "rowC gate2+ln1", "rowD b_gate2",
                 "ffn bwd", "rowE b_ln2+gate1", "attn bwd (wave-private)", "rowF b_ln1", "update",
                 "  f:qkv gemm", "  f:S+softmax+PV", "  f:wox gemm", "  b:gext gemm", "  b:dA+dS", "  b:dV,dQ,dK", "  b:qkvT gemm",
                 "  ffn: A load+split", "  ffn: W1+gelu", "  ffn: W2", "  rowE: psum", "  rowE: gate"]
        vals = list(pr.values())
        names += [f"  (extra tick {i})" for i in range(len(names), len(vals))]
        pr = {n: vals[i] for i, n in enumerate(names) if i < 24 or vals[i]}
    tot = sum(pr.values())
    print(f"wave {wv}: {a.cfg} P={a.P} steps={a.steps} kernel={model.native.last_launch()} wall={1e6*dt/a.steps:.1f} us/step  cycles/step={tot/a.steps:.0f}")
    for k, v in pr.items():
        print(f"  {k:20s} {v/a.steps:10.0f} cyc/step  {100.0*v/tot:5.1f}%")
