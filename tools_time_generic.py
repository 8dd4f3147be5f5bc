#!/usr/bin/env python3
"""Development timing: Langevin us / step and i.i.d. chains of one architecture, optionally on the forced-generic path.
usage: tools_time_generic.py <cfg> <force_generic 0|1> [P]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd, synth_weights as synth  # noqa: E401,F401
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
cfg = sys.argv[1]; force = int(sys.argv[2]); P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
if cfg.startswith("custom:"):   # custom:N:H:L
    N, H, L = (int(v) for v in cfg.split(":")[1:])
else:
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False,
                         conservative=True, state_dict=synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
if force:
    model.native.force_generic(True)
diff = GaussianDiffusion(model, num_atoms=N, norm_factor=3.0, defer_checks=True)
x0 = torch.randn(P, N, 3); x0 = (x0 - x0.mean(1, keepdim=True)) * 3.0
for rep in range(3):
    ld = LangevinDiffusion(diff, x0, 1000, save_interval=250, t=20, temp_data=300, temp_sim=300, dt=None, masses=[12.0] * N, friction=1.0, verbose=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ld.simulate(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
kn = model.native.last_launch()
diff.seed(5)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); diff.sample(P); torch.cuda.synchronize(); dti = time.perf_counter() - t0
print(cfg, "P", P, "force_generic", force, kn, f"langevin {1e6 * dt / 1000:.1f} us/step", model.native.last_launch()[0], f"iid {1e6 * dti / 1000:.1f} us/reverse step")
