import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import functools
print = functools.partial(print, flush=True)
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from dff_amd.langevin import LangevinDiffusion
from oracle import synth, reference_twin as twin
for cfg in ("ala2", "chignolin"):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
    model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=params)
    p = twin.to_torch(params)
    norm = 1.0
    diff = GaussianDiffusion(model, num_atoms=N, norm_factor=norm)
    x0 = twin.center_zero(torch.from_numpy(synth.normal((3, N, 3), 3, 3).astype(np.float32)))
    for K in (1, 2, 3, 5):
        noises = torch.from_numpy(synth.normal((K, 3, N, 3), 4, 4).astype(np.float32))
        for generic in (False, True):
            model.native.force_generic(generic)
            ld = LangevinDiffusion(diff, x0, K, save_interval=1, t=20, temp_data=300, temp_sim=300, dt=None, masses=[12.0]*N, friction=1.0, verbose=False)
            print("  launching langevin", cfg, K, generic, flush=True)
            tr = ld.sample(noises=noises).numpy()
            c = twin.langevin_constants(norm, 20, twin.make_schedule(), 300, 300, [12.0]*N, 1.0, None)
            fr, _, _, _ = twin.simulate(p, x0, noises, [12.0]*N, c, L, 1)
            ref = fr.reshape(-1, N, 3).numpy()
            torch.cuda.synchronize(); print(cfg, "langevin K", K, "generic" if generic else "small", model.native.last_launch()[0], "maxerr", np.abs(tr-ref).max()/np.abs(ref).max())
        # ddpm
        sched = twin.make_schedule()
        for generic in (False, True):
            model.native.force_generic(generic)
            print("  launching ddpm", cfg, K, generic, flush=True)
            y = diff.p_sample_loop_from(x0, 500, 500-K+1, noises=noises).cpu().numpy()
            ref = x0
            for k in range(K):
                ref = twin.center_zero(torch.clamp(twin.p_sample(p, sched, ref, 500-k, noises[k], L), -1000, 1000))
            print(cfg, "ddpm K", K, "generic" if generic else "small", "maxerr", np.abs(y-ref.numpy()).max()/np.abs(ref.numpy()).max())
    model.native.force_generic(False)
