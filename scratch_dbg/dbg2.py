import sys, os, functools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
print = functools.partial(print, flush=True)
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from oracle import synth, reference_twin as twin
which = sys.argv[1]
N, H, L = {"a": (5, 96, 2), "b": (10, 96, 2), "c": (5, 64, 2), "d": (5, 96, 3), "e": (5, 128, 2), "f": (10, 64, 2)}[which]
params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=params)
p = twin.to_torch(params)
diff = GaussianDiffusion(model, num_atoms=N, norm_factor=1.0)
x0 = twin.center_zero(torch.from_numpy(synth.normal((3, N, 3), 3, 3).astype(np.float32)))
sched = twin.make_schedule()
for K in (2, 3):
    noises = torch.from_numpy(synth.normal((K, 3, N, 3), 4, 4).astype(np.float32))
    print("launch", which, N, H, L, K)
    y = diff.p_sample_loop_from(x0, 500, 500-K+1, noises=noises).cpu().numpy()
    ref = x0
    for k in range(K):
        ref = twin.center_zero(torch.clamp(twin.p_sample(p, sched, ref, 500-k, noises[k], L), -1000, 1000))
    print(which, "ddpm K", K, model.native.last_launch()[0], "maxerr", np.abs(y-ref.numpy()).max()/np.abs(ref.numpy()).max())
