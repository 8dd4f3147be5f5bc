import sys, os, functools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
print = functools.partial(print, flush=True)
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from oracle import synth, reference_twin as twin
N, H, L = 5, 96, 2
mode, B, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=params)
diff = GaussianDiffusion(model, num_atoms=N, norm_factor=1.0)
x0 = twin.center_zero(torch.from_numpy(synth.normal((B, N, 3), 3, 3).astype(np.float32)))
noises = torch.from_numpy(synth.normal((K, B, N, 3), 4, 4).astype(np.float32)).cuda()
for rep in range(3):
    y = diff.p_sample_loop_from(x0, 500, 500 - K + 1, noises=noises if mode == "noise" else None)
    torch.cuda.synchronize()
print(mode, B, K, "ok", float(y.abs().max()))
