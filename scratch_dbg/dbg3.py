import sys, os, functools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
print = functools.partial(print, flush=True)
import dff_amd
from dff_amd.score import GraphTransformer
from dff_amd.ddpm import GaussianDiffusion
from oracle import synth, reference_twin as twin
N, H, L = 5, 96, 2
mode = sys.argv[1]
params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
model = GraphTransformer(N, H, device="cuda:0", n_layers=L, use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, conservative=True, state_dict=params)
diff = GaussianDiffusion(model, num_atoms=N, norm_factor=1.0)
x0 = twin.center_zero(torch.from_numpy(synth.normal((3, N, 3), 3, 3).astype(np.float32)))
noises = torch.from_numpy(synth.normal((2, 3, N, 3), 4, 4).astype(np.float32))
if mode == "sep":
    y = diff.p_sample_loop_from(x0, 500, 500, noises=noises[:1])
    torch.cuda.synchronize(); print("first ok")
    y = diff.p_sample_loop_from(y, 499, 499, noises=noises[1:])
    torch.cuda.synchronize(); print("second ok")
elif mode == "philox":
    y = diff.p_sample_loop_from(x0, 500, 499, noises=None)
    torch.cuda.synchronize(); print("philox ok")
elif mode == "B1":
    y = diff.p_sample_loop_from(x0[:1], 500, 499, noises=noises[:, :1].contiguous())
    torch.cuda.synchronize(); print("B1 ok")
elif mode == "score2":
    t = torch.full((3,), 0.5, device="cuda")
    for _ in range(3):
        f = model.native.score(x0.cuda(), t)
    torch.cuda.synchronize(); print("score ok")
