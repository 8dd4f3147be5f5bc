#!/usr/bin/env python3
"""End-to-end wall time of the sample.py-compatible CLI on BASELINE config 2 (chignolin Langevin, 256
parallel simulations, 10000 steps, save_interval 250) and config 1-like iid, with a synthetic
checkpoint in the reference's layout.  Prints the wall time next to the pure kernel time."""
import json, os, pickle, sys, tempfile, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd
from dff_amd import cli
import synth_weights as synth

def write_model_dir(d, cfg):
    mol, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1e-2)
    args = types.SimpleNamespace(mol=mol, hidden_features_gnn=H, num_layers_gnn=L, diffusion_steps=1000, conservative=True,
                                 use_intrinsic_coords=True, use_abs_coords=False, use_distances=False, num_beads=N,
                                 norm_factor=3.113133430480957, loss_weights="higheruntil_100", experiment_name="synthetic",
                                 atom_selection="c-alpha", mean0=True, data_folder="", fold=None, pick_checkpoint="best",
                                 traindata_subset=None, scale_data=True, start_from_last_saved=False, parallel=False)
    pickle.dump(args, open(os.path.join(d, "args.pickle"), "wb"))
    sd = {"ema_model.model." + k: torch.from_numpy(v) for k, v in params.items()}
    torch.save({"ema": sd, "step": 1}, os.path.join(d, "model-best.pt"))

with tempfile.TemporaryDirectory() as d:
    write_model_dir(d, "chignolin")
    for name, argv in [("langevin config 2", ["--gen_mode", "langevin", "--parallel_sim", "256", "--n_timesteps", "10000",
                                               "--save_interval", "250", "--batch_size_gen", "256", "--masses", "[12.0]*10"]),
                       ("iid 1024 samples, batch 256", ["--gen_mode", "iid", "--num_samples_eval", "1024", "--batch_size_gen", "256"])]:
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = cli.main(["--model_path", d, "--append_exp_name", f"r{rep}"] + argv)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({"run": name, "wall_s_second_run": dt, "output_shape": list(out.shape)}))
