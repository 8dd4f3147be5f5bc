"""``sample.py``-compatible command line (reference: sample.py:18-249).

Same flags, same inputs (``{model_path}/args.pickle``, ``{model_path}/model-{ckpt}.pt["ema"]``) and
same outputs (``{model_path}/main_eval_output_{gen_mode}[_{append}]/sample-{gen_mode}.pt`` -- a
float32 CPU tensor (n, N, 3) in Angstrom -- plus a ``.pdb`` of the first 1000 frames).  What
differs is underneath: the score network, the DDPM reverse loop and the Langevin integrator run in
the fused HIP kernel (libdff_amd.so), ``mdtraj`` / ``ema_pytorch`` / ``tensorboard`` / the dataset
classes are not needed, ``--masses`` is parsed by a small safe expression reader instead of ``eval``, and
multi-GPU is one process per GPU (``torchrun --nproc-per-node G sample.py ...``): samples /
trajectories are sharded across ranks, each rank draws from its own Philox sub-stream, and the
only collective is one all_gather of the finished samples (RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import ast
import os
import pickle
import time
from os.path import join
from pathlib import Path

import torch

from . import specs, weights
from .sampling import SamplerWrapper, dist_backend, dist_env, gather_variable, sample_from_model, shard_range


def parse_masses(text: str):
    """``--masses`` is ``type=eval`` in the reference (sample.py:62); accept the same spellings a user
    would type there -- a list literal or ``[m] * n`` / concatenations of those -- without ``eval``."""
    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return node.value
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, (ast.List, ast.Tuple)):
            return [ev(e) for e in node.elts]
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult):
            a, b = ev(node.left), ev(node.right)
            if isinstance(a, list) and isinstance(b, int):
                return a * b
            if isinstance(b, list) and isinstance(a, int):
                return b * a
            if not isinstance(a, list) and not isinstance(b, list):
                return a * b
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            a, b = ev(node.left), ev(node.right)
            if isinstance(a, list) == isinstance(b, list):
                return a + b
        raise argparse.ArgumentTypeError(f"unsupported expression in --masses: {text!r}")
    try:
        out = ev(ast.parse(text, mode="eval"))
    except SyntaxError as e:
        raise argparse.ArgumentTypeError(f"cannot parse --masses {text!r}: {e}")
    if not isinstance(out, list) or not all(isinstance(v, (int, float)) for v in out):
        raise argparse.ArgumentTypeError("--masses must be a list of numbers")
    return [float(v) for v in out]


def build_parser() -> argparse.ArgumentParser:
    """The reference's flags verbatim (sample.py:18-95) + --seed / --chunk."""
    p = argparse.ArgumentParser(description="coarse-graining-evaluator")
    p.add_argument("--model_path", type=str, required=True, help="root directory where models and args are stored")
    p.add_argument("--model_checkpoint", type=str, default="best", help="best, last, 1, 2, 3, ...")
    p.add_argument("--gen_mode", type=str, default="iid", help="generative mode, either iid or langevin")
    p.add_argument("--append_exp_name", type=str, default=None,
                   help="append this text to the results/main_eval_output folder name, append only gen_mode if None (default)")
    p.add_argument("--data_folder", type=str, default=None,
                   help="accepted for compatibility; sampling never reads a dataset")
    # i.i.d. generation arguments
    p.add_argument("--num_samples_eval", type=int, default=1000, help="number of samples for i.i.d. generation")
    p.add_argument("--batch_size_gen", type=int, default=256, help="batch size for evaluation")
    # Langevin simulation arguments
    p.add_argument("--masses", type=parse_masses, default=None, help="Units in g/mol, e.g. \"[12.0]*10\"")
    p.add_argument("--friction", type=float, default=1, help="No units yet. Ideally units should be in ps^-1, usually 1")
    p.add_argument("--parallel_sim", type=int, default=100, help="Number of parallel simulations")
    p.add_argument("--n_timesteps", type=int, default=10000, help="number of timesteps")
    p.add_argument("--save_interval", type=int, default=250, help="save interval (in timesteps)")
    p.add_argument("--noise_level", type=int, default=20, help="diffusion model noise level for extracting force fields")
    p.add_argument("--dt", type=float, default=None,
                   help="Ideally 1~2fs (units in ps), if None it will be computed automatically according to the diffusion model parameters")
    p.add_argument("--temp_data", type=float, default=None, help="temperature in Kelvin.")
    p.add_argument("--temp_sim", type=float, default=None, help="temperature in Kelvin")
    p.add_argument("--kb", type=str, default="consistent", help="consistent, kcal")
    # additions
    p.add_argument("--seed", type=int, default=0, help="Philox seed for all in-kernel draws")
    p.add_argument("--chunk", type=int, default=None, help="Langevin steps per kernel launch (default: all)")
    return p


def load_training_args(model_path: str):
    """args.pickle is an argparse.Namespace that also pickles an nn.Tanh (needs torch to load)."""
    with open(join(model_path, "args.pickle"), "rb") as f:
        return pickle.load(f)


def build_diffusion(args, model_path: str, checkpoint: str, device, seed: int = 0):
    """sample.py:131-167 without datasets / EMA wrapper: metadata from specs, weights by key name."""
    from .ddpm import GaussianDiffusion
    from .score import GraphTransformer
    mol = specs.lookup(args.mol)
    if getattr(args, "backbone_network", "graph-transformer") != "graph-transformer":
        raise Exception(f"Network {args.backbone_network} not implemented")
    norm_factor = specs.norm_std(args.mol, getattr(args, "fold", 1)) if args.scale_data else 1.0
    ckpt = join(model_path, f"model-{checkpoint}.pt")
    data = torch.load(ckpt, map_location="cpu", weights_only=False)
    params = weights.gnn_params_from_checkpoint(data)
    model_nn = GraphTransformer(mol.n_beads, hidden_nf=args.hidden_features_gnn, device=device,
                                n_layers=args.num_layers_gnn, use_intrinsic_coords=args.use_intrinsic_coords,
                                use_abs_coords=args.use_abs_coords, use_distances=args.use_distances,
                                conservative=args.conservative, state_dict=params, timesteps=args.diffusion_steps)
    ddpm = GaussianDiffusion(model=model_nn, features=torch.eye(mol.n_beads), num_atoms=mol.n_beads,
                             timesteps=args.diffusion_steps, norm_factor=norm_factor,
                             loss_weights=getattr(args, "loss_weights", "ones"), seed=seed,
                             defer_checks=True)   # generate_samples checks once after all its batches (one host sync)
    return ddpm, mol


def generate_samples(ddpm, mol, samp_args, args, rank: int, world: int):
    """sample.py:176-249 (mode dispatch), sharded over `world` ranks."""
    from .langevin import LangevinDiffusion, temp_dict
    if samp_args.gen_mode == "iid":
        n_local = samp_args.num_samples_eval // world          # sample.py:185-189 convention
        lo = rank * n_local
        ddpm._samples_drawn = lo                               # Philox counter = global sample index
        sampled = sample_from_model(SamplerWrapper(ddpm), n_local, max(1, samp_args.batch_size_gen // world),
                                    verbose=(rank == 0))
        ddpm.check_clamp()
        total = n_local * world
    elif samp_args.gen_mode == "langevin":
        if rank == 0:
            print("Total number of samples to save using Langevin Dynamics: "
                  f"{int(samp_args.parallel_sim * samp_args.n_timesteps / samp_args.save_interval)}")
        p_local = samp_args.parallel_sim // world
        lo = rank * p_local
        ddpm._samples_drawn = lo
        init_mol = sample_from_model(SamplerWrapper(ddpm), p_local, max(1, samp_args.batch_size_gen // world),
                                     verbose=(rank == 0), to_cpu=False)      # stays on the device for the integrator
        ddpm.check_clamp()
        masses = samp_args.masses
        if masses is None:
            masses = specs.default_masses(args.mol)            # sample.py:216-221
        sampler = LangevinDiffusion(ddpm, init_mol, samp_args.n_timesteps, save_interval=samp_args.save_interval,
                                    t=samp_args.noise_level, diffusion_steps=args.diffusion_steps,
                                    temp_data=samp_args.temp_data, temp_sim=samp_args.temp_sim, dt=samp_args.dt,
                                    masses=masses, friction=samp_args.friction, kb=samp_args.kb,
                                    seed=samp_args.seed + 1, chunk=samp_args.chunk, verbose=(rank == 0))
        sampled = sampler.sample(traj_offset=lo)               # simulation-major, so rank order == sim order
        total = p_local * world * (samp_args.n_timesteps // samp_args.save_interval)
    else:
        raise Exception("Wrong argument 'gen_mode'")
    if world > 1:
        import torch.distributed as dist
        dev = ddpm.device if dist.get_backend() == "nccl" else torch.device("cpu")
        sampled = gather_variable(sampled.to(dev), total, world).cpu()
    return sampled


def main(argv=None):
    from .langevin import temp_dict
    from .pdbio import save_pdb
    samp_args = build_parser().parse_args(argv)
    rank, local_rank, world = dist_env()
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this sampler has no CPU path (the HIP library is the product)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = dist_backend()    # "nccl" = RCCL over xGMI; a test knob selects "gloo" when ranks share one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    args = load_training_args(samp_args.model_path)
    if samp_args.temp_data is None:
        samp_args.temp_data = temp_dict[args.mol.upper()]
    if samp_args.temp_sim is None:
        samp_args.temp_sim = temp_dict[args.mol.upper()]
    basic_append = f"_{samp_args.gen_mode}"
    append = basic_append if samp_args.append_exp_name is None else f"{basic_append}_{samp_args.append_exp_name}"
    eval_folder = Path(join(samp_args.model_path, "main_eval_output" + append))
    if rank == 0:
        eval_folder.mkdir(exist_ok=True, parents=False)
    ddpm, mol = build_diffusion(args, samp_args.model_path, samp_args.model_checkpoint, device, seed=samp_args.seed)
    t0 = time.time()
    sampled_mol = generate_samples(ddpm, mol, samp_args, args, rank, world)
    if rank == 0:
        print(f"{len(sampled_mol)} frames in {time.time() - t0:.1f} s")
        torch.save(sampled_mol, str(eval_folder / f"sample-{samp_args.gen_mode}.pt"))
        save_pdb(str(eval_folder / f"sample-{samp_args.gen_mode}.pdb"), sampled_mol[0:1000].numpy(), args.mol)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return sampled_mol


if __name__ == "__main__":
    main()
