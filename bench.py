#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: Langevin MD-steps/s at batch 256.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "config 2"): chignolin (10 beads, H=64,
L=3), Langevin, parallel_sim=256 per GPU, noise_level t=20, T_data=T_sim=340 K, m=12, friction
1, auto dt, save_interval=250; synthetic weights (oracle/synth.py seed 1234, node_decoder x1e-2)
because the checkpoints are not in the reference mount; x0 ~ N(0,1) centred; in-kernel Philox
noise.  One "step" = all 256 trajectories of a rank advanced once: score-network forward + VJP
+ BAOAB update.  The timed region is K steps issued as persistent-kernel launches of
`--chunk` steps (default 250 = save_interval: one saved frame per launch), state resident in HBM.

  python bench.py [--gpus N --steps K --warmup W]           (N>1: under torch.distributed.run)

Prints ONE JSON line (rank 0).  `value` is the whole-job aggregate in batch-256 MD-steps/s
(weak scaling: every rank advances its own 256 trajectories; no data-path collective -- the
only collective is the final all_gather of frames, timed separately as `gather_ms`).
`roofline`: fp32 MFMA/VALU peak 157.3 TFLOP/s (MI355X_MICROARCH.md) against the ALGORITHMIC
FLOPs of SURVEY.md section 8d (22.00 MFLOP per chignolin score call, factorised formulation),
per-launch durations from HIP events on the launch stream.  `cpu_baseline`: the oracle twin of
the reference (oracle/reference_twin.py, materialised formulation, torch CPU, all host cores)
on a bounded number of the same steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per score call per protein (fwd + VJP), factorised
MFLOP_PER_CALL = {"ala2": 11.27, "chignolin": 22.00, "trp_cage": 102.97, "bba": 107.19,
                  "villin": 190.40, "protein_g": 327.47}
NORM_STD = {"chignolin": 3.113133430480957, "villin": 6.082900047302246, "protein_g": 6.354289531707764,
            "ala2": 0.9449278712272644, "trp_cage": 5.08211088180542, "bba": 6.294918537139893}
TEMP = {"chignolin": 340, "villin": 360, "protein_g": 350, "ala2": 300, "trp_cage": 290, "bba": 325}
PEAK_FP32_TFLOPS = 157.3


def hbm_traffic_from_profile(kname, cfg, P, chunk):
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    gfx950 x2 read correction) for this exact kernel + workload: profiles/<round>/traffic.json."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "**", "traffic.json"), recursive=True)):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        # rocprofv3 prints the full template argument list ("<64, 8, false>"), the library the short name ("<64,8>")
        def norm(n):
            n = n.replace(" ", "")
            return n[:-len(",false>")] + ">" if n.endswith(",false>") and n.count(",") in (2, 4) else n
        if norm(t.get("kernel", "")) == norm(kname) and t.get("workload") == f"{cfg} P={P} chunk={chunk}":
            best = t
    return None if best is None else float(best["hbm_bytes_per_launch"])


def cpu_baseline(cfg, P, t_level, budget_s=12.0, max_steps=40):
    """Oracle twin timed on the host cores (rank 0, N=1 only).  torch's default of one thread per
    logical core is far from optimal for these small ops, so a 1-step probe picks the best of a
    few thread counts first; the count used is what `cores` reports."""
    from oracle import reference_twin as twin
    from oracle import synth
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    p = twin.to_torch(synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
    masses = [12.8 if cfg == "ala2" else 12.0] * N
    c = twin.langevin_constants(NORM_STD[cfg], t_level, twin.make_schedule(), TEMP[cfg], TEMP[cfg], masses, 1.0, None)
    g = torch.Generator().manual_seed(2024)
    x = twin.center_zero(torch.randn(P, N, 3, generator=g))
    v = torch.zeros_like(x)
    m = torch.tensor(masses)

    def step(x, v):
        x = twin.center_zero(x)
        f = twin.forces(p, x, c, L)
        return twin.langevin_step(x, v, f, torch.randn(P, N, 3, generator=g), m, c)

    ncpu = os.cpu_count() or 1
    best_thr, best_t = torch.get_num_threads(), None
    x, v = step(x, v)  # warm
    for thr in sorted({t for t in (8, 16, 32, 64, ncpu // 2) if 1 <= t <= ncpu}):
        torch.set_num_threads(thr)
        x, v = step(x, v)
        t1 = time.perf_counter()
        x, v = step(x, v)
        dt1 = time.perf_counter() - t1
        if best_t is None or dt1 < best_t:
            best_thr, best_t = thr, dt1
    torch.set_num_threads(best_thr)
    for _ in range(2):
        x, v = step(x, v)
    n, t0 = 0, time.perf_counter()
    while n < max_steps and (time.perf_counter() - t0) < budget_s:
        x, v = step(x, v)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "MD-steps/s (batch 256)", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": 1e3 * dt / n,
            "sample": f"{n} Langevin steps of the same workload (P={P}, {cfg}) after warm-up and a thread-count probe, "
                      f"oracle/reference_twin.py on {torch.get_num_threads()} threads of {os.cpu_count()} logical cores"}


def north_star_extras(dev, rank, world, P):
    """The other figures BASELINE.json's north_star names, measured AFTER the timed region and reported
    under `also` (never part of `value`): i.i.d. samples/s (full 1000-step DDPM reverse chains, the layer-0
    table build included) and Langevin MD-steps/s on chignolin and villin at batch P per GPU."""
    import torch.distributed as dist
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    from dff_amd.score import GraphTransformer
    from oracle import synth
    out = {}
    for cfg, split in (("chignolin", False), ("villin", False), ("villin", True)):
        _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
        # split: the opt-in variants whose K = H weight GEMMs run on the bf16 pipe through an exact three-way split of
        # every fp32 operand; read when the model is created, off for every other number here
        os.environ["DFF_SPLIT_BF16"] = "1" if split else "0"
        model = GraphTransformer(N, H, device=dev, n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True,
                                 state_dict=synth.synth_gnn_params(N, H, L, seed=1234, decoder_scale=1e-2))
        diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=NORM_STD[cfg])
        diff.seed(77 + rank)
        x0 = torch.randn(P, N, 3, generator=torch.Generator().manual_seed(2024 + rank))
        x0 = (x0 - x0.mean(1, keepdim=True)) * NORM_STD[cfg]

        def iid():
            return diff.sample(P)

        def md():
            LangevinDiffusion(diff, x0, 1000, save_interval=250, t=20 if cfg == "chignolin" else 5, temp_data=TEMP[cfg],
                              temp_sim=TEMP[cfg], dt=None, masses=[12.0] * N, friction=1.0, seed=1234,
                              verbose=False).simulate(traj_offset=rank * P)

        for name, fn, units in (("iid_samples_per_s", iid, P), ("md_steps_per_s", md, 1000)):
            fn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            out[f"{cfg}_{name}" + ("_split_bf16_opt_in" if split else "")] = world * units / dt.item()
    os.environ["DFF_SPLIT_BF16"] = "0"
    out["note"] = (f"whole job, batch {P} per GPU; iid = complete 1000-step DDPM chains; md = 1000 Langevin steps, save_interval 250 "
                   f"(chignolin t=20, villin t=5), host set-up of each call included; *_split_bf16_opt_in: DFF_SPLIT_BF16=1 variants "
                   f"(weight GEMMs on the bf16 MFMA via an exact 3-way split of every fp32 operand, same parity tolerances), "
                   f"off for every other number in this line")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--chunk", type=int, default=250, help="steps per persistent-kernel launch (= save_interval)")
    ap.add_argument("--cfg", default="chignolin")
    ap.add_argument("--parallel_sim", type=int, default=256, help="trajectories per GPU")
    ap.add_argument("--noise_level", type=int, default=20)
    ap.add_argument("--group", type=int, default=0, help="proteins per workgroup (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary north-star figures (`also`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    import dff_amd
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.langevin import LangevinDiffusion
    from dff_amd.score import GraphTransformer
    from oracle import synth  # synthetic weights only (shared with the parity tests)

    cfg = args.cfg
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    P = args.parallel_sim
    params = synth.synth_gnn_params(N, H, L, seed=1234, decoder_scale=1e-2)
    model = GraphTransformer(N, H, device=dev, n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                             use_distances=False, conservative=True, state_dict=params)
    if args.group:
        model.native.set_group(args.group)
    diff = GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=NORM_STD[cfg])
    g = torch.Generator().manual_seed(2024 + rank)
    x0 = torch.randn(P, N, 3, generator=g)
    x0 = (x0 - x0.mean(1, keepdim=True)) * NORM_STD[cfg]
    chunk = args.chunk
    K = (args.steps // chunk) * chunk or chunk
    W = ((args.warmup + chunk - 1) // chunk) * chunk if args.warmup > 0 else 0
    masses = [12.8 if cfg == "ala2" else 12.0] * N
    ld = LangevinDiffusion(diff, x0, K + W, save_interval=chunk, t=args.noise_level, temp_data=TEMP[cfg],
                           temp_sim=TEMP[cfg], dt=None, masses=masses, friction=1.0, seed=1234, verbose=False)
    n_frames = (K + W) // chunk
    frames = torch.empty(n_frames, P, N, 3, device=dev)
    ke = torch.empty(n_frames, P, device=dev)

    def run_chunk(i):
        model.native.langevin_run(ld.params, ld.x, ld.v, chunk, chunk, noise=None, seed=1234,
                                  traj_offset=rank * P, step_offset=i * chunk,
                                  frames=frames[i:i + 1], ke=ke[i:i + 1])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W // chunk):
        run_chunk(i)
    barrier()
    nl = K // chunk
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nl)]
    t0 = time.perf_counter()
    for j in range(nl):
        ev[j][0].record()
        run_chunk(W // chunk + j)
        ev[j][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = [a.elapsed_time(b) for a, b in ev]
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    # the job's only collective: gather the saved frames (xGMI); not part of a "step"
    gather_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = [torch.empty_like(frames) for _ in range(world)]
        dist.all_gather(out, frames)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t1)
    ok = bool(torch.isfinite(frames).all().item()) and bool(torch.isfinite(ld.x).all().item())
    also = None
    if not args.no_extras:
        try:    # secondary figures must never cost the headline line
            also = north_star_extras(dev, rank, world, P)
        except Exception as e:  # noqa: BLE001
            also = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / K
        traj_steps = world * P * K / elapsed
        value = traj_steps / 256.0
        kname, grid, lds = model.native.last_launch()
        avg_launch_ms = float(np.mean(launch_ms))
        flops_per_launch = MFLOP_PER_CALL[cfg] * 1e6 * P * chunk
        achieved = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
        res = {
            "metric": "Langevin MD-steps/sec at batch 256 (chignolin, score fwd+VJP + BAOAB per step)",
            "value": value, "unit": "MD-steps/s (batch-256 steps, whole job)", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded weights, N(0,1) centred x0, in-kernel Philox noise)",
            "config": {"workload": f"BASELINE configs[1]: {cfg} ({N} beads, H={H}, L={L}) Langevin, parallel_sim={P}/GPU, "
                                   f"noise_level={args.noise_level}, save_interval={chunk}, 1 persistent launch per {chunk} steps",
                       "parallelism": f"{world} x independent trajectory shards (no data-path collective)",
                       "kernel": kname, "grid": grid, "lds_bytes": lds},
            "trajectory_steps_per_s": traj_steps, "finite": ok, "gather_ms": gather_ms,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": hbm_traffic_from_profile(kname, cfg, P, chunk),
                         "kernel": kname, "avg_launch_ms": avg_launch_ms, "launches": nl,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                         "note": "fp32-compute bound: algorithmic HBM bytes are only 600 B per trajectory-step "
                                 "(x, v in/out + noise); measured traffic is dominated by the L2-spilling "
                                 "activation stash and is ~18% of HBM peak, not the binding roof"},
        }
        if also is not None:
            res["also"] = also
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(cfg, P, args.noise_level)
            res["speedup_vs_cpu_port"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
