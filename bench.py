#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: Langevin MD-steps/s at batch 256.

Headline workload (BASELINE.json configs[1], SURVEY.md section 8d "config 2"): chignolin (10 beads, H=64, L=3),
Langevin, parallel_sim=256 per GPU, noise_level t=20, T_data=T_sim=340 K, m=12, friction 1, auto dt,
save_interval=250; synthetic weights (synth_weights.py seed 1234, node_decoder x1e-2) because the checkpoints are not in
the reference mount; x0 ~ N(0,1) centred; in-kernel Philox noise.  One "step" = all 256 trajectories of a rank advanced
once: score-network forward + VJP + BAOAB update.  Steps are issued as persistent-kernel launches of `--chunk` steps
(default 250 = save_interval: one saved frame per launch), state resident in HBM.

  python bench.py [--gpus N --steps K --warmup W]           (N>1: under torch.distributed.run)

The timed region is never shorter than 8 launches: `--steps` / `--warmup` are rounded UP to whole launches and the
line reports the steps actually run (`steps`, `warmup`) next to the request (`steps_requested`, `warmup_requested`).

Prints ONE JSON line (rank 0).  `value` is the whole-job aggregate in batch-256 MD-steps/s (weak scaling: every rank
advances its own 256 trajectories; no data-path collective -- the only collective is the final all_gather of frames,
timed separately as `gather_ms`).  `roofline`: fp32 MFMA/VALU peak 157.3 TFLOP/s (MI355X_MICROARCH.md) against the
ALGORITHMIC FLOPs of SURVEY.md section 8d (22.00 MFLOP per chignolin score call, factorised formulation), per-launch
durations from HIP events on the launch stream.  `cpu_baseline`: the oracle twin of the reference
(oracle/reference_twin.py, materialised formulation, torch CPU) on a bounded number of the same steps.
`also`: the other figures BASELINE.json's north_star names -- villin (35 beads) Langevin at 256 per GPU, protein G (56
beads) at 128 per GPU, chignolin / villin i.i.d. samples/s -- each a first-class entry (<= 700 bytes) with its own `roofline`
object (and a `cpu_baseline`), measured after the headline's timed region.  What every entry would repeat (dtype, how frac /
traffic / mfma_busy / hbm_tbps are defined) is said once, in `notes`, the LAST key: the whole line stays under 6 KB so that
the driver's stdout tail holds all of it.
"""
import argparse
import glob
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
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PEAK_BF16_DENSE_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak; an fp32-exact product costs six bf16 products or three fp16 ones
MIN_LAUNCHES = 8
# DFF_FORCE_DIST=1: run the N > 1 code (process group over RCCL, barriers, max-over-ranks all_reduce, the frame all_gather)
# at ANY world size -- with one rank under torchrun this loads RCCL and executes every collective of the job on the one GPU
# a test box has (tests/test_gpu_parity.py::test_bench_rccl_world_one), before the first 8-GPU run does.
FORCE_DIST = os.environ.get("DFF_FORCE_DIST") == "1"


def library_src_sha():
    """Hash of the sources the loaded libdff_amd.so was built from (dff_version(); build.sh)."""
    try:
        from dff_amd import srcsha
        return srcsha.library_sha()
    except Exception:   # noqa: BLE001
        return "unknown"


CHECK_PROFILE_SHA = True   # (tests of the kernel-name matching against older rounds' profiles switch it off)


def profile_figures(kname, cfg, P, chunk, lib_sha=None):
    """rocprofv3 figures for this exact kernel + workload from profiles/<round>/**/traffic.json (latest round wins;
    written by tools_profile_report.py from separate --pmc passes): HBM bytes per launch (FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE), HBM TB/s, MFMA pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles)), L2 hit rate.
    Older traffic.json files carry the bytes only: the rest is derived from the summary.json next to them.
    A traffic.json is tied to the sources it profiled (`src_sha`, stamped by tools_profile_report.py): when that is not the
    hash of the running library the counters are NOT reported (all None) and `profile_stale` is True -- a kernel changed
    after its last profile must not carry the old counters next to fresh timings."""
    best = None
    lib_sha = lib_sha or (library_src_sha() if CHECK_PROFILE_SHA else None)

    def norm(n):   # rocprofv3 prints the full template argument list, the library its own short name
        n = n.replace(" ", "").replace("void", "").replace("split_f16", "split_bf16")   # (one template flag: the split engine, whichever pieces)
        if "<" not in n:
            return n
        base, args = n.split("<", 1)
        args = args.rstrip(">").split(",")
        if all(x in ("true", "false") or x.isdigit() for x in args):   # rocprofv3 form
            flags = (("gen", "split_bf16", "fold_kv") if base == "dff_small_kernel" else ("gen", "split_bf16", "pair"))
            nfix = 2 if base == "dff_small_kernel" else 4               # <H, NW, ...> / <H, MT, HGS, SPILL, ...>
            tail = [f for f, v in zip(flags, args[nfix:]) if v == "true"]
            args = args[:nfix] + tail
        return base + "<" + ",".join(args) + ">"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "**", "traffic.json"), recursive=True)):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if norm(t.get("kernel", "")) == norm(kname) and t.get("workload") == f"{cfg} P={P} chunk={chunk}":
            best = (t, os.path.dirname(f))
    if best is None:
        return {"traffic": None, "hbm_tbps": None, "mfma_busy": None, "l2_hit": None, "profile": None, "profile_stale": None}
    t, d = best
    if CHECK_PROFILE_SHA and (t.get("src_sha") != lib_sha or lib_sha == "unknown"):
        return {"traffic": None, "hbm_tbps": None, "mfma_busy": None, "l2_hit": None, "profile": os.path.relpath(d, ROOT),
                "profile_stale": True}
    out = {"traffic": float(t["hbm_bytes_per_launch"]),
           "hbm_tbps": t.get("hbm_tbps", float(t["hbm_bytes_per_launch"]) / (float(t["avg_launch_ms"]) * 1e-3) / 1e12),
           "mfma_busy": t.get("mfma_busy"), "l2_hit": t.get("l2_hit"), "profile": os.path.relpath(d, ROOT),
           "profile_stale": False}
    if out["mfma_busy"] is None or out["l2_hit"] is None:
        try:
            c = {k: v["mean_per_launch"] for k, v in json.load(open(os.path.join(d, "summary.json")))["counters"].items()}
            if out["mfma_busy"] is None and c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
            if out["l2_hit"] is None and "TCC_HIT_sum" in c:
                out["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        except Exception:
            pass
    for k in ("hbm_tbps", "mfma_busy", "l2_hit"):
        if out[k] is not None:
            out[k] = round(out[k], 4)
    return out


# Said ONCE per line (`notes`), not once per entry: the driver keeps only the tail of stdout.
NOTES = {
    "dtype": "derived from the kernel that ran.  State, activations, accumulators: f32 everywhere.  split_f16: weight GEMMs on a 2-way "
             "fp16 split of every f32 operand (2^-22 per operand, three v_mfma_f32_16x16x32_f16 per product, gradients row-scaled by powers "
             "of two); split_bf16: exact 3-way bf16 split (six products); held to the fp32 reference's own error.  "
             "Attention, other kernels, DFF_SPLIT_BF16=0: v_mfma_f32_16x16x4_f32 / f32 VALU",
    "roofline": "bound = fp32 compute (SURVEY 8d): frac = algorithmic TFLOP/s (factorised FLOP count x proteins x steps / HIP-event "
                "launch time) / 157.3.  mixed_peak_frac: vs the running engine's own roof (weight-GEMM FLOPs at fp16 / 3 = 833 or bf16 / 6 "
                "= 416.7 TFLOP/s, the rest at 157.3).  traffic (HBM bytes per launch), hbm_tbps, mfma_busy, l2_hit: rocprofv3 PMC passes of "
                "the same workload under `profile`; None + profile_stale when that profile's sources (src_sha) are not the running library's",
    "fold_kv": "hidden == head dim: k / v projections folded into q / out (exact); ~26 % fewer MFMAs issued than the FLOP count used",
    "timing": "persistent launches of `chunk` fused steps; --steps / --warmup rounded up to whole launches, >= 8 timed",
    "cpu": "oracle/reference_twin.py (torch CPU port of the reference, materialised formulation), thread count picked by a probe",
    "also": "roofline objects there omit bound / peak / unit (= the headline's)",
}


def kernel_dtype(kname):
    """The arithmetic the kernel that ran multiplies its weights in (its name says which engine): inputs, outputs, accumulators
    and everything outside the weight GEMMs are f32 in every variant."""
    if "split_f16" in kname:
        return "f32 (weight GEMMs: 2xf16 split, f32 acc)"
    if "split_bf16" in kname:
        return "f32 (weight GEMMs: 3xbf16 split, f32 acc)"
    return "f32"


SHAPES = {"ala2": (5, 96, 2), "chignolin": (10, 64, 3), "trp_cage": (20, 128, 3), "bba": (28, 96, 3), "villin": (35, 128, 3),
          "protein_g": (56, 128, 3)}   # (beads, hidden, layers): synth_weights.SHIPPED_CONFIGS


def weight_gemm_mflop(cfg):
    """The share of MFLOP_PER_CALL[cfg] that is weight GEMMs (q, k, v, out projections + FFN, forward and VJP): SURVEY.md 8(d)'s
    closed form, 2 L lin with lin = 8 N H I + 16 N H^2, I = 512 -- what the split engines run on the fp16 / bf16 pipe."""
    N, H, L = SHAPES[cfg]
    return 2.0 * L * (8.0 * N * H * 512 + 16.0 * N * H * H) / 1e6


def mixed_peak_frac(cfg, P, steps_per_launch, avg_launch_ms, kname):
    """Against the running engine's OWN roof: the time one launch would take with its weight-GEMM FLOPs at the split products'
    rate (dense fp16 / 3 = 833 TFLOP/s, bf16 / 6 = 416.7; the fp32 engine: 157.3) and every other FLOP (attention, gates,
    heads) at the fp32 rate, over the measured time.  `frac` (all FLOPs at 157.3) stays the headline figure."""
    g = weight_gemm_mflop(cfg) * 1e6
    rest = MFLOP_PER_CALL[cfg] * 1e6 - g
    rate = PEAK_BF16_DENSE_TFLOPS / 3.0 if "split_f16" in kname else PEAK_BF16_DENSE_TFLOPS / 6.0 if "split_bf16" in kname else PEAK_FP32_TFLOPS
    ideal_s = (g / (rate * 1e12) + rest / (PEAK_FP32_TFLOPS * 1e12)) * P * steps_per_launch
    return ideal_s / (avg_launch_ms * 1e-3)


def roofline(cfg, P, steps_per_launch, launch_ms, kname, brief=False):
    """bound = fp32 compute: achieved algorithmic TFLOP/s of one launch against 157.3."""
    avg = float(np.mean(launch_ms))
    flops = MFLOP_PER_CALL[cfg] * 1e6 * P * steps_per_launch
    ach = flops / (avg * 1e-3) / 1e12
    pf = profile_figures(kname, cfg, P, steps_per_launch)
    r = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
         "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": pf["traffic"], "hbm_tbps": pf["hbm_tbps"],
         "mfma_busy": pf["mfma_busy"], "l2_hit": pf["l2_hit"], "kernel": kname, "avg_launch_ms": round(avg, 4),
         "min_launch_ms": round(float(np.min(launch_ms)), 4), "launches": len(launch_ms)}
    r["profile"] = pf["profile"]
    if pf["profile_stale"]:
        r["profile_stale"] = True
    if brief:   # `also` entries: bound / peak / unit are the headline's; counters of a stale / missing profile are not spelled out
        for k in ("bound", "peak", "unit", "min_launch_ms", "launches"):
            del r[k]
        if pf["traffic"] is None:
            for k in ("traffic", "hbm_tbps", "mfma_busy", "l2_hit"):
                del r[k]
    else:
        r["algorithmic_flops_per_launch"] = flops
    r["mixed_peak_frac"] = round(mixed_peak_frac(cfg, P, steps_per_launch, avg, kname), 4)
    return r


def cpu_baseline(cfg, P, t_level, budget_s=12.0, max_steps=40, threads=None):
    """Oracle twin timed on the host cores (rank 0, N=1 only).  torch's default of one thread per
    logical core is far from optimal for these small ops, so a 1-step probe picks the best of a
    few thread counts first (or `threads` is taken as given); the count used is what `cores` reports."""
    from oracle import reference_twin as twin
    import synth_weights as synth
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
    if threads:
        torch.set_num_threads(threads)
    x, v = step(x, v)  # warm
    for thr in ([] if threads else sorted({t for t in (8, 16, 32, 64, ncpu // 2) if 1 <= t <= ncpu})):
        torch.set_num_threads(thr)
        x, v = step(x, v)
        t1 = time.perf_counter()
        x, v = step(x, v)
        dt1 = time.perf_counter() - t1
        if best_t is None or dt1 < best_t:
            best_thr, best_t = thr, dt1
    torch.set_num_threads(threads or best_thr)
    if not threads:
        x, v = step(x, v)
    n, t0 = 0, time.perf_counter()
    while n < max_steps and (n < 2 or (time.perf_counter() - t0) < budget_s):
        x, v = step(x, v)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": f"MD-steps/s (batch {P})", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": 1e3 * dt / n,
            "sample": f"{n} Langevin steps, same workload (P={P}, {cfg}), {torch.get_num_threads()} of {os.cpu_count()} logical cores"}


def cpu_baseline_iid(cfg, P, budget_s=8.0, max_steps=20, threads=8):
    """The i.i.d. metric's CPU leg: reverse DDPM steps (models/ddpm.py:221-232: score network + posterior update) of the
    oracle twin at batch P on the host cores.  A sample is a complete 1000-step chain, so samples/s = P / (1000 x seconds
    per reverse step); the steps timed are a bounded sample from the top of the chain (every step costs the same: one
    score call of the same shape)."""
    from oracle import reference_twin as twin
    import synth_weights as synth
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    p = twin.to_torch(synth.synth_gnn_params(N, H, L, decoder_scale=1e-2))
    sched = twin.make_schedule()
    g = torch.Generator().manual_seed(2025)
    x = twin.center_zero(torch.randn(P, N, 3, generator=g))
    torch.set_num_threads(threads)
    t = 999
    x = twin.center_zero(twin.p_sample(p, sched, x, t, torch.randn(P, N, 3, generator=g), L))   # warm
    n, t0 = 0, time.perf_counter()
    while n < max_steps and (n < 2 or (time.perf_counter() - t0) < budget_s):
        t -= 1
        x = twin.center_zero(torch.clamp(twin.p_sample(p, sched, x, t, torch.randn(P, N, 3, generator=g), L), -1000, 1000))
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": P / (1000.0 * dt), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "ms_per_reverse_step": 1e3 * dt,
            "sample": f"{n} reverse steps (t = 998 .. {t}), batch {P}, {cfg}; a sample = 1000 such steps"}


def make_model(cfg, dev, fp32_engine=False):
    """fp32_engine: DFF_SPLIT_BF16=0 while the model is created -- every weight GEMM on v_mfma_f32_16x16x4_f32."""
    import synth_weights as synth
    from dff_amd.ddpm import GaussianDiffusion
    from dff_amd.score import GraphTransformer
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    old = os.environ.get("DFF_SPLIT_BF16")
    if fp32_engine:
        os.environ["DFF_SPLIT_BF16"] = "0"
    try:
        model = GraphTransformer(N, H, device=dev, n_layers=L, use_intrinsic_coords=True, use_abs_coords=False,
                                 use_distances=False, conservative=True,
                                 state_dict=synth.synth_gnn_params(N, H, L, seed=1234, decoder_scale=1e-2))
    finally:
        if fp32_engine:
            if old is None:
                os.environ.pop("DFF_SPLIT_BF16", None)
            else:
                os.environ["DFF_SPLIT_BF16"] = old
    return model, GaussianDiffusion(model, num_atoms=N, timesteps=1000, norm_factor=NORM_STD[cfg], defer_checks=True), (N, H, L)


class Timer:
    """barrier + synchronize on both sides of the timed launches, HIP events around every launch, max over ranks."""

    def __init__(self, dev, world):
        self.dev, self.world = dev, world
        self.dist = world > 1 or FORCE_DIST

    def barrier(self):
        if self.dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, fn, n_warm, n_timed):
        for i in range(n_warm):
            fn(i)
        self.barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_timed)]
        t0 = time.perf_counter()
        for j in range(n_timed):
            ev[j][0].record()
            fn(n_warm + j)
            ev[j][1].record()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.dist:
            import torch.distributed as dist
            tmax = torch.tensor([elapsed], device=self.dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        return elapsed, [a.elapsed_time(b) for a, b in ev]


def langevin_entry(cfg, P, chunk, n_warm, n_timed, dev, rank, world, noise_level=20, group=0, fp32_engine=False):
    """Langevin MD-steps/s of `cfg` at P trajectories per GPU: n_timed launches of `chunk` fused steps each."""
    from dff_amd.langevin import LangevinDiffusion
    model, diff, (N, H, L) = make_model(cfg, dev, fp32_engine)
    if group:
        model.native.set_group(group)
    x0 = torch.randn(P, N, 3, generator=torch.Generator().manual_seed(2024 + rank))
    x0 = (x0 - x0.mean(1, keepdim=True)) * NORM_STD[cfg]
    masses = [12.8 if cfg == "ala2" else 12.0] * N
    total = (n_warm + n_timed) * chunk
    ld = LangevinDiffusion(diff, x0, total, save_interval=chunk, t=noise_level, temp_data=TEMP[cfg],
                           temp_sim=TEMP[cfg], dt=None, masses=masses, friction=1.0, seed=1234, verbose=False)
    frames = torch.empty(n_warm + n_timed, P, N, 3, device=dev)
    ke = torch.empty(n_warm + n_timed, P, device=dev)

    def run_chunk(i):
        model.native.langevin_run(ld.params, ld.x, ld.v, chunk, chunk, noise=None, seed=1234,
                                  traj_offset=rank * P, step_offset=i * chunk, frames=frames[i:i + 1], ke=ke[i:i + 1])

    elapsed, launch_ms = Timer(dev, world).run(run_chunk, n_warm, n_timed)
    K = n_timed * chunk
    kname, grid, lds = model.native.last_launch()
    ok = bool(torch.isfinite(frames).all().item()) and bool(torch.isfinite(ld.x).all().item())
    return {"cfg": cfg, "N": N, "H": H, "L": L, "P": P, "K": K, "elapsed": elapsed, "launch_ms": launch_ms, "kernel": kname,
            "grid": grid, "lds": lds, "finite": ok, "frames": frames, "chunk": chunk}


def iid_entry(cfg, P, n_warm, n_timed, dev, rank, world):
    """i.i.d. samples/s: complete 1000-step reverse DDPM chains (one fused launch each; the layer-0 table for the 1000
    noise levels is built by the warm-up call)."""
    model, diff, (N, H, L) = make_model(cfg, dev)
    diff.seed(77 + rank)
    elapsed, launch_ms = Timer(dev, world).run(lambda i: diff.sample(P), n_warm, n_timed)
    diff.check_clamp()
    kname, grid, lds = model.native.last_launch()
    return {"cfg": cfg, "N": N, "H": H, "L": L, "P": P, "elapsed": elapsed, "launch_ms": launch_ms, "kernel": kname,
            "grid": grid, "lds": lds, "chains": n_timed}


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
    ap.add_argument("--mode", default="langevin", choices=["langevin", "iid"],
                    help="iid: time complete 1000-step reverse DDPM chains of --cfg at --parallel_sim per GPU instead (profiling runs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-extras", action="store_true", help="skip the other north-star figures (`also`)")
    args = ap.parse_args()

    import dff_amd  # noqa: F401  (loads libdff_amd.so: loud failure if the HIP extension is missing)
    from dff_amd.sampling import dist_backend, dist_env

    # RANK / LOCAL_RANK / WORLD_SIZE from torchrun; the test knobs of dist_env() (several ranks on ONE GPU over gloo, under
    # DFF_TEST_KNOBS=1) let the one-GPU box execute this N > 1 path: tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or FORCE_DIST
    backend = None
    if use_dist:
        import torch.distributed as dist
        backend = dist_backend()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    cfg, P, chunk = args.cfg, args.parallel_sim, args.chunk
    if args.mode == "iid":   # the i.i.d. workload on its own (what tools_rocprof.sh profiles for `also.*_iid.roofline.traffic`)
        nt = max(4, -(-args.steps // 1000))
        e = iid_entry(cfg, P, 1, nt, dev, rank, world)
        if rank == 0:
            print(json.dumps({"metric": f"i.i.d. samples/sec at batch {P} per GPU ({cfg}: complete 1000-step reverse DDPM chains)",
                              "value": world * P * nt / e["elapsed"], "unit": "samples/s (whole job)", "n_gpus": world,
                              "steps": nt * 1000, "warmup": 1000, "ms_per_step": 1e3 * e["elapsed"] / (nt * 1000),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": kernel_dtype(e["kernel"]),
                              "data": "synthetic (seeded weights, in-kernel Philox noise)",
                              "config": {"workload": f"{cfg} iid, batch {P}/GPU, 1000 reverse steps per launch", "kernel": e["kernel"]},
                              "roofline": roofline(cfg, P, 1000, e["launch_ms"], e["kernel"])}))
        if use_dist:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    n_timed = max(MIN_LAUNCHES, -(-args.steps // chunk))
    n_warm = max(1, -(-args.warmup // chunk)) if args.warmup > 0 else 0
    h = langevin_entry(cfg, P, chunk, n_warm, n_timed, dev, rank, world, args.noise_level, args.group)
    K, W = h["K"], n_warm * chunk
    # the job's only collective: gather the saved frames (xGMI); not part of a "step"
    gather_ms = 0.0
    if use_dist:
        import torch.distributed as dist
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fr = h["frames"] if dist.get_backend() == "nccl" else h["frames"].cpu()
        out = [torch.empty_like(fr) for _ in range(world)]
        dist.all_gather(out, fr)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t1)

    also = None
    if not args.no_extras:
        also = {}
        try:    # secondary figures must never cost the headline line
            # chignolin_langevin_fp32_engine: the headline workload with every weight GEMM on the fp32 matrix pipe (DFF_SPLIT_BF16=0)
            for name, c2, P2, ch2, nt, f32e in (("villin_langevin", "villin", 256, 250, MIN_LAUNCHES, False),
                                                ("protein_g_langevin", "protein_g", 128, 250, MIN_LAUNCHES, False),
                                                ("chignolin_langevin_fp32_engine", "chignolin", 256, 250, MIN_LAUNCHES, True),
                                                # the reference's published protocol: --parallel_sim 100 (evaluate/sampling_commands.md:13)
                                                ("chignolin_langevin_p100", "chignolin", 100, 250, MIN_LAUNCHES, False)):
                e = langevin_entry(c2, P2, ch2, 1, nt, dev, rank, world, fp32_engine=f32e)
                also[name] = {
                    "workload": f"{c2} ({e['N']} beads, H={e['H']}, L={e['L']}) Langevin, {P2}/GPU" + (", DFF_SPLIT_BF16=0" if f32e else ""),
                    "dtype": kernel_dtype(e["kernel"]),
                    "value": round(world * e["K"] / e["elapsed"], 2), "unit": f"MD-steps/s (batch {P2}, whole job)",
                    "ms_per_step": round(1e3 * e["elapsed"] / e["K"], 5), "steps": e["K"], "finite": e["finite"],
                    "roofline": roofline(c2, P2, ch2, e["launch_ms"], e["kernel"], brief=True)}
            # chignolin_iid_512: BASELINE configs[2] in its own per-GPU shape (batch 4096 over 8 GPUs, sample.py:185-189)
            for name, c2, P2, nt in (("chignolin_iid", "chignolin", 256, MIN_LAUNCHES), ("chignolin_iid_512", "chignolin", 512, 4),
                                     ("villin_iid", "villin", 256, 4)):
                e = iid_entry(c2, P2, 1, nt, dev, rank, world)
                also[name] = {
                    "workload": f"{c2} iid, batch {P2}/GPU, complete 1000-step reverse chains", "dtype": kernel_dtype(e["kernel"]),
                    "value": round(world * P2 * nt / e["elapsed"], 2), "unit": "samples/s (whole job)",
                    "ms_per_reverse_step": round(1e3 * e["elapsed"] / (nt * 1000), 5), "chains_timed": nt,
                    "roofline": roofline(c2, P2, 1000, e["launch_ms"], e["kernel"], brief=True)}
        except Exception as ex:  # noqa: BLE001
            also["error"] = f"{type(ex).__name__}: {ex}"[:300]

    if rank == 0:
        N, H, L = h["N"], h["H"], h["L"]
        traj_steps = world * P * K / h["elapsed"]
        value = traj_steps / 256.0
        res = {
            "metric": f"Langevin MD-steps/sec at batch 256 ({cfg}, score fwd+VJP + BAOAB per step)",
            "value": value, "unit": "MD-steps/s (batch-256 steps, whole job)", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": 1e3 * h["elapsed"] / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": kernel_dtype(h["kernel"]),
            "data": "synthetic (seeded weights, N(0,1) centred x0, in-kernel Philox noise)",
            "steps_requested": args.steps, "warmup_requested": args.warmup,
            "config": {"workload": f"BASELINE configs[1]: {cfg} ({N} beads, H={H}, L={L}) Langevin, parallel_sim={P}/GPU, "
                                   f"noise_level={args.noise_level}, save_interval={chunk}, 1 persistent launch per {chunk} steps",
                       "parallelism": f"{world} x independent trajectory shards (no data-path collective)",
                       "kernel": h["kernel"], "grid": h["grid"], "lds_bytes": h["lds"], "launches_timed": n_timed,
                       "collective_backend": backend},
            "trajectory_steps_per_s": traj_steps, "finite": h["finite"], "gather_ms": gather_ms,
            "roofline": roofline(cfg, P, chunk, h["launch_ms"], h["kernel"]),
        }
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(cfg, P, args.noise_level)
            res["speedup_vs_cpu_port"] = round(value / res["cpu_baseline"]["value"], 1)
            thr = res["cpu_baseline"]["cores"]

            def brief(cb):   # the `also` entries carry the figures only (kind / unit: see notes.cpu and the entry's own unit)
                return {"value": round(cb["value"], 4), "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"]}
            if also is not None and "villin_langevin" in also:    # ~5 s per step on the host: three steps, same thread count
                also["villin_langevin"]["cpu_baseline"] = brief(cpu_baseline("villin", 256, 20, budget_s=10.0, max_steps=3, threads=thr))
            if also is not None:    # the i.i.d. metric's CPU legs (north_star: "alongside the CPU-reference number")
                for name, c2, bud, mx in (("chignolin_iid", "chignolin", 8.0, 20), ("villin_iid", "villin", 8.0, 2)):
                    if name in also:
                        also[name]["cpu_baseline"] = brief(cpu_baseline_iid(c2, 256, budget_s=bud, max_steps=mx, threads=thr))
        if also is not None:
            res["also"] = also
        res["notes"] = NOTES
        line = json.dumps(res)
        print(line)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
