#!/usr/bin/env python3
"""Look inside fuzz cases of the other input branches that land outside their bar (tests/fuzz_shapes.py; `-m gpu` box): per sampled
trajectory, the kernel's forces, the reference twin's float32 run and the factorised algebra of oracle/kernel_model_gen.py in numpy
float32, all against the twin's float64 run -- is the distance of an outlier case spread over the samples or one sample's, and does
the same algebra in numpy float32 show it too?   usage: python tests/gen_outliers.py <seed> <case> [<case> ...]   (test infrastructure)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_shapes as fz  # noqa: E402
from oracle import kernel_model_gen as kg, reference_twin as twin  # noqa: E402

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "kernel_model_gen.py")).read()
ns32 = {}
exec(compile(src.replace("np.float64", "np.float32").replace("from .kernel_model import", "from oracle.kernel_model import"), "gen32", "exec"), ns32)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


def hook(case, tag, model, params, x, t, sub, f, fl, cons, L):
    N = x.shape[1]
    xs_, ts_ = torch.from_numpy(x[sub]), torch.from_numpy(t[sub])
    r64 = twin.score(twin.to_torch(params, torch.float64), xs_.double(), ts_.double(), L, conservative=cons, flags=fl).numpy()
    r32 = twin.score(twin.to_torch(params), xs_, ts_, L, conservative=cons, flags=fl).numpy()
    fw64 = kg.fold_weights(params, L, N, fl[0], fl[1], fl[2])
    fw32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v) for k, v in fw64.items()}
    fw32["layers"] = [{k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v) for k, v in l.items()} for l in fw64["layers"]]
    xc = x[sub] - x[sub].mean(1, keepdims=True)
    e32, st32 = ns32["forward"](fw32, xc.astype(np.float32), t[sub].astype(np.float32))
    m32 = -ns32["backward"](fw32, xc.astype(np.float32), st32)
    e64, st64 = kg.forward(fw64, xc.astype(np.float64), t[sub].astype(np.float64))
    m64 = -kg.backward(fw64, xc.astype(np.float64), st64)
    if os.environ.get("GEN_OUTLIERS_DUMP"):   # the case's inputs for CPU-side analysis
        np.savez(os.path.join(os.environ["GEN_OUTLIERS_DUMP"], f"case{case}.npz"), x=x[sub], t=t[sub], flags=np.array(fl), L=L, N=N, H=tag["H"],
                 **{"p_" + k: v for k, v in params.items()})
    print(f"case {case} {tag}: whole sample set: kernel {rel(f[sub], r64):.3e}  twin f32 {rel(r32, r64):.3e}  numpy-f32 factorised {rel(m32, r64):.3e}"
          f"  (fp64 factorised vs twin f64 {rel(m64, r64):.1e})  |F|max {np.abs(r64).max():.3e}")
    for k, b in enumerate(sub):
        print(f"   sample {b:4d} t={t[b]:.3f} |x|max={np.abs(x[b]).max():7.2f} |F|={np.linalg.norm(r64[k]):.3e}:  kernel {rel(f[b], r64[k]):.3e}"
              f"  twin f32 {rel(r32[k], r64[k]):.3e}  numpy-f32 factorised {rel(m32[k], r64[k]):.3e}")
    # the same inputs on the other engine
    for split in ("0", "1"):
        os.environ["DFF_SPLIT_BF16"] = split
        from dff_amd.score import GraphTransformer
        m2 = GraphTransformer(N, model.hidden if hasattr(model, "hidden") else tag["H"], device="cuda:0", n_layers=L, use_intrinsic_coords=fl[0],
                              use_abs_coords=fl[2], use_distances=fl[1], conservative=cons, state_dict=params)
        f2 = m2.native.score(torch.from_numpy(x[sub]).cuda(), torch.from_numpy(t[sub]).cuda()).cpu().numpy()
        print(f"   DFF_SPLIT_BF16={split} on the sampled set alone: {m2.native.last_launch()[0]}  {rel(f2, r64):.3e}")


if __name__ == "__main__":
    seed = int(sys.argv[1]); cases = [int(a) for a in sys.argv[2:]]
    fz.run(max(cases) + 1, seed, log=lambda m: None, only=set(cases), hook=hook)
