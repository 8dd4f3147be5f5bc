#!/usr/bin/env python3
"""HBM-roofline measurement of the PWD histogram kernels (dff_pwd_max + dff_pwd_hist) at the output
sizes of BASELINE.json's configs.  Algorithmic bytes per pass = n * N * 12 (the structures, read once);
the histograms (n_pairs x bins x 4 B) are noise next to that.  Prints one JSON line per config."""
import json
import sys
import time

import torch

import dff_amd
from dff_amd import binding

PEAK_HBM = 8.0e12
CASES = [("chignolin config 2 (10240 x 10)", 10, 10240), ("chignolin iid config 3 (100000 x 10)", 10, 100000),
         ("villin config 4 (819200 x 35)", 35, 819200), ("protein G config 5 (409600 x 56)", 56, 409600)]


def ev_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    dff_amd.load_library()
    for name, N, n in CASES:
        x = torch.randn((n, N, 3), device="cuda") * 5.0
        mx = binding.pwd_max(x, 3)
        nb = (torch.div(mx.cpu(), 0.1, rounding_mode="floor") + 1).to(torch.int32)
        hmax = torch.tensor([0.1 * int(b) for b in nb], dtype=torch.float64).float()
        nb_d, hm_d = nb.cuda(), hmax.cuda()
        t_max = ev_time(lambda: binding.pwd_max(x, 3), 20)
        t_hist = ev_time(lambda: binding.pwd_hist(x, 3, nb_d, hm_d), 20)
        byts = n * N * 12
        # CPU: what the reference does (distance matrix + histc per pair), bounded sample
        ns = min(n, 20000)
        xc = x[:ns].cpu()
        t0 = time.time()
        d = torch.norm(xc[:, :, None, :] - xc[:, None, :, :], dim=-1)
        ti = torch.triu_indices(N, N, offset=3)
        pw = d[:, ti[0], ti[1]]
        for p in range(pw.shape[1]):
            torch.histc(pw[:, p], bins=int(nb[p]), min=0, max=0.1 * int(nb[p]))
        t_cpu = (time.time() - t0) * n / ns
        print(json.dumps({"workload": name, "pairs": int(nb.numel()), "max_bins": int(nb.max()),
                          "max_ms": t_max * 1e3, "hist_ms": t_hist * 1e3,
                          "max_GBps": byts / t_max / 1e9, "hist_GBps": byts / t_hist / 1e9,
                          "hist_frac_hbm": byts / t_hist / PEAK_HBM,
                          "pair_evals_per_s": n * int(nb.numel()) / t_hist,
                          "cpu_torch_s_extrapolated": t_cpu, "cpu_threads": torch.get_num_threads()}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
