#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools_rocprof.sh (gpurun_out/prof_<tag>/) into the committed
evidence under profiles/<round>/: the kernel-stats and counter CSVs, summary.json, traffic.json (what
bench.py reports as roofline.traffic) and a README.md with the derived figures.

usage: tools_profile_report.py <tag> <round-dir> [--cfg chignolin --P 256 --chunk 250 --steps 1000 --warmup 250]
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.abspath(__file__))
MFMA_CYCLES = 32  # v_mfma_f32_16x16x4_f32 on one SIMD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("round")
    ap.add_argument("--cfg", default="chignolin")
    ap.add_argument("--P", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=250)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=250)
    ap.add_argument("--beads", type=int, default=10)
    ap.add_argument("--bench-args", default="", help="extra bench.py arguments the profile was taken with")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}")
    dst = os.path.join(ROOT, "profiles", a.round)
    os.makedirs(dst, exist_ok=True)
    summ = json.load(open(os.path.join(src, "summary.json")))
    shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, "summary.json"))
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, "trace_kernel_stats.csv"))
    for i in range(1, 10):
        for f in glob.glob(os.path.join(src, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
            # keep the dff kernel rows only (the torch fill / copy kernels are noise)
            rows = list(csv.DictReader(open(f)))
            keep = [r for r in rows if "dff_" in r.get("Kernel_Name", "")]
            if keep:
                with open(os.path.join(dst, f"pmc{i}_counter_collection.csv"), "w", newline="") as o:
                    w = csv.DictWriter(o, fieldnames=list(keep[0].keys()))
                    w.writeheader()
                    w.writerows(keep)

    # The timed launches are the full-grid ones; one-off helper launches of the same kernel (the layer-0
    # table build: one workgroup, one step) are reported separately and kept out of the per-launch means.
    disp = []
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dff_" in r["Kernel_Name"]:
                disp.append((r["Kernel_Name"].replace("void ", "").split("(")[0].strip(), int(r["Grid_Size_X"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    # (the main kernel = the (name, grid) group with the largest total time: the layer-0 table builder of an i.i.d. run
    # launches the SCORE-mode kernel of the same family over an equally large grid)
    groups = {}
    for d in disp:
        groups.setdefault((d[0], d[1]), []).append(d)
    (kname, gmax), grp = max(groups.items(), key=lambda kv: sum(x[2] for x in kv[1]))
    # ... and, where every mode runs ONE kernel (the <= 64-row kernel), the launches at least half as long as the longest
    dmax = max(d[2] for d in grp)
    main = [d for d in grp if d[2] >= 0.5 * dmax]
    main_ord = {i for i, d in enumerate([d for d in disp if (d[0], d[1]) == (kname, gmax)]) if d[2] >= 0.5 * dmax}
    # per-launch duration = the MEDIAN of the main launches: one launch in ten lands on a box hiccup (round 3: one 14.9 ms
    # launch among eight of 13.2-13.3), and the mean then disagrees with bench.py's HIP-event figure by more than 1 %
    durs = sorted(d[2] for d in main)
    avg_ms = durs[len(durs) // 2] if len(durs) % 2 else 0.5 * (durs[len(durs) // 2 - 1] + durs[len(durs) // 2])
    mean_ms = sum(durs) / len(durs)
    total_ms = sum(float(k["TotalDurationNs"]) for k in summ["kernels"]) / 1e6
    kern = [{"Name": kname, "Calls": len(main), "AverageNs": avg_ms * 1e6,
             "Percentage": 100.0 * sum(d[2] for d in main) / total_ms}]
    helpers = [d for d in disp if (d[0], d[1]) != (kname, gmax) or d[2] < 0.5 * dmax]
    if helpers:
        kern.append({"Name": helpers[0][0] + " (layer-0 table build, grid %d threads)" % helpers[0][1], "Calls": len(helpers),
                     "AverageNs": sum(d[2] for d in helpers) / len(helpers) * 1e6,
                     "Percentage": 100.0 * sum(d[2] for d in helpers) / total_ms})
    C = {}
    for i in range(1, 10):
        for f in glob.glob(os.path.join(src, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
            acc, seen = {}, {}
            for r in csv.DictReader(open(f)):
                if "dff_" in r.get("Kernel_Name", "") and int(r["Grid_Size"]) == gmax and \
                        r["Kernel_Name"].replace("void ", "").split("(")[0].strip() == kname:
                    # the same launches in the same order as in the trace run: keep the main ones (by ordinal of the dispatch)
                    o = seen.setdefault(r["Counter_Name"], [0])
                    if o[0] in main_ord:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    o[0] += 1
            for c, v in acc.items():
                C[c] = sum(v) / len(v)
    fetch = C.get("FETCH_SIZE", 0.0) * 1024 * 2  # KiB units, x2 on gfx950 (MI355X_MICROARCH.md)
    write = C.get("WRITE_SIZE", 0.0) * 1024
    traffic = dict(workload=f"{a.cfg} P={a.P} chunk={a.chunk}", kernel=kname, hbm_bytes_per_launch=fetch + write,
                   fetch_bytes=fetch, write_bytes=write, avg_launch_ms=avg_ms, steps_per_launch=a.chunk)
    gui = C.get("GRBM_GUI_ACTIVE", 0.0)
    clock = gui / 8 / (avg_ms * 1e-3) / 1e9 if gui else float("nan")
    mfma_util = C.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8) if gui else float("nan")
    # what bench.py carries in its roofline objects next to the traffic (north_star: rocprof-reported HBM rate and MFMA utilisation)
    traffic["mean_launch_ms"] = mean_ms
    traffic["hbm_tbps"] = (fetch + write) / (avg_ms * 1e-3) / 1e12
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in C:
        traffic["mfma_busy"] = mfma_util
    if "TCC_HIT_sum" in C:
        traffic["l2_hit"] = C["TCC_HIT_sum"] / (C["TCC_HIT_sum"] + C["TCC_MISS_sum"])
    # the sources the profiled library was built from (tools_rocprof.sh asked the library itself); bench.py reports these
    # counters only next to a library with the same hash
    try:
        traffic["src_sha"] = open(os.path.join(src, "src_sha.txt")).read().strip() or "unknown"
    except OSError:
        # (the working tree's hash would certify counters that may have been taken on an older library)
        traffic["src_sha"] = "unknown"
        print(f"warning: no src_sha.txt in {src}: stamped 'unknown' -- bench.py will report this profile as stale")
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"))
    wc = C.get("SQ_WAVE_CYCLES", 1.0)
    lines = []
    lines.append(f"# profiles/{a.round} -- rocprofv3 of `python bench.py --steps {a.steps} --warmup {a.warmup} --no-cpu --no-extras{(' ' + a.bench_args) if a.bench_args else ''}` "
                 f"({a.cfg}, P={a.P}, 1 x MI355X)")
    lines.append("Collected by tools_rocprof.sh on the GPU box (kernel trace + stats in one run, each PMC set in its own "
                 "run) and condensed by tools_profile_report.py.")
    lines.append(f"Raw CSVs: profiles/{a.round}/trace_kernel_stats.csv, profiles/{a.round}/pmc*_counter_collection.csv; "
                 f"stage breakdown: profiles/{a.round}/stages.txt\n")
    lines.append("## kernel trace (--kernel-trace --stats)\n")
    lines.append(f"| kernel | calls | median ms / launch ({a.chunk} MD-steps) | us / MD-step | % of GPU time |")
    lines.append("|---|---|---|---|---|")
    for k in kern:
        nm = k["Name"]
        lines.append(f"| {nm} | {k['Calls']} | {float(k['AverageNs']) / 1e6:.3f} | "
                     f"{float(k['AverageNs']) / 1e3 / a.chunk:.1f} | {float(k['Percentage']):.2f} |")
    lines.append(f"\n## PMC (mean per launch = {a.chunk} MD-steps x {a.P} trajectories)\n")
    lines.append("| counter | value |")
    lines.append("|---|---|")
    for c in sorted(C):
        lines.append(f"| {c} | {C[c]:.4e} |")
    lines.append("\nDerived:")
    lines.append(f"- effective shader clock = GRBM_GUI_ACTIVE/8 XCDs / launch time = {clock:.2f} GHz")
    if "SQ_INSTS_MFMA" in C:
        per = C["SQ_INSTS_MFMA"] / (a.chunk * a.P)
        lines.append(f"- MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles) = {mfma_util:.3f}  "
                     f"({per:.0f} MFMA instructions per trajectory-step; v_mfma_f32_16x16x4_f32: {MFMA_CYCLES} cycles each, "
                     f"v_mfma_f32_16x16x32_bf16 of the split-bf16 variants: 16)")
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in C and C.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            lines.append(f"- VALU co-executing with an MFMA: SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES = "
                         f"{C['SQ_VALU_MFMA_COEXEC_CYCLES'] / C['SQ_VALU_MFMA_BUSY_CYCLES']:.3f}")
    if "SQ_INSTS_VALU" in C and "SQ_INSTS_MFMA" in C:
        lines.append(f"- instruction mix per launch: VALU {C['SQ_INSTS_VALU']:.3e}, MFMA {C['SQ_INSTS_MFMA']:.3e}, "
                     f"LDS {C.get('SQ_INSTS_LDS', 0):.3e}, VMEM rd {C.get('SQ_INSTS_VMEM_RD', 0):.3e} / wr {C.get('SQ_INSTS_VMEM_WR', 0):.3e}, "
                     f"SALU {C.get('SQ_INSTS_SALU', 0):.3e}")
    if "SQ_WAIT_ANY" in C:
        lines.append(f"- wave cycles: waiting (s_waitcnt/barrier) {C['SQ_WAIT_ANY'] / wc:.2f}, issue-stalled "
                     f"{C.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}, issuing {C.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
    if "TCC_HIT_sum" in C:
        lines.append(f"- L2 hit rate = {C['TCC_HIT_sum'] / (C['TCC_HIT_sum'] + C['TCC_MISS_sum']):.3f}")
    state = a.chunk * a.P * 60 * a.beads
    lines.append(f"- HBM traffic per launch: read {fetch / 1e9:.2f} GB (FETCH_SIZE x 1024 x 2, gfx950 correction of "
                 f"MI355X_MICROARCH.md) + write {write / 1e9:.2f} GB = {(fetch + write) / 1e9:.2f} GB -> "
                 f"{(fetch + write) / (avg_ms * 1e-3) / 1e12:.2f} TB/s; algorithmic state traffic is only "
                 f"{state / 1e9:.2f} GB" + (": the rest is the per-workgroup activation stash spilling out of L2" if fetch + write > 20 * state else
                                           ": what is left on top of it is the weights' first touch per XCD, the layer-0 table and the saved frames "
                                           "(no activation goes through the stash in this kernel's sampling loops)"))
    if "SQ_LDS_BANK_CONFLICT" in C:
        lines.append(f"- LDS: bank-conflict cycles / active cycles = "
                     f"{C['SQ_LDS_BANK_CONFLICT'] / max(C.get('SQ_LDS_IDX_ACTIVE', 1.0), 1.0):.2f}")
    extra = ""
    old = os.path.join(dst, "README.md")
    if os.path.exists(old) and "## other kernels" in open(old).read() and "/" not in a.round:
        extra = "\n" + open(old).read()[open(old).read().index("## other kernels"):]
    open(old, "w").write("\n".join(lines) + "\n" + extra)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
