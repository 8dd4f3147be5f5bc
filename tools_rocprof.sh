#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box; outputs under gpurun_out/prof_<tag>/).
# usage: tools_rocprof.sh <tag> [bench args...]
# Counters are collected in their own runs (no trace domains together with --pmc).
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python $ROOT/bench.py --steps 1000 --warmup 250 --no-cpu --no-extras $*"
# which sources the profiled library was built from (dff_version(); tools_profile_report.py stamps it into traffic.json)
(cd $ROOT && python -c "import dff_amd; from dff_amd import srcsha; print(srcsha.library_sha())") > $OUT/src_sha.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
i=0
if [ "${DFF_PMC_SETS:-all}" = "traffic" ]; then   # HBM bytes only: FETCH_SIZE and WRITE_SIZE, each in its own pass
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
  done
else
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
done
fi
python - <<PY
import csv, glob, os, collections, json
out = "$OUT"
summ = {"kernels": [], "counters": {}}
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        summ["kernels"].append({k: r[k] for k in r})
for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dff_" in r.get("Kernel_Name", ""):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        summ["counters"][c] = {"kernel": k, "launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:6000])
PY
