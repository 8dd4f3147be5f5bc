#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box; outputs under gpurun_out/prof_<tag>/).
# usage: tools_rocprof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $ROOT/bench.py --steps 500 --warmup 250 --no-cpu $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1
done
cd $OUT && find . -name "*.csv" | head -50
python - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in sorted(glob.glob(out + "/**/*kernel_stats.csv", recursive=True)):
    print("==", os.path.relpath(f, out)); print(open(f).read()[:1500])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dff_fused" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, out))
    for k, v in agg.items():
        print(f"   {k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
