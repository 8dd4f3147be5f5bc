#!/bin/bash
# One rocprofv3 --pmc pass over bench.py for a development library (on the GPU box): per-launch mean of each counter for the
# main kernel, appended to gpurun_out/pmc.jsonl.   usage: tools_pmc.sh <exp name | base> "<counters>" [bench args...]
set -u
NAME=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$ROOT/build/exp/$NAME/libdff_amd.so; [ "$NAME" = base ] && LIB=$ROOT/two-for-one-diffusion_amd/libdff_amd.so
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_$NAME; rm -rf $OUT
DFF_LIB_PATH=$LIB rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python $ROOT/bench.py --steps 1000 --warmup 250 --no-cpu --no-extras "$@" > $OUT.log 2>&1
python - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dff_" in r.get("Kernel_Name", ""):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
best = max(agg.items(), key=lambda kv: sum(kv[1]))[0][:2] if agg else None
res = {"name": "$NAME", "args": "$*", "kernel": best[0] if best else None}
for (k, g, c), v in agg.items():
    if (k, g) == best:
        v = sorted(v)[len(v) // 2:]          # the long launches (250 steps), not the layer-0 table builds
        res[c] = sum(v) / len(v)
print(json.dumps(res))
open("$ROOT/gpurun_out/pmc.jsonl", "a").write(json.dumps(res) + "\n")
PY
