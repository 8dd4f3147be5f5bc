#!/bin/bash
# Everything profiles/<round>/ is made of, in one call on the GPU box (outputs under gpurun_out/; condense with
# tools_profile_report.py afterwards).  usage: tools_evidence.sh <round tag, e.g. r04>
set -u
TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools_bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2>/dev/null
python tools_bench_configs.py --P 128 > gpurun_out/${TAG}_configs_p128.jsonl 2>/dev/null
python tools_e2e_cli.py > gpurun_out/${TAG}_e2e_cli.jsonl 2>/dev/null
python tests/pair_check.py protein_g villin trp_cage bba > gpurun_out/${TAG}_pair_check.txt 2>&1
python tools_stress_repeat.py > gpurun_out/${TAG}_stress_repeat.txt 2>&1
bash tools_rocprof.sh ${TAG}_head > /dev/null 2>&1
bash tools_rocprof.sh ${TAG}_villin --cfg villin > /dev/null 2>&1
bash tools_rocprof.sh ${TAG}_pg --cfg protein_g --parallel_sim 128 > /dev/null 2>&1
for c in trp_cage bba ala2; do bash tools_rocprof.sh ${TAG}_$c --cfg $c > /dev/null 2>&1; done
# stage profiles (needs build/exp/prof/libdff_amd.so: every translation unit with -DDFF_PROF=1)
if [ -f build/exp/prof/libdff_amd.so ]; then
  for c in chignolin villin trp_cage bba; do DFF_LIB_PATH=$PWD/build/exp/prof/libdff_amd.so python tools_profile_stages.py --cfg $c --waves 0,7 > gpurun_out/${TAG}_stages_$c.txt 2>/dev/null; done
  DFF_LIB_PATH=$PWD/build/exp/prof/libdff_amd.so python tools_profile_stages.py --cfg protein_g --P 128 > gpurun_out/${TAG}_stages_protein_g.txt 2>/dev/null
fi
DFF_PMC_SETS=traffic bash tools_rocprof.sh ${TAG}_iid_chig --mode iid --steps 4000 --warmup 1000 > /dev/null 2>&1
DFF_PMC_SETS=traffic bash tools_rocprof.sh ${TAG}_iid_chig512 --mode iid --parallel_sim 512 --steps 4000 --warmup 1000 > /dev/null 2>&1
DFF_PMC_SETS=traffic bash tools_rocprof.sh ${TAG}_iid_villin --mode iid --cfg villin --steps 4000 --warmup 1000 > /dev/null 2>&1
ls gpurun_out | head -40
