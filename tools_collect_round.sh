#!/bin/bash
# Condense one tools_evidence.sh run (gpurun_out/) into profiles/<round>/.  usage: tools_collect_round.sh r05
set -e
R=${1:-r06}
cd "$(dirname "$(readlink -f "$0")")"
P=tools_profile_report.py
python $P ${R}_head $R
python $P ${R}_villin $R/villin --cfg villin --beads 35 --bench-args "--cfg villin"
python $P ${R}_pg $R/protein_g --cfg protein_g --P 128 --beads 56 --bench-args "--cfg protein_g --parallel_sim 128"
python $P ${R}_trp_cage $R/trp_cage --cfg trp_cage --beads 20 --bench-args "--cfg trp_cage"
python $P ${R}_bba $R/bba --cfg bba --beads 28 --bench-args "--cfg bba"
python $P ${R}_ala2 $R/ala2 --cfg ala2 --beads 5 --bench-args "--cfg ala2"
python $P ${R}_iid_chig $R/iid_chignolin --chunk 1000 --steps 4000 --warmup 1000 --bench-args "--mode iid"
python $P ${R}_iid_chig512 $R/iid_chignolin_512 --P 512 --chunk 1000 --steps 4000 --warmup 1000 --bench-args "--mode iid --parallel_sim 512"
python $P ${R}_iid_villin $R/iid_villin --cfg villin --beads 35 --chunk 1000 --steps 4000 --warmup 1000 --bench-args "--mode iid --cfg villin"
for f in bench.json configs.jsonl configs_p128.jsonl e2e_cli.jsonl pair_check.txt stress_repeat.txt; do cp gpurun_out/${R}_$f profiles/$R/$f; done
cp gpurun_out/${R}_stages_chignolin.txt profiles/$R/stages.txt
for c in villin trp_cage bba protein_g; do cp gpurun_out/${R}_stages_$c.txt profiles/$R/$c/stages.txt; done
