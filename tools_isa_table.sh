#!/bin/bash
# profiles/<round>/regs.txt: the ISA hygiene table of every shipped kernel variant (tools_isa_stats.py over hipcc -S listings of the
# four device translation units, built with build.sh's flags).  usage: tools_isa_table.sh r05      (takes a few minutes of CPU)
set -e
cd "$(dirname "$(readlink -f "$0")")"
R=${1:-r06}
SRC=two-for-one-diffusion_amd/csrc
SHA=$(cat $(ls $SRC/* | LC_ALL=C sort) include/dff.h | sha256sum | cut -c1-16)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result"
S="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers"
mkdir -p build/isa
for k in 0 1 2; do hipcc $F -DDFF_SMALL_MODE=$k $S -S --cuda-device-only $SRC/dff_small.hip -o build/isa/small_m$k.s 2>/dev/null & done
hipcc $F ${DFF_KERNELS_SCHED--mllvm -disable-machine-licm} -S --cuda-device-only $SRC/dff_kernels.hip -o build/isa/kernels.s 2>/dev/null &
wait
{
  echo "# ISA hygiene table of every shipped kernel variant (tools_isa_stats.py over hipcc -S listings built with build.sh's flags), round ${R#r}, sources $SHA"
  echo "# columns: registers, VGPR / SGPR spill counts, scratch bytes per lane | wide loads by addressing form (64-bit VGPR pair / saddr; buffer_load not counted) | 64-bit & quarter-rate integer VALU | packed-f32 VALU between the first and last MFMA | v_readlane+v_writelane, MFMA count"
  for tu in kernels small_m0 small_m1 small_m2; do echo "## $tu"; python3 tools_isa_stats.py build/isa/$tu.s; done
} > profiles/$R/regs.txt
echo "wrote profiles/$R/regs.txt ($SHA)"
