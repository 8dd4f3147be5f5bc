#!/bin/bash
# per-kernel register / scratch summary of the device code: tools_regs.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-unused-result -DDFF_SMALL_MODE=${DFF_SMALL_MODE:-1} "$@" -S --cuda-device-only \
    two-for-one-diffusion_amd/csrc/${DFF_TU:-dff_small}.hip -o /tmp/dff_regs.s 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/dff_regs.s').read()
for blk in txt.split('  - .agpr_count:')[1:]:
    ag = blk.split('\n', 1)[0].strip()
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    sc = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1)
    sg = re.search(r'\.sgpr_count:\s+(\d+)', blk).group(1)
    print(f"{name[:60]:60s} vgpr+agpr={vg:>4s} agpr={ag:>4s} sgpr={sg:>4s} scratch={sc:>5s} B/lane")
PY
