#!/bin/bash
# Build libdff_amd.so (HIP kernels + C ABI) for gfx950, in-tree.
set -e
cd "$(dirname "$(readlink -f "$0")")"
SRC=two-for-one-diffusion_amd/csrc
OUT=two-for-one-diffusion_amd/libdff_amd.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude \
      -Wno-unused-result ${DFF_EXTRA_FLAGS} \
      $SRC/dff_host.hip -o $OUT
echo "built $OUT"
