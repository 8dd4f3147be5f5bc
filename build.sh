#!/bin/bash
# Build libdff_amd.so (HIP kernels + C ABI) for gfx950, in-tree.  Three translation units compiled in
# parallel: the <= 64-row kernel, the <= 16-row kernel, and the host half (ABI, dispatch, PWD kernels).
#   DFF_EXTRA_FLAGS="-DDFF_FAST_BUILD"   development build: headline variants only (fast to compile)
set -e
cd "$(dirname "$(readlink -f "$0")")"
SRC=two-for-one-diffusion_amd/csrc
OUT=two-for-one-diffusion_amd/libdff_amd.so
OBJ=build/obj
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result ${DFF_EXTRA_FLAGS}"
pids=()
for tu in dff_kernels dff_small dff_host; do
    # rebuild a unit only when one of the sources is newer than its object (or the flags changed)
    stamp="$OBJ/$tu.flags"
    if [ ! -f "$OBJ/$tu.o" ] || [ "$(cat $stamp 2>/dev/null)" != "$FLAGS" ] || \
       [ -n "$(find $SRC include -newer $OBJ/$tu.o \( -name '*.hip' -o -name '*.h' \) | head -1)" ]; then
        ( hipcc $FLAGS -c $SRC/$tu.hip -o $OBJ/$tu.o.tmp && mv $OBJ/$tu.o.tmp $OBJ/$tu.o && echo "$FLAGS" > $stamp ) &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/dff_kernels.o $OBJ/dff_small.o $OBJ/dff_host.o -o $OUT
echo "built $OUT"
