#!/bin/bash
# Build libdff_amd.so (HIP kernels + C ABI) for gfx950, in-tree.  Five translation units compiled in
# parallel: the <= 64-row kernel, the <= 16-row kernel once per sampler mode (score / Langevin / DDPM), and the host half
# (ABI, dispatch, PWD kernels).
#   DFF_EXTRA_FLAGS="-DDFF_FAST_BUILD"   development build: headline variants only (fast to compile)
set -e
cd "$(dirname "$(readlink -f "$0")")"
SRC=two-for-one-diffusion_amd/csrc
OUT=two-for-one-diffusion_amd/libdff_amd.so
OBJ=build/obj
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result ${DFF_EXTRA_FLAGS}"
# what the device code was built from: sha256 over csrc/* (sorted) + include/dff.h (+ non-default flags), first 16 hex digits.  dff_version() carries
# it, tools_profile_report.py writes it into every traffic.json, bench.py drops profile counters whose hash is not the library's
# (two-for-one-diffusion_amd/srcsha.py computes the same hash from the tree).
# Non-default compile flags (DFF_EXTRA_FLAGS, a scheduler override) are part of what the device code is built from: they are hashed
# in (empty for the product build, whose hash stays the tree's) and dff_version() names them (flags=[...]).
BUILD_FLAGS="${DFF_EXTRA_FLAGS}${DFF_SMALL_SCHED+ sched:${DFF_SMALL_SCHED}}${DFF_KERNELS_SCHED+ ksched:${DFF_KERNELS_SCHED}}"
SRC_SHA=$( (cat $(ls $SRC/* | LC_ALL=C sort) include/dff.h; printf '%s' "$BUILD_FLAGS") | sha256sum | cut -c1-16)
printf '#define DFF_BUILD_FLAGS "%s"\n' "$(printf '%s' "$BUILD_FLAGS" | sed 's/[\\"]/\\&/g')" > $OBJ/dff_build_info.h.tmp
cmp -s $OBJ/dff_build_info.h.tmp $OBJ/dff_build_info.h || mv $OBJ/dff_build_info.h.tmp $OBJ/dff_build_info.h
pids=()
for tu in dff_kernels dff_small_m0 dff_small_m1 dff_small_m2 dff_host; do
    # rebuild a unit only when one of the sources is newer than its object (or the flags changed)
    stamp="$OBJ/$tu.flags"
    src=$tu; extra=""
    # the <= 16-row kernels are scheduled for ILP with the AMDGPU register-pressure trackers (measured on the headline kernel:
    # 54.2 -> 53.1 us / step; the same switches LOSE 1 - 2 % on the <= 64-row kernels, which keep the default strategy)
    case $tu in dff_host) extra="-DDFF_SRC_SHA=$SRC_SHA -include $OBJ/dff_build_info.h";; esac
    # the <= 64-row kernels without machine-level loop-invariant code motion (round 6): hoisted scalars are what its 104 SGPRs
    # spill to VGPR lanes -- v_readlane in the head loops 147 -> 71, scratch 68 -> 36 B on villin's variant; villin -0.7 %,
    # trp-cage -0.65 %, BBA / protein G +-0.2 % (profiles/r06/qkv_all_heads/compiler_flags.txt); it LOSES 1.6 % on the <= 16-row kernel
    case $tu in dff_kernels) extra="${DFF_KERNELS_SCHED--mllvm -disable-machine-licm}";; esac
    case $tu in dff_small_m*) src=dff_small; extra="-DDFF_SMALL_MODE=${tu#dff_small_m} ${DFF_SMALL_SCHED--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers}";; esac
    if [ ! -f "$OBJ/$tu.o" ] || [ "$(cat $stamp 2>/dev/null)" != "$FLAGS $extra" ] || \
       [ -n "$(find $SRC include -newer $OBJ/$tu.o \( -name '*.hip' -o -name '*.h' \) | head -1)" ]; then
        ( hipcc $FLAGS $extra -c $SRC/$src.hip -o $OBJ/$tu.o.tmp && mv $OBJ/$tu.o.tmp $OBJ/$tu.o && echo "$FLAGS $extra" > $stamp ) &
        pids+=($!)
    fi
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo "compile failed"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/dff_kernels.o $OBJ/dff_small_m0.o $OBJ/dff_small_m1.o $OBJ/dff_small_m2.o $OBJ/dff_host.o -o $OUT
echo "built $OUT"
