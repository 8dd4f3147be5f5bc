#!/bin/bash
# Development harness for kernel experiments on the <= 16-row kernel (not part of the product).
#   tools_exp.sh build <name> "<extra hipcc flags>"   -> build/exp/<name>/libdff_amd.so  (dff_small.hip rebuilt with
#                                                        -DDFF_FAST_BUILD + flags, linked with the objects of ./build.sh)
#   tools_exp.sh run <name>...                         (on the GPU box) bench headline + smoke check per variant,
#                                                        results appended to gpurun_out/exp.jsonl
set -e
cd "$(dirname "$(readlink -f "$0")")"
cmd=$1; shift
SRC=two-for-one-diffusion_amd/csrc
if [ "$cmd" = build ]; then
    name=$1; flags=$2
    d=build/exp/$name; mkdir -p $d
    md=${DFF_EXP_MODE:-1}   # the sampler mode whose kernel is rebuilt (1 = Langevin, the headline); the others come from ./build.sh
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result -DDFF_FAST_BUILD -DDFF_SMALL_MODE=$md ${DFF_SMALL_SCHED--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers} $flags -c $SRC/dff_small.hip -o $d/dff_small_m$md.o
    objs=""
    for k in 0 1 2; do if [ $k = $md ]; then objs="$objs $d/dff_small_m$k.o"; else objs="$objs build/obj/dff_small_m$k.o"; fi; done
    # its own host half: dff_version() of an experiment names it (src=exp-<name>: never the product build's hash) and its flags;
    # the stage ticks need the host half built with them too
    printf '#define DFF_BUILD_FLAGS "%s"\n' "$(printf '%s' "$flags" | sed 's/[\\"]/\\&/g')" > $d/dff_build_info.h
    pf=""; case "$flags" in *DFF_PROF=1*) pf="-DDFF_PROF=1";; esac
    case "$flags" in *DFF_EXPERIMENT*) pf="$pf -DDFF_EXPERIMENT";; esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result $pf -DDFF_SRC_SHA=exp-$name -include $d/dff_build_info.h -c $SRC/dff_host.hip -o $d/dff_host.o
    host=$d/dff_host.o
    hipcc --offload-arch=gfx950 -shared -fPIC build/obj/dff_kernels.o $objs $host -o $d/libdff_amd.so
    echo "$flags" > $d/flags
    echo "built $d ($flags)"
elif [ "$cmd" = buildk ]; then   # the <= 64-row kernel TU instead (e.g. flags: -DDFF_FAST_BUILD -DDFF_ONLY="VAR_SPW(128,3,1)")
    name=$1; flags=$2
    d=build/exp/$name; mkdir -p $d
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result ${DFF_KERNELS_SCHED--mllvm -disable-machine-licm} $flags -c $SRC/dff_kernels.hip -o $d/dff_kernels.o &
    host=build/obj/dff_host.o
    case "$flags" in *DFF_PROF=1*)   # the host half refuses dff_debug_profile unless it was built with the stage ticks too
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result -DDFF_PROF=1 -c $SRC/dff_host.hip -o $d/dff_host.o; host=$d/dff_host.o;; esac
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC $d/dff_kernels.o build/obj/dff_small_m0.o build/obj/dff_small_m1.o build/obj/dff_small_m2.o $host -o $d/libdff_amd.so
    echo "$flags" > $d/flags
    echo "built $d ($flags)"
elif [ "$cmd" = buildall ]; then   # EVERY translation unit with the flags (e.g. prof -DDFF_PROF=1: the stage profiles of tools_evidence.sh)
    name=$1; flags=$2
    d=build/exp/$name; mkdir -p $d
    printf '#define DFF_BUILD_FLAGS "%s"\n' "$(printf '%s' "$flags" | sed 's/[\\"]/\\&/g')" > $d/dff_build_info.h
    C="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result $flags"
    $C ${DFF_KERNELS_SCHED--mllvm -disable-machine-licm} -c $SRC/dff_kernels.hip -o $d/dff_kernels.o &
    for k in 0 1 2; do $C -DDFF_SMALL_MODE=$k ${DFF_SMALL_SCHED--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers} -c $SRC/dff_small.hip -o $d/dff_small_m$k.o & done
    $C -DDFF_SRC_SHA=exp-$name -include $d/dff_build_info.h -c $SRC/dff_host.hip -o $d/dff_host.o &
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC $d/dff_kernels.o $d/dff_small_m0.o $d/dff_small_m1.o $d/dff_small_m2.o $d/dff_host.o -o $d/libdff_amd.so
    echo "$flags" > $d/flags
    echo "built $d ($flags)"
elif [ "$cmd" = run ]; then
    mkdir -p gpurun_out
    for name in "$@"; do
        lib=$PWD/build/exp/$name/libdff_amd.so
        [ "$name" = base ] && lib=$PWD/two-for-one-diffusion_amd/libdff_amd.so
        for rep in 1 2; do
            DFF_LIB_PATH=$lib ${DFF_EXP_ENV} python bench.py --no-cpu --no-extras --steps ${DFF_EXP_STEPS:-2000} ${DFF_EXP_ARGS} 2>gpurun_out/exp_$name.err | \
                python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'name':'$name','rep':$rep,'us_per_step':1e3*r['ms_per_step'],'frac':r['roofline']['frac'],'finite':r.get('finite'),'kernel':r['config']['kernel'],'flags':open('build/exp/$name/flags').read().strip() if '$name'!='base' else ''}))" | tee -a gpurun_out/exp.jsonl
        done
    done
fi
