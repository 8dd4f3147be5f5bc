// microbenchmark: can fp32 MFMA (16x16x4) and independent VALU FMAs overlap on one SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NM, int NV>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((k<NM, NV>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NM, NV>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s NM=%2d NV=%3d : %.1f cycles/iter (s_memtime units)\n", name, NM, NV, (double)c / iters);
}
int main() {
    run<4, 0>("mfma only");
    run<0, 32>("valu only");
    run<4, 32>("mfma + 32 valu");
    run<4, 16>("mfma + 16 valu");
    run<4, 64>("mfma + 64 valu");
    run<8, 0>("8 mfma");
    run<8, 64>("8 mfma + 64 valu");
    run<0, 64>("64 valu");
    return 0;
}
