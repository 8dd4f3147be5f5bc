// microbenchmark 2: two waves on one SIMD (waves w and w+4 of a 512-thread block): does the VALU work of
// one overlap the fp32 MFMA work of the other?  mode 0: both MFMA; 1: both VALU; 2: wave<4 MFMA, wave>=4 VALU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode, int nwaves_active) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool active = wave < nwaves_active || (nwaves_active == 1 && wave == 0);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (active) {
        if (do_mfma) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 64; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 20000;
    struct { int mode, nw; const char* name; } cases[] = {
        {0, 4, "4 waves (1/SIMD) MFMA (8 per iter)"}, {1, 4, "4 waves (1/SIMD) VALU (64 per iter)"},
        {0, 8, "8 waves (2/SIMD) all MFMA"}, {1, 8, "8 waves (2/SIMD) all VALU"},
        {2, 8, "waves 0-3 MFMA, 4-7 VALU"}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, iters, c.mode, c.nw);
            hipDeviceSynchronize();
        }
        long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-40s cycles/iter per wave:", c.name);
        for (int w = 0; w < 8; ++w) printf(" %6.1f", (double)h[w] / iters);
        printf("\n");
    }
    return 0;
}
