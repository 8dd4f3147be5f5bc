// microbenchmark 4: does VALU work overlap BF16 MFMA work on a SIMD (it does not overlap fp32 MFMA, overlap2.hip)?
// 512-thread block = two waves per SIMD (w and w+4).  mode 0: all waves bf16 MFMA (8 x 16x16x32 per iteration);
// 1: all VALU (64 independent v_fma per iteration); 2: waves 0-3 bf16 MFMA, waves 4-7 VALU; 3: one wave per SIMD
// alternating 8 bf16 MFMA and 64 VALU in its own instruction stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode, int nwaves_active) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const float fa = threadIdx.x * 0.5f, fb = 1.0001f;
    const bool do_mfma = mode == 0 || mode == 3 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || mode == 3 || (mode == 2 && wave >= 4);
    const bool active = wave < nwaves_active;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (active)
        for (int it = 0; it < iters; ++it) {
            if (do_mfma) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 3] = MF(a, b, acc[m & 3]);
            }
            if (do_valu) {
#pragma unroll
                for (int q = 0; q < 64; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], fb, fa);
            }
        }
    const long long t1 = __builtin_readcyclecounter();
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
        {0, 4, "1 wave/SIMD: 8 bf16 MFMA"}, {1, 4, "1 wave/SIMD: 64 VALU"}, {3, 4, "1 wave/SIMD: 8 bf16 MFMA then 64 VALU"},
        {0, 8, "2 waves/SIMD: both 8 bf16 MFMA"}, {1, 8, "2 waves/SIMD: both 64 VALU"},
        {2, 8, "waves 0-3 bf16 MFMA, waves 4-7 VALU"}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, iters, c.mode, c.nw); hipDeviceSynchronize(); }
        long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-42s cycles/iter per wave:", c.name);
        for (int w = 0; w < 8; ++w) printf(" %6.1f", (double)h[w] / iters);
        printf("\n");
    }
    return 0;
}
