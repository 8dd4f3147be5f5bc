// Issue rate of the fp16 MFMA shapes on one SIMD (cycles per instruction, 4 independent accumulator chains): is the K = 16 form
// (v_mfma_f32_16x16x16_f16) half the cost of the K = 32 one on gfx950, and how does either compare with v_mfma_f32_16x16x4_f32?
//   hipcc --offload-arch=gfx950 -O3 -w tools_ubench/mfma_f16_rates.hip -o tools_ubench/mfma_f16_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4] = {};
    f16x4 a4 = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)threadIdx.x}, b4 = a4;
    f16x8 a8 = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)threadIdx.x, (_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f}, b8 = a8;
    float af = 1.0f + threadIdx.x, bf = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[c], 0, 0, 0);
                if (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[c], 0, 0, 0);
                if (KIND == 2) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[c], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int KIND> void run(const char* name, int threads) {
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %d wave(s) per SIMD: %.1f cycles per instruction per wave\n", name, threads / 256, (double)h / (iters * 16.0));
}
int main() {
    for (int th : {256, 512}) {
        run<0>("v_mfma_f32_16x16x16_f16", th);
        run<1>("v_mfma_f32_16x16x32_f16", th);
        run<2>("v_mfma_f32_16x16x4_f32", th);
    }
    return 0;
}
