// l2stream.hip -- what does a CU get out of L2 when every CU streams the same few MB (the weight stream of the
// <= 16-row sampler kernel), as a function of the bytes each wave keeps in flight?
// 256 workgroups x 8 waves; wave w of every workgroup reads the same slice of a `MB`-sized buffer `reps` times with D
// independent 16-byte loads per lane outstanding (D KiB per wave).  Prints bytes / clock / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(512) void stream(const u32x4* __restrict__ buf, size_t n16_per_wave, int reps, unsigned* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = buf + (size_t)wave * n16_per_wave + lane;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        for (size_t i = 0; i < n16_per_wave; i += 64 * D) {
            u32x4 v[D];
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = p[i + 64 * d];
#pragma unroll
            for (int d = 0; d < D; ++d) acc ^= v[d];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] == 0x12345678u) out[0] = acc[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main(int argc, char** argv) {
    const double MB = argc > 1 ? atof(argv[1]) : 5.0;
    const int reps = 50;
    size_t n16_per_wave = (size_t)(MB * 1024 * 1024 / 16 / 8);
    n16_per_wave = n16_per_wave / (64 * 48) * (64 * 48);
    u32x4* buf; unsigned* out; unsigned long long* cyc;
    hipMalloc(&buf, n16_per_wave * 8 * 16); hipMemset(buf, 1, n16_per_wave * 8 * 16);
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    auto run = [&](auto kern, int D) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, buf, n16_per_wave, 2, out, cyc);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, buf, n16_per_wave, reps, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double bytes = (double)n16_per_wave * 8 * 16 * reps;
        printf("%.1f MB image, %2d KiB/wave in flight: %.1f B/clk/CU (s_memtime), %.1f GB/s/CU, %.2f TB/s chip\n", MB, D,
               bytes / (double)c, bytes / (ms * 1e-3) / 1e9, bytes * 256 / (ms * 1e-3) / 1e12);
    };
    run(stream<4>, 4); run(stream<8>, 8); run(stream<12>, 12); run(stream<16>, 16); run(stream<24>, 24); run(stream<48>, 48);
    return 0;
}
