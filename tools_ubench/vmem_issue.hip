// vmem_issue.hip -- is ISSUING a burst of global loads free for the issuing wave?  (round 5, the "parking" experiment)
// 256 workgroups x NW waves; every wave writes N independent global_load_dwordx4 back to back (N KiB per wave, L2-resident
// image), reads the clock right after the last one is ISSUED (no s_waitcnt in between) and again once all have returned.
//   hipcc --offload-arch=gfx950 -O3 -w tools_ubench/vmem_issue.hip -o tools_ubench/vmem_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ buf, size_t per_wave16, int reps, unsigned* sink, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = buf + (size_t)wave * per_wave16 + lane;
    unsigned long long t_issue = 0, t_all = 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        u32x4 v[N];
        const unsigned long long t0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < N; ++d) v[d] = p[(size_t)((r * N + d) % 48) * 64];
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_readcyclecounter();   // all N loads issued, none waited for
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < N; ++d) acc ^= v[d];
        asm volatile("" ::"v"(acc));
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0; t_all += t2 - t0;
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
    if (blockIdx.x == 0 && lane == 0) { out[2 * wave] = t_issue; out[2 * wave + 1] = t_all; }
}
template <int N>
void run(const u32x4* buf, size_t per_wave16, int nw, unsigned* sink, unsigned long long* out) {
    const int reps = 200;
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(64 * nw), 0, 0, buf, per_wave16, reps, sink, out);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double iss = 0, all = 0;
    for (int w = 0; w < nw; ++w) { iss += h[2 * w]; all += h[2 * w + 1]; }
    printf("%d waves x %2d loads (%2d KiB / wave): issue %6.0f cycles / wave, issue + return %6.0f  -> %.1f B/clk/CU over the burst\n", nw, N, N,
           iss / nw / reps, all / nw / reps, (double)N * 1024 * nw / (all / nw / reps));
}
int main() {
    const size_t per_wave16 = 48 * 64;   // 48 KiB per wave, 384 KiB per workgroup image, shared by all workgroups
    u32x4* buf; unsigned* sink; unsigned long long* out;
    hipMalloc(&buf, per_wave16 * 8 * 16); hipMemset(buf, 1, per_wave16 * 8 * 16);
    hipMalloc(&sink, 4); hipMalloc(&out, 16 * 8);
    for (int nw : {1, 8}) {
        run<4>(buf, per_wave16, nw, sink, out); run<12>(buf, per_wave16, nw, sink, out); run<24>(buf, per_wave16, nw, sink, out);
    }
    return 0;
}
