// What would one partial-sum exchange cost the <= 16-row kernel if a protein were split over two workgroups?
// Two blocks of a pair (b, b + 8: one XCD under the observed placement) each own a 10 x 64 fp32 partial tile (2.5 KB) in LDS and
// exchange it 13 times per "step" with the protocol of dff_fused_kernel<..., PAIR> (csrc/dff_kernels.hip pair_exchange: 16-byte
// stores -> s_waitcnt -> barrier -> flag store -> poll -> barrier -> 16-byte loads -> add -> barrier), a stand-in compute phase of
// ~WORK cycles in between.  Reported: us per step with the exchange, without it (same phases and barriers, no traffic), and the
// difference per exchange -- for the same-XCD form (plain stores acknowledged by the shared L2, sc1 loads) and the agent-scope form.
//   hipcc --offload-arch=gfx950 -O3 -w tools_ubench/pair_exchange.hip -o tools_ubench/pair_exchange.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 10, COLS = 64, NX = 13, NT = 512;

template <int MODE>   // 0: no exchange, 1: same-XCD (plain stores), 2: agent scope (sc1 stores)
__global__ __launch_bounds__(NT) void k(float* xchg, unsigned* flags, int steps, int work, unsigned long long* cyc, float* sink) {
    extern __shared__ float smem[];
    float* tile = smem;   // [ROWS][COLS]
    const int tid = threadIdx.x, b = blockIdx.x;
    const int unit = (b >> 4) * 8 + (b & 7), hf = (b >> 3) & 1;   // partner: b ^ 8
    for (int i = tid; i < ROWS * COLS; i += NT) tile[i] = 1.0f + 0.001f * i;
    __syncthreads();
    unsigned xseq = 0;
    const size_t slot = ROWS * COLS;
    float acc = 1.0f + tid * 1e-6f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s)
        for (int x = 0; x < NX; ++x) {
            // stand-in compute: a dependent FMA chain of ~`work` cycles per wave
            for (int i = 0; i < work / 8; ++i) acc = __builtin_fmaf(acc, 0.999999f, 1e-7f);
            if (tid < ROWS * COLS / 4) tile[4 * tid] += acc * 1e-9f;
            __syncthreads();
            if (MODE != 0) {
                float* const mine = xchg + ((size_t)(2 * unit + hf) * 2 + (xseq & 1)) * slot;
                const float* const theirs = xchg + ((size_t)(2 * unit + (1 - hf)) * 2 + (xseq & 1)) * slot;
                unsigned* const fl = flags + 2 * unit;
                if (tid < ROWS * COLS / 4) {
                    const f32x4 v = *(const f32x4*)(tile + 4 * tid);
                    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine + 4 * tid), "v"(v) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + 4 * tid), "v"(v) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    if (MODE == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(fl + hf), "v"(xseq + 1) : "memory");
                    else __hip_atomic_store(fl + hf, xseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    while (__hip_atomic_load(fl + (1 - hf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < xseq + 1) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 22)) break;
                    }
                }
                __syncthreads();
                if (tid < ROWS * COLS / 4) {
                    f32x4 pv;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pv) : "v"(theirs + 4 * tid) : "memory");
                    *(f32x4*)(tile + 4 * tid) = *(const f32x4*)(tile + 4 * tid) * 0.5f + pv * 0.5f;
                }
                __syncthreads();
                ++xseq;
            }
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[b] = t1 - t0;
    if (tid < ROWS * COLS / 4) sink[b * 256 + tid] = tile[4 * tid] + acc;
}

template <int MODE>
static double run(int nb, int steps, int work, float* xchg, unsigned* flags, unsigned long long* cyc, float* sink) {
    hipMemset(flags, 0, 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(NT), 150 * 1024, 0, xchg, flags, 2, work, cyc, sink);   // warm
    hipDeviceSynchronize();
    hipMemset(flags, 0, 1024 * sizeof(unsigned));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(NT), 150 * 1024, 0, xchg, flags, steps, work, cyc, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / steps;   // us per step
}

int main() {
    float *xchg, *sink; unsigned* flags; unsigned long long* cyc;
    hipMalloc(&xchg, (size_t)256 * 2 * 2 * ROWS * COLS * sizeof(float));
    hipMalloc(&flags, 1024 * sizeof(unsigned));
    hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    hipMalloc(&sink, 256 * 256 * sizeof(float));
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int steps = 400;
    for (int nb : {256, 64}) {          // 128 pairs (every CU busy) / 32 pairs (the P = 32-per-GPU case)
        for (int work : {1500, 3500}) { // cycles of compute between exchanges (a step of the <= 16-row kernel: 13 x ~3.5 k at half the work)
            const double t0 = run<0>(nb, steps, work, xchg, flags, cyc, sink);
            const double t1 = run<1>(nb, steps, work, xchg, flags, cyc, sink);
            const double t2 = run<2>(nb, steps, work, xchg, flags, cyc, sink);
            printf("blocks %3d work %4d cyc: no exchange %7.2f us/step | same-XCD %7.2f (+%5.2f us per exchange) | agent scope %7.2f (+%5.2f us per exchange)\n",
                   nb, work, t0, t1, (t1 - t0) / NX, t2, (t2 - t0) / NX);
        }
    }
    return 0;
}
