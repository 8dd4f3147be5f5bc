// microbenchmark 3: could the weight GEMMs run on the bf16 MFMA at fp32 accuracy?  An fp32 value splits
// exactly into three bf16 pieces (8 + 8 + 8 significand bits, truncation split); keeping the six products of
// order >= 2^-16 reproduces the fp32 product to ~2^-23.  Compared here, for 16 x K x 16 products of N(0,1)
// data against an fp64 host result: v_mfma_f32_16x16x4_f32, the 6-term split (one accumulator; three
// accumulators summed small-to-large) and the 3-term split (two pieces).  Also: issue cost per 16x16x32 block.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split8(const float* a, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hh[2], mm[2], ll[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v = a[2 * p + q];
            const unsigned uh = __float_as_uint(v) & 0xffff0000u;
            const float r = v - __uint_as_float(uh);
            const unsigned um = __float_as_uint(r) & 0xffff0000u;
            const float r2 = r - __uint_as_float(um);
            hh[q] = uh; mm[q] = um; ll[q] = __float_as_uint(r2);
        }
        h[p] = __builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u);
        m[p] = __builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u);
        l[p] = __builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u);
    }
}
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// out[mode][16][16]; one wave
__global__ void acc_kernel(const float* A, const float* B, int K, float* out) {
    const int lane = threadIdx.x, rc = lane & 15, kg = lane >> 4;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0, e0 = d0, e1 = d0, e2 = d0;
    for (int k = 0; k < K; k += 4) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rc * K + k + kg], B[(k + kg) * 16 + rc], d0, 0, 0, 0);
    for (int kb = 0; kb < K; kb += 32) {
        float a[8], b[8];
        for (int j = 0; j < 8; ++j) { a[j] = A[rc * K + kb + kg * 8 + j]; b[j] = B[(kb + kg * 8 + j) * 16 + rc]; }
        u32x4 ah, am, al, bh, bm, bl;
        split8(a, ah, am, al); split8(b, bh, bm, bl);
        d1 = MF(al, bh, d1); d1 = MF(ah, bl, d1); d1 = MF(am, bm, d1); d1 = MF(am, bh, d1); d1 = MF(ah, bm, d1); d1 = MF(ah, bh, d1);
        e0 = MF(al, bh, e0); e0 = MF(ah, bl, e0); e0 = MF(am, bm, e0); e1 = MF(am, bh, e1); e1 = MF(ah, bm, e1); e2 = MF(ah, bh, e2);
        d3 = MF(am, bh, d3); d3 = MF(ah, bm, d3); d3 = MF(ah, bh, d3);
    }
    d2 = e2 + (e1 + e0);
    for (int r = 0; r < 4; ++r) {
        const int o = (kg * 4 + r) * 16 + rc;
        out[o] = d0[r]; out[256 + o] = d1[r]; out[512 + o] = d2[r]; out[768 + o] = d3[r];
    }
}

// issue cost: per iteration 4 independent 16x16x32 blocks (accumulators), mode 0: 8 fp32 MFMA each; 1: 6 bf16 MFMA each;
// 2: mode 1 + re-splitting one A fragment (8 floats) per block
__global__ __launch_bounds__(512) void cost_kernel(float* out, long long* cyc, int iters, int mode) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 0.001f + j;
    u32x4 ah, am, al;
    split8(a, ah, am, al);
    const u32x4 bh = ah + 3u, bm = am + 5u, bl = al + 7u;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], a[(k + t) & 7], acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (mode == 2) { a[t] += acc[t][0]; split8(a, ah, am, al); }
                acc[t] = MF(al, bh, acc[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MF(ah, bl, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MF(am, bm, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MF(am, bh, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MF(ah, bm, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MF(ah, bh, acc[t]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static double gauss() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}
int main() {
    srand(1);
    const int Ks[] = {64, 128, 512, 1664};
    const char* names[] = {"fp32 mfma 16x16x4", "bf16 x6, one accumulator", "bf16 x6, three accumulators", "bf16 x3 (two pieces)"};
    for (int K : Ks) {
        std::vector<float> A(16 * K), B(K * 16), O(1024);
        for (auto& x : A) x = (float)gauss();
        for (auto& x : B) x = (float)gauss();
        float *dA, *dB, *dO;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dO);
        hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
        for (int mode = 0; mode < 4; ++mode) {
            double emax = 0, e2 = 0, r2 = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ref = 0, sc = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; sc += fabs((double)A[i * K + k] * B[k * 16 + j]); }
                const double e = O[mode * 256 + i * 16 + j] - ref;
                emax = fmax(emax, fabs(e) / sc); e2 += e * e; r2 += ref * ref;
            }
            printf("K=%4d %-30s max|err|/sum|a||b| = %.3e   rms err / rms = %.3e\n", K, names[mode], emax, sqrt(e2 / r2));
        }
        hipFree(dA); hipFree(dB); hipFree(dO);
    }
    float* out; long long* cyc;
    hipMalloc(&out, 2048); hipMalloc(&cyc, 64);
    const char* cn[] = {"8 x fp32 16x16x4 per block", "6 x bf16 16x16x32 per block", "6 x bf16 + split of 8 floats per block"};
    for (int nthr : {256, 512})
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(cost_kernel, dim3(1), dim3(nthr), 0, 0, out, cyc, 5000, mode); hipDeviceSynchronize(); }
            long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("%d waves/SIMD  %-42s cycles per 16x16x32 block per wave: %.1f\n", nthr / 256, cn[mode], h[0] / 5000.0 / 4);
        }
    return 0;
}
