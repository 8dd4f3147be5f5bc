// dff_pwd.hip -- pairwise-distance histograms for the PWD Jensen-Shannon metric on the GPU.
//
// Replaces, for (n, N, 3) structures already resident in HBM:
//   get_pwd_triu_batch            evaluate/evaluators.py:934-948   d_ij = ||x_i - x_j||, j >= i + offset
//   the per-pair max + torch.histc of PwdEvaluator.__init__ / js_divergence_pwd   :238-247, :258-263
// without ever materialising the (n, n_pairs) distance matrix the reference builds (config 4: 819200 x 528
// floats = 1.7 GB).  The tiny (n_pairs x bins) Jensen-Shannon reduction stays on the host (evaluate.py).
//
// Arithmetic is pinned to what the reference computes with torch (CPU, float32), so the counts are
// integers equal to the reference's:
//   distance  sqrt(fma(dz, dz, fma(dy, dy, dx * dx)))          (vector_norm over the last dim)
//   bin       (int64)((d - 0) * nbins / (max - 0)) in float32, bins == nbins folded into the last bin,
//             values outside [0, max] dropped                   (histc's linear bin selection)
//
// This is HBM / atomic bound work, not MFMA work: x is streamed once per pass in coalesced tiles through
// LDS; histograms are privatised in LDS per (pair chunk, sample chunk) workgroup and flushed with one
// global atomic per non-empty bin.  Workgroups that share a sample chunk are placed on the same XCD
// (blockIdx % 8) so that only the first of them reads the tile from HBM; the others hit that XCD's L2.
#pragma once
#include "dff_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DFF_PWD_THREADS 256
// the histogram kernel owns the whole CU's LDS (one workgroup per CU), so it brings 16 waves of its own:
// 4 per SIMD hide the LDS-read -> sqrt -> divide -> LDS-atomic chain of each evaluation
#define DFF_PWD_HIST_THREADS 1024
#define DFF_PWD_TILE 64          // structures per LDS tile (multiple of 4: float4-aligned tiles)
#define DFF_PWD_LDS_BINS 30720   // uint32 histogram slots per workgroup (120 KB)

// (i, j) of pair p in torch.triu_indices(N, N, offset) order
__device__ __forceinline__ void pwd_pair(int p, int N, int offset, int& i, int& j) {
    int row = 0, left = p;
    while (true) {
        const int cnt = N - row - offset;   // pairs in this row (> 0 while p is valid)
        if (left < cnt) break;
        left -= cnt;
        ++row;
    }
    i = row;
    j = row + offset + left;
}

// cooperative, coalesced load of `cnt` structures starting at s0 into LDS
template <int NT>
__device__ __forceinline__ void pwd_load_tile(float* tile, const float* __restrict__ x, long long s0, int cnt,
                                              int N3, bool vec4) {
    const float* src = x + s0 * N3;
    const int nf = cnt * N3;
    if (vec4) {
        const int n4 = nf >> 2;
        for (int k = threadIdx.x; k < n4; k += NT)
            ((f32x4*)tile)[k] = __builtin_nontemporal_load((const f32x4*)src + k);
        for (int k = (n4 << 2) + threadIdx.x; k < nf; k += NT) tile[k] = src[k];
    } else {
        for (int k = threadIdx.x; k < nf; k += NT) tile[k] = src[k];
    }
}

__device__ __forceinline__ float pwd_dist2(const float* xs, int oi, int oj) {
    const float* a = xs + oi;
    const float* b = xs + oj;
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    // sqrtf and '/' are correctly rounded here (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt);
    // the __fsqrt_rn / __fdiv_rn intrinsics are NOT (they lower to the fast native ops)
    return sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
}

// Thread layout of both kernels: PC (a power of two <= 256) pair lanes x threads/PC structure groups.  A
// thread keeps ITS pair's constants (bead indices, bin count, range, LDS histogram row) in registers and
// walks the structures of the tile with stride threads/PC: no index arithmetic per evaluation, the lanes of a
// wave read the same structure (LDS broadcast for shared beads) and update different histogram rows.

// per-pair maximum distance: max_out[p] = max_s d_p(s)   (bit pattern max: distances are >= 0)
#define DFF_PWD_MAXG 16   // pair groups per thread: n_pairs <= 16 * PC
__global__ __launch_bounds__(DFF_PWD_THREADS) void dff_pwd_max_kernel(const float* __restrict__ x, long long n,
                                                                       int N, int offset, int npairs, int pc_log2,
                                                                       long long chunk, unsigned* max_out, int vec4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N3 = 3 * N;
    float* tile = smem;                                      // DFF_PWD_TILE * N3
    const int PC = 1 << pc_log2, lane_p = threadIdx.x & (PC - 1), sgrp = threadIdx.x >> pc_log2;
    const int sstride = DFF_PWD_THREADS >> pc_log2;
    const int ngrp = (npairs + PC - 1) >> pc_log2;
    int oi[DFF_PWD_MAXG], oj[DFF_PWD_MAXG];
    float mx[DFF_PWD_MAXG];
#pragma unroll
    for (int g = 0; g < DFF_PWD_MAXG; ++g) {
        const int p = (g << pc_log2) + lane_p;
        int i = 0, j = 0;
        if (g < ngrp && p < npairs) pwd_pair(p, N, offset, i, j);
        oi[g] = 3 * i;
        oj[g] = 3 * j;
        mx[g] = 0.f;
    }
    const long long s_begin = (long long)blockIdx.x * chunk;
    const long long s_end = s_begin + chunk < n ? s_begin + chunk : n;
    for (long long s0 = s_begin; s0 < s_end; s0 += DFF_PWD_TILE) {
        const int cnt = (int)(s_end - s0 < DFF_PWD_TILE ? s_end - s0 : DFF_PWD_TILE);
        __syncthreads();
        pwd_load_tile<DFF_PWD_THREADS>(tile, x, s0, cnt, N3, vec4 != 0);
        __syncthreads();
        for (int s = sgrp; s < cnt; s += sstride) {
            const float* xs = tile + s * N3;
#pragma unroll
            for (int g = 0; g < DFF_PWD_MAXG; ++g)
                if (g < ngrp) mx[g] = fmaxf(mx[g], pwd_dist2(xs, oi[g], oj[g]));
        }
    }
    // structure groups -> one value per pair in LDS (the tile is dead), then ONE global atomic per pair
    __syncthreads();
    unsigned* red = (unsigned*)tile;   // npairs <= N (N + 1) / 2 < 192 N words
    for (int p = threadIdx.x; p < npairs; p += DFF_PWD_THREADS) red[p] = 0u;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < DFF_PWD_MAXG; ++g) {
        const int p = (g << pc_log2) + lane_p;
        if (g < ngrp && p < npairs && mx[g] > 0.f) atomicMax(&red[p], __float_as_uint(mx[g]));
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npairs; p += DFF_PWD_THREADS)
        if (red[p]) atomicMax(&max_out[p], red[p]);
}

// histograms of a chunk of PC pairs over a chunk of structures.  LDS: tile | hist[PC][ldl]
__global__ __launch_bounds__(DFF_PWD_HIST_THREADS) void dff_pwd_hist_kernel(const float* __restrict__ x, long long n,
                                                                        int N, int offset, int npairs,
                                                                        const int* __restrict__ nbins,
                                                                        const float* __restrict__ hmax, int ld,
                                                                        int pc_log2, int npc, long long chunk, int ldl,
                                                                        unsigned* hist, int vec4, int tile_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N3 = 3 * N;
    // XCD-aware placement: the npc workgroups of one structure chunk sit on one XCD, back to back
    const int w = blockIdx.x, xcd = w & 7, q = w >> 3;
    const int pchunk = q % npc;
    const long long schunk = (long long)(q / npc) * 8 + xcd;
    const int PC = 1 << pc_log2, lane_p = threadIdx.x & (PC - 1), sgrp = threadIdx.x >> pc_log2;
    const int sstride = DFF_PWD_HIST_THREADS >> pc_log2;
    const int p0 = pchunk << pc_log2;
    const int pc = npairs - p0 < PC ? npairs - p0 : PC;
    float* tile = smem;
    unsigned* hl = (unsigned*)(smem + tile_n * N3);          // PC * ldl  (tile_n structures per LDS tile: 64, 32 or 16)
    for (int k = threadIdx.x; k < pc * ldl; k += DFF_PWD_HIST_THREADS) hl[k] = 0u;
    const bool live = lane_p < pc;
    int i = 0, j = 0, b = 1;
    float m = 0.f;
    if (live) {
        pwd_pair(p0 + lane_p, N, offset, i, j);
        b = nbins[p0 + lane_p];
        m = hmax[p0 + lane_p];
    }
    const int oi = 3 * i, oj = 3 * j;
    const float bf = (float)b;
    unsigned* row = hl + lane_p * ldl;
    const long long s_begin = schunk * chunk;
    const long long s_end = s_begin + chunk < n ? s_begin + chunk : n;
    for (long long s0 = s_begin; s0 < s_end; s0 += tile_n) {
        const int cnt = (int)(s_end - s0 < tile_n ? s_end - s0 : tile_n);
        __syncthreads();
        pwd_load_tile<DFF_PWD_HIST_THREADS>(tile, x, s0, cnt, N3, vec4 != 0);
        __syncthreads();
        if (live)
#pragma unroll 4
            for (int s = sgrp; s < cnt; s += sstride) {
                const float d = pwd_dist2(tile + s * N3, oi, oj);
                if (d <= m) {   // NaN fails the test, d >= 0 always: histc's range check
                    int pos = (int)(long long)((d * bf) / m);
                    pos = pos < b ? pos : b - 1;
                    atomicAdd(&row[pos], 1u);
                }
            }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < pc * ldl; k += DFF_PWD_HIST_THREADS) {
        const unsigned v = hl[k];
        if (v) {
            const int p = k / ldl, bin = k - p * ldl;
            atomicAdd(&hist[(size_t)(p0 + p) * ld + bin], v);
        }
    }
}
