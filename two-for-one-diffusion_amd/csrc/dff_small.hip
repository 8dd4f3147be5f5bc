// dff_small.hip -- fast path of the fused sampler kernel for workgroups of <= 16 bead rows
// (chignolin 10 beads, alanine dipeptide 5 beads x up to 3 proteins, any N <= 16 model).
//
// Same algorithm and stash contract as dff_fused_kernel (dff_kernels.hip); what changes is the
// work decomposition, chosen so that nothing but the row-wise LayerNorm / gate stages needs a
// workgroup barrier:
//
//   * ONE ATTENTION HEAD PER WAVE.  Wave w owns heads {w, w+4}: it computes that head's q|u|k|v
//     tiles, the logits, softmax, the P.V product and the head's slice of the output projection
//     entirely out of its private LDS region; the 4 waves' partial output projections (a K-split
//     by head) are summed by the next row stage.  The FFN is split the same way (each wave owns
//     H of the 4H hidden columns: W1 slice -> GELU -> partial W2).
//   * x RIDES ALONG AS A 16-COLUMN HEAD EXTENSION.  Each 64-wide head is widened to 80 columns:
//         Q_ext = [q | u 0..]   K_ext = [k | x 0..]   V_ext = [v | x 0..]
//     so  Q_ext K_ext^T = q.k + u.x   (the folded edge bias of the logits),
//         P V_ext       = [sum_j a v_j | xbar]   (xbar - x_i = xrel goes through W_oc rows of W_o_ext),
//     and in the VJP  G_ext = dattn W_o_ext^T = [G | r],  dV_ext = P^T G_ext = [dv | sum_i a r],
//     dQ_ext = dS K_ext = [dq | du],  dK_ext = dS^T Q_ext = [dk | sum_i dS u]  -- i.e. u, xrel, r,
//     du and both x-gradient terms of SURVEY.md section 8a "Derived math" fall out of the same
//     16x16x4 MFMA tiles; no VALU contraction is left in the attention block.
//
// Everything dense runs on v_mfma_f32_16x16x4_f32; VALU work is softmax (in the MFMA C layout),
// LayerNorm, gates, GELU and the integrator update.
#pragma once
#include "dff_internal.h"

#define DFF_XH 80       // extended head width
#define DFF_XLD 84      // leading dim of the per-wave head buffers
#define DFF_PLD 20      // leading dim of the per-wave P / dS tiles
#define DFF_WREG (4 * 16 * DFF_XLD + 2 * 16 * DFF_PLD)   // floats per wave region

struct SmallStash {
    unsigned nodes_in, attn_out, ff, h_pre, qx, k, v, P;
    unsigned layer_stride, total;
};
__host__ __device__ inline SmallStash dff_small_stash(int N, int G, int H, int L) {
    SmallStash s;
    const unsigned R = (unsigned)(G * N), F = 4u * H;
    unsigned o = 0;
    s.nodes_in = o; o += R * H;
    s.attn_out = o; o += R * H;
    s.ff = o;       o += R * H;
    s.h_pre = o;    o += R * F;
    s.qx = o;       o += DFF_HEADS * R * DFF_XH;
    s.k = o;        o += DFF_HEADS * R * 64;
    s.v = o;        o += DFF_HEADS * R * 64;
    s.P = o;        o += DFF_HEADS * R * 16;
    s.layer_stride = o;
    s.total = (o * (unsigned)L + 63u) & ~63u;
    return s;
}

template <int H>
struct SmallLds {
    static constexpr int LH = H + 4;
    static constexpr unsigned xst = 0, xs = 64, dxs = 128, vst = 192, cm = 256, tn = 384, prof = 400,
                              dxw = 448,                      // [4][64] per-wave dx partials
                              abuf = 704, resbuf = abuf + 16 * LH, part = resbuf + 16 * LH,
                              wreg = part + 4 * 16 * LH, total = wreg + 4 * DFF_WREG + 64;
};

// ---------------------------------------------------------------- wave-private MFMA pieces
// C layout: acc[r] <-> (row 4*(lane>>4)+r, col lane&15)
DEVI void c_store(float* dst, int ld, int col0, const f32x4& acc, int rows) {
    const int lane = threadIdx.x & 63, quad = lane >> 4, col = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = quad * 4 + r;
        if (row < rows) dst[row * ld + col0 + col] = acc[r];
    }
}

template <int KB>
DEVI void load_afrag(f32x4 (&a)[KB], const float* A, int lda) {
    const int lane = threadIdx.x & 63;
    const float* ap = A + (lane & 15) * lda + 4 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) a[kb] = *(const f32x4*)(ap + 16 * kb);
}

// sequential tiles nt0 .. nt0+ntn-1 of ONE wave, K = 16*KB, ring of D tiles of B in flight
template <int KB, int NAUX, class Pre, class Epi>
DEVI void wv_wide(const f32x4 (&a)[KB], const float* __restrict__ Wp, int KBtot, int nt0, int ntn,
                  Pre pre, Epi epi) {
    constexpr int D = 4;
    static_assert(KB % 2 == 0, "KB even");
    const int lane = threadIdx.x & 63;
    const f32x4* wp = (const f32x4*)Wp + lane;
    f32x4 b[D][KB];
    float aux[D][NAUX];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < ntn) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b[d][kb] = wp[((size_t)(nt0 + d) * KBtot + kb) * 64];
            pre(d, aux[d]);
        }
    for (int i0 = 0; i0 < ntn; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < ntn) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; kb += 2)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb][s], b[d][kb][s], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb + 1][s], b[d][kb + 1][s], acc2, 0, 0, 0);
                    }
                float auxc[NAUX];
#pragma unroll
                for (int q = 0; q < NAUX; ++q) auxc[q] = aux[d][q];
                if (i + D < ntn) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) b[d][kb] = wp[((size_t)(nt0 + i + D) * KBtot + kb) * 64];
                    pre(i + D, aux[d]);
                }
                epi(i, acc + acc2, auxc);
            }
        }
    }
}

// acc[nt] += A(:, k-block kb) . W(k-block fw(kb), tile nt), kb = 0..nkb-1, for all NT tiles of an
// H-wide output; fa(kb) = this lane's A fragment address; ring of D k-blocks of B in flight.
template <int NT, class FA, class FW>
DEVI void wv_tall(f32x4 (&acc)[NT], int nkb, FA fa, FW fw, const float* __restrict__ Wp, int KBtot) {
    constexpr int D = NT <= 4 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const f32x4* wp = (const f32x4*)Wp + lane;
    f32x4 b[D][NT];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nkb) {
            const int kw = fw(d);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[d][nt] = wp[((size_t)nt * KBtot + kw) * 64];
        }
    for (int k0 = 0; k0 < nkb; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kb = k0 + d;
            if (kb < nkb) {
                const f32x4 a = *(const f32x4*)fa(kb);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[d][nt][s], acc[nt], 0, 0, 0);
                if (kb + D < nkb) {
                    const int kw = fw(kb + D);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[d][nt] = wp[((size_t)nt * KBtot + kw) * 64];
                }
            }
        }
    }
}

// C[i][j] = sum_k A[i][k] B[j][k], K = 80 (5 k-blocks), both operands rows of a head buffer
DEVI f32x4 wv_dot_rows(const float* A, const float* B) {
    const int lane = threadIdx.x & 63;
    const float* ap = A + (lane & 15) * DFF_XLD + 4 * (lane >> 4);
    const float* bp = B + (lane & 15) * DFF_XLD + 4 * (lane >> 4);
    f32x4 av[5], bv[5];
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) { av[kb] = *(const f32x4*)(ap + 16 * kb); bv[kb] = *(const f32x4*)(bp + 16 * kb); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][s], bv[0][s], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][s], bv[1][s], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2][s], bv[2][s], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3][s], bv[3][s], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4][s], bv[4][s], acc, 0, 0, 0);
    return acc + acc2;
}

// C[m][16nt+n] = sum_{k<16} Aop[m][k] B[k][16nt+n] for tiles nt in [NT0, NT1).
// TRANS = false: Aop[m][k] = T[m][k] (T = 16x16 tile, ld DFF_PLD);  true: Aop[m][k] = T[k][m].
template <int NT0, int NT1, bool TRANS, class Epi>
DEVI void wv_mm(const float* T, const float* B, Epi epi) {
    const int lane = threadIdx.x & 63, kk = lane >> 4, mm = lane & 15;
    float as[4];
    if (TRANS) {
#pragma unroll
        for (int s = 0; s < 4; ++s) as[s] = T[(4 * kk + s) * DFF_PLD + mm];
    } else {
        const f32x4 t = *(const f32x4*)(T + mm * DFF_PLD + 4 * kk);
#pragma unroll
        for (int s = 0; s < 4; ++s) as[s] = t[s];
    }
    float bv[NT1 - NT0][4];
#pragma unroll
    for (int nt = NT0; nt < NT1; ++nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[nt - NT0][s] = B[(4 * kk + s) * DFF_XLD + 16 * nt + mm];
    f32x4 acc[NT1 - NT0];
#pragma unroll
    for (int nt = 0; nt < NT1 - NT0; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nt = 0; nt < NT1 - NT0; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s], bv[nt][s], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = NT0; nt < NT1; ++nt) epi(nt, acc[nt - NT0]);
}

// ---------------------------------------------------------------- the kernel
template <int H>
__global__ __launch_bounds__(DFF_NTHREADS) void dff_small_kernel(const DffModelDev m, const DffRunArgs a) {
    using LL = SmallLds<H>;
    constexpr int LH = LL::LH, F = 4 * H, NT_H = H / 16, KB_H = H / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = lane >> 4, col = lane & 15;
    const int N = m.N, G = a.G;
    const int b0 = blockIdx.x * G;
    const int gcnt = min(G, a.B - b0);
    if (gcnt <= 0) return;
    const int rows = gcnt * N, RA = G * N;   // real rows of this workgroup / rows per stash slot
    float* const xst = smem + LL::xst; float* const xs = smem + LL::xs; float* const dxs = smem + LL::dxs;
    float* const vst = smem + LL::vst; float* const cm = smem + LL::cm; float* const tn = smem + LL::tn;
    float* const abuf = smem + LL::abuf; float* const resbuf = smem + LL::resbuf;
    float* const part = smem + LL::part;
    float* const dxw = smem + LL::dxw + wave * 64;
    float* const wr = smem + LL::wreg + wave * DFF_WREG;
    float* const Qx = wr; float* const Kx = wr + 16 * DFF_XLD; float* const Vx = wr + 2 * 16 * DFF_XLD;
    float* const Gx = wr + 3 * 16 * DFF_XLD;
    float* const pb = wr + 4 * 16 * DFF_XLD; float* const dsb = pb + 16 * DFF_PLD;
    float* const hbuf = wr;          // FFN hidden slice of this wave (aliases the head buffers)
    float* const mypart = part + wave * 16 * LH;
    const SmallStash sl = dff_small_stash(N, G, H, m.L);
    float* const stash = a.stash + (size_t)blockIdx.x * a.stash_stride;
    Ctx c;  // only what bead_mean() needs
    c.N = N; c.G = G; c.gcnt = gcnt; c.rows = rows;

    for (int i = tid; i < (int)LL::total; i += DFF_NTHREADS) smem[i] = 0.f;
    __syncthreads();
    Prof pf;
    pf.on = (a.prof != nullptr) && blockIdx.x == 0 && tid == 0;
    pf.acc = (unsigned long long*)(smem + LL::prof);
    pf.last = __builtin_readcyclecounter();

    // ---- load state (as dff_fused_kernel) ----
    {
        const float* xin = (a.mode == DFF_MODE_SCORE) ? a.x_in : a.x_io;
        if (tid < rows * 4) {
            const int row = tid >> 2, cc = tid & 3;
            float xv = 0.f, vv = 0.f;
            if (cc < 3) {
                const size_t gi = ((size_t)b0 * N + row) * 3 + cc;
                if (a.mode == DFF_MODE_DDPM && a.init_prior)
                    xv = philox_normal(a.seed, a.item_offset + b0 + row / N, 0xFFFFFFFFull, row % N, cc);
                else
                    xv = xin[gi];
                if (a.mode == DFF_MODE_LANGEVIN && !a.overdamped) vv = a.v_io[gi];
            }
            xst[tid] = xv;
            vst[tid] = vv;
        }
        if (tid < gcnt) tn[tid] = (a.mode == DFF_MODE_SCORE) ? a.tnorm[b0 + tid] : a.t_norm;
        __syncthreads();
        if (a.mode == DFF_MODE_DDPM && a.init_prior) {
            bead_mean(c, xst, cm);
            __syncthreads();
            if (tid < rows * 4) xst[tid] -= cm[(tid >> 2) / N * 4 + (tid & 3)];
            __syncthreads();
        }
    }
    // row-stage thread mapping: 16 lanes per row
    const int rrow = tid >> 4, sub = tid & 15;
    const bool ract = rrow < rows;
    constexpr int HC = H / 16;

    for (int step = 0; step < a.n_steps; ++step) {
        int t_int = 0;
        if (a.mode == DFF_MODE_DDPM) {
            t_int = a.t_start - step;
            if (tid < gcnt) tn[tid] = (1.0f * (float)t_int) / (float)m.T;
        }
        bead_mean(c, xst, cm);
        __syncthreads();
        if (tid < rows * 4) {
            const float xc = xst[tid] - cm[(tid >> 2) / N * 4 + (tid & 3)];
            if (a.mode == DFF_MODE_LANGEVIN) xst[tid] = xc;
            xs[tid] = xc;
        }
        dxw[lane] = 0.f;                              // every wave clears its own partial
        __syncthreads();
        if (a.mode == DFF_MODE_LANGEVIN) {
            bead_mean(c, xs, cm);
            __syncthreads();
            if (tid < rows * 4) xs[tid] -= cm[(tid >> 2) / N * 4 + (tid & 3)];
            __syncthreads();
        }
        pf.tick(0);

        // x extension values this lane writes into K_ext / V_ext: entries (row, 64+cc), 4 per lane
        // idx = lane + 64 e -> row = idx >> 4, cc = idx & 15
        const bool cached0 = (a.mode == DFF_MODE_LANGEVIN) && step > 0;

        // =============================== forward ===============================
        if (!cached0) {
            // node features of layer 0 -> resbuf, LN1 -> abuf
            for (int idx = tid; idx < rows * H; idx += DFF_NTHREADS) {
                const int row = idx / H, cl = idx - row * H;
                const int g = row / N, i = row - g * N;
                resbuf[row * LH + cl] = m.WnT[i * H + cl] + tn[g] * m.WnT[N * H + cl] + m.bn[cl];
            }
            __syncthreads();
        }
        for (int l = 0; l < m.L; ++l) {
            const DffLayerDev& lw = m.layer[l];
            float* const sb = stash + (size_t)l * sl.layer_stride;
            const bool cached = cached0 && l == 0;
            // ---- row stage A: (l == 0 only; later layers get LN1 fused into stage C) nodes -> stash, LN1 -> abuf
            if (l == 0) {
                if (cached) {
                    if (ract) {
#pragma unroll
                        for (int i = 0; i < HC; ++i) resbuf[rrow * LH + sub + 16 * i] = ld_nt(sb + sl.nodes_in + rrow * H + sub + 16 * i);
                    }
                } else if (ract) {
                    float x[HC];
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        x[i] = resbuf[rrow * LH + sub + 16 * i];
                        st_nt(sb + sl.nodes_in + rrow * H + sub + 16 * i, x[i]);
                    }
                    float mean, rstd;
                    ln_stats<H>(x, mean, rstd);
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + 16 * i;
                        abuf[rrow * LH + cl] = (x[i] - mean) * rstd * lw.ln1_g[cl] + lw.ln1_b[cl];
                    }
                }
                __syncthreads();
            }
            pf.tick(1);
            // ---- attention block: wave w owns heads w and w+4 ----
            {
                f32x4 afr[KB_H];
                if (!cached) load_afrag<KB_H>(afr, abuf, LH);
                f32x4 acc_o[NT_H];
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) acc_o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int hp = 0; hp < 2; ++hp) {
                    const int h = wave + 4 * hp;
                    float* const sq = sb + sl.qx + (size_t)h * RA * DFF_XH;
                    float* const sk = sb + sl.k + (size_t)h * RA * 64;
                    float* const sv = sb + sl.v + (size_t)h * RA * 64;
                    if (cached) {
                        // q_ext, k, v of layer 0 from the stash (rows x 80 / 64 / 64 floats)
                        for (int it = lane; it < rows * 20; it += 64) {
                            const int row = it / 20, c4 = it - row * 20;
                            *(f32x4*)(Qx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sq + row * DFF_XH + 4 * c4));
                        }
                        for (int it = lane; it < rows * 16; it += 64) {
                            const int row = it >> 4, c4 = it & 15;
                            *(f32x4*)(Kx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sk + row * 64 + 4 * c4));
                            *(f32x4*)(Vx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sv + row * 64 + 4 * c4));
                        }
                    } else {
                        // 13 tiles: [q 4 | u 1 | k 4 | v 4] of head h
                        wv_wide<KB_H, 1>(afr, lw.Wqkvx_p, KB_H, h * 13, 13,
                            [&](int t, float (&aux)[1]) { aux[0] = lw.bqkvx[(h * 13 + t) * 16 + col]; },
                            [&](int t, const f32x4& acc, const float (&aux)[1]) {
                                float* dl; float* ds; int ldS, c0;
                                if (t < 5)      { dl = Qx; ds = sq; ldS = DFF_XH; c0 = 16 * t; }
                                else if (t < 9) { dl = Kx; ds = sk; ldS = 64; c0 = 16 * (t - 5); }
                                else            { dl = Vx; ds = sv; ldS = 64; c0 = 16 * (t - 9); }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = quad * 4 + r;
                                    if (row < rows) {
                                        const float v = acc[r] + aux[0];
                                        dl[row * DFF_XLD + c0 + col] = v;
                                        st_nt(ds + row * ldS + c0 + col, v);
                                    }
                                }
                            });
                    }
                    // x extension of K and V: columns 64..79 = [x_j, 0...]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = lane + 64 * e, row = idx >> 4, cc = idx & 15;
                        const float xv = (cc < 3 && row < rows) ? xs[row * 4 + cc] : 0.f;
                        Kx[row * DFF_XLD + 64 + cc] = xv;
                        Vx[row * DFF_XLD + 64 + cc] = xv;
                    }
                    // logits (C layout: rows i = 4 quad + r, col j) + softmax over j
                    const f32x4 S = wv_dot_rows(Qx, Kx);
                    {
                        const int pj = col / N;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = quad * 4 + r;
                            const bool ok = (col < rows) && (i < rows) && (i / N == pj);
                            const float s = ok ? S[r] * 0.125f : -INFINITY;
                            float mx = s;
                            mx = fmaxf(mx, __shfl_xor(mx, 8, 16)); mx = fmaxf(mx, __shfl_xor(mx, 4, 16));
                            mx = fmaxf(mx, __shfl_xor(mx, 2, 16)); mx = fmaxf(mx, __shfl_xor(mx, 1, 16));
                            const float e = ok ? expf(s - mx) : 0.f;
                            const float den = grp16_sum(e);
                            const float p = den > 0.f ? e / den : 0.f;
                            pb[i * DFF_PLD + col] = p;
                            if (i < rows) st_nt(sb + sl.P + ((size_t)h * RA + i) * 16 + col, p);
                        }
                    }
                    // O_ext = P V_ext (5 tiles) -> Q region; extension columns become xrel = xbar - x_i
                    wv_mm<0, 5, false>(pb, Vx, [&](int nt, const f32x4& acc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = quad * 4 + r;
                            if (row < rows) {
                                float v = acc[r];
                                if (nt == 4 && col < 3) v -= xs[row * 4 + col];
                                Qx[row * DFF_XLD + 16 * nt + col] = v;
                            }
                        }
                    });
                    // partial output projection: acc_o += O_ext(16x80) W_o_ext[h]  (K = 80)
                    wv_tall<NT_H>(acc_o, 5,
                        [&](int kb) { return Qx + col * DFF_XLD + 4 * quad + 16 * kb; },
                        [&](int kb) { return h * 5 + kb; }, lw.Wox_p, DFF_HEADS * 5);
                }
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) c_store(mypart, LH, 16 * nt, acc_o[nt], rows);
            }
            __syncthreads();
            pf.tick(2);
            // ---- row stage B: attn_out = sum_w part + bo ; gate1 ; LN2 -> abuf ----
            if (ract) {
                float x[HC], res[HC], n1[HC];
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i, o = rrow * LH + cl;
                    x[i] = part[o] + part[16 * LH + o] + part[32 * LH + o] + part[48 * LH + o] + lw.bo[cl];
                    res[i] = resbuf[o];
                    st_nt(sb + sl.attn_out + rrow * H + cl, x[i]);
                }
                const float g = gate_value<H>(x, res, lw.g1, sub);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    n1[i] = x[i] * g + res[i] * (1.0f - g);
                    resbuf[rrow * LH + sub + 16 * i] = n1[i];
                }
                float mean, rstd;
                ln_stats<H>(n1, mean, rstd);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i;
                    abuf[rrow * LH + cl] = (n1[i] - mean) * rstd * lw.ln2_g[cl] + lw.ln2_b[cl];
                }
            }
            __syncthreads();
            pf.tick(3);
            // ---- FFN: wave w owns hidden columns [w H, (w+1) H) ----
            {
                f32x4 afr[KB_H];
                load_afrag<KB_H>(afr, abuf, LH);
                wv_wide<KB_H, 1>(afr, lw.W1_p, KB_H, wave * NT_H, NT_H,
                    [&](int t, float (&aux)[1]) { aux[0] = lw.b1[wave * H + 16 * t + col]; },
                    [&](int t, const f32x4& acc, const float (&aux)[1]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = quad * 4 + r;
                            if (row < rows) {
                                const float hp = acc[r] + aux[0];
                                st_nt(sb + sl.h_pre + (size_t)row * F + wave * H + 16 * t + col, hp);
                                hbuf[row * LH + 16 * t + col] = gelu_f(hp);
                            }
                        }
                    });
                f32x4 acc_f[NT_H];
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) acc_f[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                wv_tall<NT_H>(acc_f, KB_H,
                    [&](int kb) { return hbuf + col * LH + 4 * quad + 16 * kb; },
                    [&](int kb) { return wave * KB_H + kb; }, lw.W2_p, F / 16);
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) c_store(mypart, LH, 16 * nt, acc_f[nt], rows);
            }
            __syncthreads();
            pf.tick(4);
            // ---- row stage C: ff = sum_w part + b2 ; gate2 ; then next layer's LN1 or the energy head ----
            if (ract) {
                const bool last = l == m.L - 1;
                float x[HC], res[HC], n2[HC];
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i, o = rrow * LH + cl;
                    x[i] = part[o] + part[16 * LH + o] + part[32 * LH + o] + part[48 * LH + o] + lw.b2[cl];
                    res[i] = resbuf[o];
                    st_nt(sb + sl.ff + rrow * H + cl, x[i]);
                }
                const float g = gate_value<H>(x, res, lw.g2, sub);
#pragma unroll
                for (int i = 0; i < HC; ++i) n2[i] = x[i] * g + res[i] * (1.0f - g);
                if (last) {
                    float e = 0.f;
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + 16 * i;
                        e += n2[i] * m.wdec[cl];
                        resbuf[rrow * LH + cl] = m.wdec[cl];   // dn = d(sum e)/d nodes_L
                    }
                    if (a.energy_out) {
                        e = grp16_sum(e);
                        if (sub == 0) a.energy_out[(size_t)b0 * N + rrow] = e + m.bdec;
                    }
                } else {
                    const DffLayerDev& ln = m.layer[l + 1];
                    float* const sbn = stash + (size_t)(l + 1) * sl.layer_stride;
                    float mean, rstd;
                    ln_stats<H>(n2, mean, rstd);
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + 16 * i;
                        resbuf[rrow * LH + cl] = n2[i];
                        st_nt(sbn + sl.nodes_in + rrow * H + cl, n2[i]);
                        abuf[rrow * LH + cl] = (n2[i] - mean) * rstd * ln.ln1_g[cl] + ln.ln1_b[cl];
                    }
                }
            }
            __syncthreads();
            pf.tick(5);
        }

        // the stash written above is re-read below by other lanes / waves of this workgroup
        __threadfence_block();
        __syncthreads();
        // =============================== backward ===============================
        for (int l = m.L - 1; l >= 0; --l) {
            const DffLayerDev& lw = m.layer[l];
            const float* const sb = stash + (size_t)l * sl.layer_stride;
            // ---- row stage D: gate2 backward: dn (resbuf) -> dff (abuf), dn1 partial (resbuf) ----
            if (ract) {
                float ao[HC], nin[HC], n1[HC], ff[HC], dn[HC];
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i;
                    ao[i] = ld_nt(sb + sl.attn_out + rrow * H + cl);
                    nin[i] = ld_nt(sb + sl.nodes_in + rrow * H + cl);
                    ff[i] = ld_nt(sb + sl.ff + rrow * H + cl);
                    dn[i] = resbuf[rrow * LH + cl];
                }
                const float g1 = gate_value<H>(ao, nin, lw.g1, sub);
#pragma unroll
                for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + nin[i] * (1.0f - g1);
                const float g2 = gate_value<H>(ff, n1, lw.g2, sub);
                float dg = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) dg += dn[i] * (ff[i] - n1[i]);
                dg = grp16_sum(dg);
                const float dz = dg * g2 * (1.0f - g2);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i;
                    abuf[rrow * LH + cl] = dn[i] * g2 + dz * (lw.g2[cl] + lw.g2[2 * H + cl]);
                    resbuf[rrow * LH + cl] = dn[i] * (1.0f - g2) + dz * (lw.g2[H + cl] - lw.g2[2 * H + cl]);
                }
            }
            __syncthreads();
            pf.tick(6);
            // ---- FFN backward slice: dh = dff W2[:, slice] ; * gelu'(h_pre) ; partial df = dh_pre W1[slice, :] ----
            {
                f32x4 afr[KB_H];
                load_afrag<KB_H>(afr, abuf, LH);
                wv_wide<KB_H, 4>(afr, lw.W2T_p, KB_H, wave * NT_H, NT_H,
                    [&](int t, float (&aux)[4]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int row = quad * 4 + r;
                            row = row < rows ? row : rows - 1;
                            aux[r] = ld_nt(sb + sl.h_pre + (size_t)row * F + wave * H + 16 * t + col);
                        }
                    },
                    [&](int t, const f32x4& acc, const float (&aux)[4]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = quad * 4 + r;
                            if (row < rows) hbuf[row * LH + 16 * t + col] = acc[r] * gelu_grad_f(aux[r]);
                        }
                    });
                f32x4 acc_f[NT_H];
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) acc_f[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                wv_tall<NT_H>(acc_f, KB_H,
                    [&](int kb) { return hbuf + col * LH + 4 * quad + 16 * kb; },
                    [&](int kb) { return wave * KB_H + kb; }, lw.W1T_p, F / 16);
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) c_store(mypart, LH, 16 * nt, acc_f[nt], rows);
            }
            __syncthreads();
            pf.tick(7);
            // ---- row stage E: df = sum_w part ; LN2 backward ; gate1 backward -> dattn (abuf), dn_in partial (resbuf) ----
            if (ract) {
                float ao[HC], nin[HC], n1[HC], d1[HC], dyg[HC], xh[HC];
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i;
                    ao[i] = ld_nt(sb + sl.attn_out + rrow * H + cl);
                    nin[i] = ld_nt(sb + sl.nodes_in + rrow * H + cl);
                }
                const float g1 = gate_value<H>(ao, nin, lw.g1, sub);
#pragma unroll
                for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + nin[i] * (1.0f - g1);
                float mean, rstd;
                ln_stats<H>(n1, mean, rstd);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i, o = rrow * LH + cl;
                    xh[i] = (n1[i] - mean) * rstd;
                    dyg[i] = (part[o] + part[16 * LH + o] + part[32 * LH + o] + part[48 * LH + o]) * lw.ln2_g[cl];
                    s1 += dyg[i];
                    s2 += dyg[i] * xh[i];
                }
                s1 = grp16_sum(s1) * (1.0f / H);
                s2 = grp16_sum(s2) * (1.0f / H);
                float dg = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    d1[i] = resbuf[rrow * LH + sub + 16 * i] + rstd * (dyg[i] - s1 - xh[i] * s2);
                    dg += d1[i] * (ao[i] - nin[i]);
                }
                dg = grp16_sum(dg);
                const float dz = dg * g1 * (1.0f - g1);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + 16 * i;
                    abuf[rrow * LH + cl] = d1[i] * g1 + dz * (lw.g1[cl] + lw.g1[2 * H + cl]);
                    resbuf[rrow * LH + cl] = d1[i] * (1.0f - g1) + dz * (lw.g1[H + cl] - lw.g1[2 * H + cl]);
                }
            }
            __syncthreads();
            pf.tick(8);
            // ---- attention backward: wave w owns heads w and w+4 ----
            {
                f32x4 afr[KB_H];
                load_afrag<KB_H>(afr, abuf, LH);   // dattn
                f32x4 acc_a[NT_H];
#pragma unroll
                for (int nt = 0; nt < NT_H; ++nt) acc_a[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int hp = 0; hp < 2; ++hp) {
                    const int h = wave + 4 * hp;
                    const float* const sq = sb + sl.qx + (size_t)h * RA * DFF_XH;
                    const float* const sk = sb + sl.k + (size_t)h * RA * 64;
                    const float* const sv = sb + sl.v + (size_t)h * RA * 64;
                    // reload Q_ext, K, V, P of (l, h)
                    for (int it = lane; it < rows * 20; it += 64) {
                        const int row = it / 20, c4 = it - row * 20;
                        *(f32x4*)(Qx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sq + row * DFF_XH + 4 * c4));
                    }
                    for (int it = lane; it < rows * 16; it += 64) {
                        const int row = it >> 4, c4 = it & 15;
                        *(f32x4*)(Kx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sk + row * 64 + 4 * c4));
                        *(f32x4*)(Vx + row * DFF_XLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sv + row * 64 + 4 * c4));
                    }
                    for (int it = lane; it < rows * 4; it += 64) {
                        const int row = it >> 2, c4 = it & 3;
                        *(f32x4*)(pb + row * DFF_PLD + 4 * c4) = __builtin_nontemporal_load((const f32x4*)(sb + sl.P + ((size_t)h * RA + row) * 16 + 4 * c4));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = lane + 64 * e, row = idx >> 4, cc = idx & 15;
                        const float xv = (cc < 3 && row < rows) ? xs[row * 4 + cc] : 0.f;
                        Kx[row * DFF_XLD + 64 + cc] = xv;
                        Vx[row * DFF_XLD + 64 + cc] = xv;
                    }
                    // G_ext = dattn W_o_ext[h]^T  (5 tiles: [G 64 | r 3 | 0]) -> G region ; dx_i -= r_i
                    wv_wide<KB_H, 1>(afr, lw.WoxT_p, KB_H, h * 5, 5,
                        [&](int, float (&)[1]) {},
                        [&](int t, const f32x4& acc, const float (&)[1]) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = quad * 4 + r;
                                if (row < rows) {
                                    Gx[row * DFF_XLD + 16 * t + col] = acc[r];
                                    if (t == 4 && col < 3) dxw[row * 4 + col] -= acc[r];
                                }
                            }
                        });
                    // dA = G_ext V_ext^T ; dS = scale * P (dA - sum_j P dA)
                    const f32x4 dA = wv_dot_rows(Gx, Vx);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = quad * 4 + r;
                        const float p = pb[i * DFF_PLD + col];
                        const float sm = grp16_sum(p * dA[r]);
                        dsb[i * DFF_PLD + col] = 0.125f * p * (dA[r] - sm);
                    }
                    if (l > 0) {
                        // dV_ext = P^T G_ext -> V region (ext columns: dx term sum_i a_ij r_i)
                        wv_mm<0, 5, true>(pb, Gx, [&](int nt, const f32x4& acc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = quad * 4 + r;
                                if (row < rows) {
                                    Vx[row * DFF_XLD + 16 * nt + col] = acc[r];
                                    if (nt == 4 && col < 3) dxw[row * 4 + col] += acc[r];
                                }
                            }
                        });
                        // dQ_ext = dS K_ext -> G region (ext columns: du)
                        wv_mm<0, 5, false>(dsb, Kx, [&](int nt, const f32x4& acc) { c_store(Gx, DFF_XLD, 16 * nt, acc, rows); });
                        // dK_ext = dS^T Q_ext -> K region (ext columns: dx term sum_i dS_ij u_i)
                        wv_mm<0, 5, true>(dsb, Qx, [&](int nt, const f32x4& acc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = quad * 4 + r;
                                if (row < rows) {
                                    Kx[row * DFF_XLD + 16 * nt + col] = acc[r];
                                    if (nt == 4 && col < 3) dxw[row * 4 + col] += acc[r];
                                }
                            }
                        });
                        // d(LN1 out) partial += [dQ_ext | dK | dV] W_qkv_ext[h]   (K = 80 + 64 + 64)
                        wv_tall<NT_H>(acc_a, 13,
                            [&](int kb) {
                                const float* base = kb < 5 ? Gx + 16 * kb : kb < 9 ? Kx + 16 * (kb - 5) : Vx + 16 * (kb - 9);
                                return base + col * DFF_XLD + 4 * quad;
                            },
                            [&](int kb) { return h * 13 + kb; }, lw.WqkvxT_p, DFF_HEADS * 13);
                    } else {
                        // layer 0: node inputs do not depend on x -> only the x-gradient tiles
                        wv_mm<4, 5, true>(pb, Gx, [&](int, const f32x4& acc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = quad * 4 + r;
                                if (row < rows && col < 3) dxw[row * 4 + col] += acc[r];
                            }
                        });
                        wv_mm<4, 5, true>(dsb, Qx, [&](int, const f32x4& acc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = quad * 4 + r;
                                if (row < rows && col < 3) dxw[row * 4 + col] += acc[r];
                            }
                        });
                    }
                }
                if (l > 0) {
#pragma unroll
                    for (int nt = 0; nt < NT_H; ++nt) c_store(mypart, LH, 16 * nt, acc_a[nt], rows);
                }
            }
            __syncthreads();
            pf.tick(9);
            // ---- row stage F: dn = dn_in partial + LN1 backward(sum_w part)  (l > 0) ----
            if (l > 0) {
                if (ract) {
                    float nin[HC], dyg[HC], xh[HC];
#pragma unroll
                    for (int i = 0; i < HC; ++i) nin[i] = ld_nt(sb + sl.nodes_in + rrow * H + sub + 16 * i);
                    float mean, rstd;
                    ln_stats<H>(nin, mean, rstd);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + 16 * i, o = rrow * LH + cl;
                        xh[i] = (nin[i] - mean) * rstd;
                        dyg[i] = (part[o] + part[16 * LH + o] + part[32 * LH + o] + part[48 * LH + o]) * lw.ln1_g[cl];
                        s1 += dyg[i];
                        s2 += dyg[i] * xh[i];
                    }
                    s1 = grp16_sum(s1) * (1.0f / H);
                    s2 = grp16_sum(s2) * (1.0f / H);
#pragma unroll
                    for (int i = 0; i < HC; ++i) resbuf[rrow * LH + sub + 16 * i] += rstd * (dyg[i] - s1 - xh[i] * s2);
                }
                __syncthreads();
            }
            pf.tick(10);
        }
        // dxs = sum of the 4 waves' partial x-gradients
        if (tid < 64) {
            const float* d0 = smem + LL::dxw;
            dxs[tid] = d0[tid] + d0[64 + tid] + d0[128 + tid] + d0[192 + tid];
        }
        __syncthreads();

        // =============================== update (as dff_fused_kernel) ===============================
        if (a.mode == DFF_MODE_SCORE) {
            if (tid < rows * 4 && (tid & 3) < 3)
                a.force_out[((size_t)b0 * N + (tid >> 2)) * 3 + (tid & 3)] = -dxs[tid];
        } else if (a.mode == DFF_MODE_LANGEVIN) {
            const bool save = ((step + 1) % a.save_interval) == 0;
            const int fi = (step + 1) / a.save_interval - 1;
            if (tid < rows * 4 && (tid & 3) < 3) {
                const int row = tid >> 2, cc = tid & 3;
                const int g = row / N, i = row - g * N;
                const size_t item = (size_t)b0 + g;
                float xi;
                if (a.noise) xi = a.noise[(((size_t)step * a.B + item) * N + i) * 3 + cc];
                else xi = philox_normal(a.seed, a.item_offset + item, a.step_offset + step, i, cc);
                const float f = dxs[tid] * a.force_scale;
                const float x = xst[tid];
                float xn, vn = 0.f;
                if (a.overdamped) {
                    xn = x + f * a.dtau + a.brown_sigma * xi;
                } else {
                    vn = vst[tid] + (a.dt * f) / a.mass[i];
                    xn = x + (vn * a.dt) / 2.0f;
                    const float nz = a.noise_sigma[i] * xi;
                    vn = vn * a.vscale;
                    vn = vn + a.noisescale * nz;
                    xn = xn + (vn * a.dt) / 2.0f;
                }
                xst[tid] = xn;
                vst[tid] = vn;
                if (save && a.frames) a.frames[(((size_t)fi * a.B + item) * N + i) * 3 + cc] = xn;
            }
            __syncthreads();
            if (save && a.ke && !a.overdamped && tid < gcnt) {
                float ke = 0.f;
                for (int i = 0; i < N; ++i) {
                    const float* vp = vst + (tid * N + i) * 4;
                    ke += a.mass[i] * (vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]);
                }
                a.ke[(size_t)fi * a.B + b0 + tid] = 0.5f * ke;
            }
        } else {
            const bool act = tid < rows * 4 && (tid & 3) < 3;
            const int row = tid >> 2, cc = tid & 3;
            const int g = act ? row / N : 0, i = act ? row - g * N : 0;
            const size_t item = (size_t)b0 + g;
            float eps = act ? -dxs[tid] : 0.f;
            float xi = 0.f;
            if (act) {
                if (a.noise) xi = a.noise[(((size_t)step * a.B + item) * N + i) * 3 + cc];
                else xi = philox_normal(a.seed, a.item_offset + item, (uint64_t)t_int, i, cc);
            }
            if (tid < rows * 4) { dxs[tid] = eps; xs[tid] = xi; }
            __syncthreads();
            bead_mean(c, dxs, cm);
            bead_mean(c, xs, cm + 64);
            __syncthreads();
            const float x = act ? xst[tid] : 0.f;
            float x0 = 0.f;
            if (act) {
                eps -= cm[g * 4 + cc];
                xi -= cm[64 + g * 4 + cc];
                x0 = m.sqrt_recip_ac[t_int] * x - m.sqrt_recipm1_ac[t_int] * eps;
            }
            __syncthreads();
            if (tid < rows * 4) dxs[tid] = x0;
            __syncthreads();
            bead_mean(c, dxs, cm);
            __syncthreads();
            float xn = 0.f;
            if (act) {
                x0 -= cm[g * 4 + cc];
                const float mean = m.post_c1[t_int] * x0 + m.post_c2[t_int] * x;
                const float nzm = (t_int == 0) ? 0.f : 1.f;
                xn = mean + nzm * expf(0.5f * m.post_logvar[t_int]) * xi;
                if (xn > 1000.f || xn < -1000.f) {
                    if (a.clamp_flag) *a.clamp_flag = 1;
                    xn = fminf(fmaxf(xn, -1000.f), 1000.f);
                }
            }
            __syncthreads();
            if (tid < rows * 4) dxs[tid] = xn;
            __syncthreads();
            bead_mean(c, dxs, cm);
            __syncthreads();
            if (act) xst[tid] = xn - cm[g * 4 + cc];
        }
        __syncthreads();
        pf.tick(11);
    }
    if (pf.on)
        for (int i = 0; i < DFF_NPROF; ++i) a.prof[i] = pf.acc[i];
    if (a.mode != DFF_MODE_SCORE && tid < rows * 4 && (tid & 3) < 3) {
        const size_t gi = ((size_t)b0 * N + (tid >> 2)) * 3 + (tid & 3);
        a.x_io[gi] = xst[tid];
        if (a.mode == DFF_MODE_LANGEVIN && !a.overdamped) a.v_io[gi] = vst[tid];
    }
}

template __global__ void dff_small_kernel<64>(const DffModelDev, const DffRunArgs);
template __global__ void dff_small_kernel<96>(const DffModelDev, const DffRunArgs);
template __global__ void dff_small_kernel<128>(const DffModelDev, const DffRunArgs);
