// dff_kernels.hip -- the MI355X (gfx950 / CDNA4) device code of the denoising-force-field sampler.
// (build.sh compiles this translation unit with -mllvm -disable-machine-licm: see there.)
//
// ONE persistent kernel runs the whole hot path for a group of proteins per workgroup:
//
//   for step in 0 .. n_steps-1:
//       x <- center(x)                                      utils.py:65-70
//       E  = energy(x, t)         forward, all layers        models/graph_transformer.py:90-108
//       dx = d(sum E)/dx          hand-written VJP           models/graph_transformer.py:143-159
//       x,v <- BAOAB / Brownian / DDPM p_sample update       dynamics/langevin_cgnet.py:447-500,
//                                                            models/ddpm.py:195-232,248-251
//
// Trajectories / samples are independent, so there is no inter-workgroup communication at all:
// a workgroup owns G proteins (rows = G*N <= 16*MT bead rows) for the entire launch, keeps x, v
// and every activation of the current layer in LDS, streams the (pre-packed) weights from L2
// straight into MFMA B-operand registers, and parks what the backward pass needs (q, k, v,
// attention probabilities, gate inputs, FFN pre-activations) in a per-workgroup global "stash"
// that never leaves L2 / Infinity Cache.
//
// Formulation: the reference materialises e_kv = edges_to_kv(edge_embedding(x_j - x_i)) as a
// (B*8, N, N, 64) tensor (graph_transformer.py:96,235-245).  There is no nonlinearity between
// the two Linear layers, so it is folded away exactly (see oracle/kernel_model.py and
// SURVEY.md section 0.3): x enters only through  u_ih . x_j  in the logits and
// xrel_ih = sum_j a_ihj x_j - x_i  in the values, with u = LN1(nodes) W_u^T + b_u and the xrel
// term routed through W_oc = W_o W_c.  All dense projections (q|k|v, u, W_o, W_oc, FFN and
// their transposes for the VJP) run on v_mfma_f32_16x16x4_f32 (exact fp32 == fmaf chain);
// softmax / LayerNorm / GELU / gates / the N x N contractions run on the VALU out of LDS.
#include "dff_device.h"
// DFF_F16G (round 5): which weight GEMMs of the split variants run on the two-piece fp16 format (4 B per weight, three MFMAs per
// unit and row tile; dff_device.h split8h) instead of the three-piece bf16 one: bit 0 = the forward images (Wqkvx, Wox, W1, W2;
// their A operands are O(1) activations: no scaling), bit 1 = the FFN backward (W2T, W1T; row-scaled), bit 2 = G_ext (WoxT).
#ifndef DFF_F16G
#define DFF_F16G 15   // ... bit 3 = the QKV_ext^T back-projection (WqkvxT; dQ / dK / dV scaled by one power of two per workgroup)
#endif
#define DFF_QT16 ((DFF_F16G & 12) == 12)   // (the block scale is derived from the row scales of G_ext's input: bit 2 as well)
#include <type_traits>
#ifndef DFF_AUXLATE
#define DFF_AUXLATE 1   // wide split GEMMs: the tiles' auxiliary rows are requested behind the ring's first entries (protein G -1.1 %, trp-cage -0.5 %, villin / BBA -0.2 %)
#endif
#ifndef DFF_APRE
#define DFF_APRE 1
#endif
#ifndef DFF_PIPEB_128_2
#define DFF_PIPEB_128_2 1   // trp-cage's shape in the backward head pipeline (half-unit ring entries)
#endif
#ifndef DFF_QTPRE
#define DFF_QTPRE 1
#endif
#ifndef DFF_TPRE_MT
#define DFF_TPRE_MT 3   // tall split GEMMs: A fragments one k-block ahead, up to this many row tiles (four: measured neutral)
#endif
#ifndef DFF_K2_MT
#define DFF_K2_MT 3    // two output tiles per wave in the wide split GEMMs from this many row tiles on (three: -0.3 % with the fp16 engine; it spilled with three pieces)
#endif
#ifndef DFF_ARES_LIM
#define DFF_ARES_LIM 8 // A operand register-resident in the wide split GEMMs while MT * KB32 <= this
#endif
#ifndef DFF_ARES
#define DFF_ARES 1
#endif
// The image formats of the host (dff_fused_f16_mask) and the kernels are coupled through DFF_F16G: the QKV_ext^T image is read as
// fp16 pieces only when the G_ext group is too (DFF_QT16) -- a build with bit 3 but not bit 2 would multiply a two-piece image as three
static_assert((DFF_F16G & 8) == 0 || (DFF_F16G & 4) != 0, "DFF_F16G: bit 3 (QKV_ext^T) needs bit 2 (G_ext)");

// ------------------------------------------------------------------------------------------
// MFMA GEMM stages.  A (rows x K) lives in LDS with leading dimension lda (multiple of 4);
// B is the packed weight image in global memory (see dff_internal.h).  The products are issued TRANSPOSED (the weight
// fragment as the MFMA's first operand, the activation fragment as its second), so that of an output tile lane l holds
// ROW (l & 15) and the four consecutive COLUMNS 4*(l >> 4) + r, r = 0..3: every epilogue then moves 16 bytes per lane
// (one LDS write, one stash store) instead of four scattered dwords.
// ------------------------------------------------------------------------------------------
DEVI void mfma4(f32x4& acc, const f32x4 a, const f32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// Weight fetch latency out of L2 is ~1 us when all 256 CUs stream the same packed image, and a
// workgroup has only 8 waves, so each wave keeps a ring of D tiles (wide) / k-blocks (tall) of
// B operands in flight in registers (Little: D tiles * KB KiB * 8 waves per CU).

// "wide" GEMM: K = 16*KB (small, compile time), many output tiles; the waves take tiles
// round-robin.  epi(nt_local, mt, acc, aux) consumes one 16x16 output tile.
// pre(nt_local, aux) issues the global loads the tile's epilogue needs (bias, stashed values)
// together with the tile's weights, so their latency is hidden by the ring as well.
// Register diet (two waves per SIMD share 512 registers): the ring holds D tiles with D * KB ~ 12-16
// k-blocks in flight, and the A fragments are re-read from LDS for every tile unless they are few.
// hook() runs right after the first ring loads are issued: the place to start long-latency loads (stash rows out of
// HBM) that must NOT be older than the weight loads -- vmcnt retires in order, so a wave that issued them first would
// wait for the HBM round trip before its first MFMA.
struct NoHook { DEVI void operator()() const {} };
// NTN_MAX > 0: the caller guarantees ntn <= NTN_MAX; when that fits the ring (no refills) the tile loop is
// unrolled away, which also lets the compiler count the loads in flight exactly (a loop with refills makes it wait
// for everything younger, hook loads included).
template <int MT, int KB, int NAUX, int NTN_MAX = 0, class Pre, class Epi, class Hook = NoHook>
DEVI void gemm_wide(const lfloat* A, int lda, int rowsA, const float* __restrict__ Wp, int KBtot,
                    int kb0, int nt0, int ntn, Pre pre, Epi epi, Hook hook = Hook()) {
    const int tid_ = tid_now();
    constexpr int D = KB >= 6 ? 2 : 3;
    constexpr bool HOLD = MT * KB <= 16;   // keep all A fragments in registers
    constexpr int NC = MT == 1 ? 2 : 1;    // independent accumulator chains per output tile
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kk = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * lda + 4 * kk;
    f32x4 ah[HOLD ? MT : 1][HOLD ? KB : 1];
    if (HOLD) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) ah[HOLD ? mt : 0][HOLD ? kb : 0] = *(const lf32x4*)(A + rowoff[mt] + 16 * kb);
    }
    const WPtr<gf32x4, DFF_WMODE(MT)> wp((const gf32x4*)Wp + (size_t)kb0 * 64, (unsigned)lane & 63u);
    const int cnt = wave < ntn ? (ntn - wave + DFF_NWAVES - 1) / DFF_NWAVES : 0;
    f32x4 b[D][KB];
    float aux[D][NAUX];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < cnt) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b[d][kb] = wp[((size_t)(nt0 + wave + DFF_NWAVES * d) * KBtot + kb) * 64];
            pre(wave + DFF_NWAVES * d, aux[d]);
        }
    hook();
    constexpr bool ONCE = NTN_MAX > 0 && NTN_MAX <= D * DFF_NWAVES;
    for (int i0 = 0; i0 < (ONCE ? 1 : cnt); i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < cnt) {
                const int nt = wave + DFF_NWAVES * i;
                f32x4 acc[MT][NC];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < NC; ++q) acc[mt][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    f32x4 a[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[mt] = HOLD ? ah[HOLD ? mt : 0][HOLD ? kb : 0] : *(const lf32x4*)(A + rowoff[mt] + 16 * kb);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][kb % NC] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[d][kb][s4], a[mt][s4], acc[mt][kb % NC], 0, 0, 0);
                }
                float auxc[NAUX];
#pragma unroll
                for (int q = 0; q < NAUX; ++q) auxc[q] = aux[d][q];
                if (!ONCE && i + D < cnt) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        b[d][kb] = wp[((size_t)(nt0 + nt + DFF_NWAVES * D) * KBtot + kb) * 64];
                    pre(nt + DFF_NWAVES * D, aux[d]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) epi(nt, mt, NC == 2 ? acc[mt][0] + acc[mt][NC - 1] : acc[mt][0], auxc, true, 0);
            }
        }
    }
}

// gemm_wide, loop-free (see gemm_wide_split_st below for why): NTN tiles over the 8 waves, ceil(NTN / 8) rounds, ring of
// half tiles (KB / 2 k-blocks), a surplus tile in the last round repeats tile NTN - 1 with valid = false, and pre / epi
// issue the same VMEM operations for every tile.
template <int MT, int KB, int NTN, int NAUX, class Pre, class Epi>
DEVI void gemm_wide_st(const lfloat* A, int lda, int rowsA, const float* __restrict__ Wp, int nt0, Pre pre, Epi epi) {
    static_assert(KB % 2 == 0, "even number of k-blocks");
    constexpr int HB = KB / 2, CNT = (NTN + DFF_NWAVES - 1) / DFF_NWAVES, NE = 2 * CNT, DR = NE < 3 ? NE : 3, NA = 3;
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kk = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * lda + 4 * kk;
    const WPtr<gf32x4, DFF_WMODE(MT)> wp((const gf32x4*)Wp, (unsigned)lane & 63u);
    f32x4 b[DR][HB];
    float aux[NA][NAUX];
    auto tile_of = [&](int i) { return min(wave + DFF_NWAVES * i, NTN - 1); };
    auto fill = [&](f32x4 (&slot)[HB], int e) {
        const size_t tile = (size_t)(nt0 + tile_of(e / 2));
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) slot[kb] = wp[(tile * KB + (e % 2) * HB + kb) * 64];
    };
#pragma unroll
    for (int j = 0; j < DR; ++j) {
        fill(b[j], j);
        if (j % 2 == 0) pre(tile_of(j / 2), aux[(j / 2) % NA]);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[MT];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = e / 2, half = e % 2, slot = e % DR;
        if (half == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (NTN % DFF_NWAVES == 0 || i < CNT - 1 || wave + DFF_NWAVES * i < NTN) {
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
                f32x4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *(const lf32x4*)(A + rowoff[mt] + 16 * (half * HB + kb));
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[slot][kb][s4], a[mt][s4], acc[mt], 0, 0, 0);
            }
        }
        if (e + DR < NE) {
            fill(b[slot], e + DR);
            if ((e + DR) % 2 == 0) pre(tile_of((e + DR) / 2), aux[((e + DR) / 2) % NA]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (half == 1) {
            const bool valid = wave + DFF_NWAVES * i < NTN;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) epi(tile_of(i), mt, acc[mt], aux[i % NA], valid, i);
        }
    }
}

// gemm_wide on the split operands (Wp = pack_b_split image: [tile][k32-block][piece][lane] x 8 bf16; as = the piece arrays the
// row stages write, rstore_a), loop-free: NTN is a compile-time tile count, every wave runs ceil(NTN / 8) tiles (a wave without a tile
// of its own in the last round repeats tile NTN - 1 with valid = false: it would idle at the barrier anyway), and
// pre / epi issue the SAME global loads and stores for every tile (epilogues redirect what must not be stored to the
// stash's junk slot).  With no control flow around VMEM the compiler counts the operations in flight exactly;
// in the looped version every `if` around a store made it wait for one more YOUNGER load, i.e. the ring drained
// (s_waitcnt vmcnt(0) in front of every MFMA block).  epi(nt, mt, acc, aux, valid).
// Waves W0 .. W0 + NWV - 1 share the tiles (the others must not call); epi also gets the round index i (compile time
// after unrolling: an epilogue may park the tile in registers, see the head loop of the forward pass).
// F16 (round 5): the two-piece fp16 format (dff_device.h split8h) -- image [tile][k-block][h | l'][lane], A pieces as[0] = h, as[1] = l',
// three v_mfma_f32_16x16x32_f16 per (unit, row tile): cb = h.h, cs = the two 2^11-scaled cross terms, result cb + cs / 2048.
template <int MT, int KB32, int NTN, int NAUX, int W0 = 0, int NWV = DFF_NWAVES, int DRMAX = 3, bool F16 = false, class Pre, class Epi>
DEVI void gemm_wide_split_st(const lu32* as, int R, int rowsA, const unsigned* __restrict__ Wp, int nt0, Pre pre, Epi epi) {
    constexpr int NP = F16 ? 2 : 3;
    constexpr bool HALVES = KB32 % 2 == 0 && KB32 >= 4;
    constexpr int NHALF = HALVES ? 2 : 1, HB = KB32 / NHALF, LHS2 = (32 * KB32 + DFF_SPAD) / 2;
    constexpr int CNT = (NTN + NWV - 1) / NWV, NE = CNT * NHALF;
    // ARES (two row tiles at most): the whole split A operand -- 12 MT KB32 registers -- is read from LDS ONCE and stays in
    // registers while the wave's tiles stream by.  Otherwise every wave re-reads all of A for every tile: at MT = 2 that is
    // as many LDS cycles as the GEMM has MFMA cycles, and the two do not overlap (trp-cage's QKV_ext GEMM: 14.5 k cycles per
    // call against 5 k of products).
    constexpr bool ARES = MT * KB32 <= DFF_ARES_LIM && DFF_ARES;
    constexpr int DR0 = HALVES ? (DRMAX < 3 ? DRMAX : 3) : 2, DR = NE < DR0 ? NE : DR0;
    constexpr int NA = 3;
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6) - W0;
    const int kg = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * LHS2 + 4 * kg;
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
    u32x4 b[DR][HB][3];
    float aux[NA][NAUX];
    auto tile_of = [&](int i) { return min(wave + NWV * i, NTN - 1); };
    auto fill = [&](u32x4 (&slot)[HB][3], int e) {
        const size_t tile = (size_t)(nt0 + tile_of(e / NHALF));
#pragma unroll
        for (int kb = 0; kb < HB; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p) slot[kb][p] = wp[((tile * KB32 + (e % NHALF) * HB + kb) * NP + p) * 64];
    };
    // DFF_AUXLATE: the ring's first entries are requested BEFORE the tiles' auxiliary rows.  Loads return in order, and an
    // auxiliary row may come from HBM (the backward's gelu' rows out of the stash): requested in between, every ring entry
    // behind it waits a memory latency the first entries' products could have covered.
#pragma unroll
    for (int j = 0; j < DR; ++j) {
        fill(b[j], j);
        if constexpr (!DFF_AUXLATE) if (j % NHALF == 0) pre(tile_of(j / NHALF), aux[(j / NHALF) % NA]);
    }
    if constexpr (DFF_AUXLATE) {
#pragma unroll
        for (int j = 0; j < DR; ++j)
            if (j % NHALF == 0) pre(tile_of(j / NHALF), aux[(j / NHALF) % NA]);
    }
    // the loads above are issued HERE: left alone, the scheduler sinks each next to its first use (one L2 latency
    // per k-block instead of one per GEMM)
    __builtin_amdgcn_sched_barrier(0);
    u32x4 ares[ARES ? MT : 1][ARES ? KB32 : 1][3];
    if constexpr (ARES) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kb = 0; kb < KB32; ++kb) {
                const int o = rowoff[mt] + 16 * kb;
                ares[mt][kb][0] = *(const lu32x4*)(as + o);
                ares[mt][kb][1] = *(const lu32x4*)(as + R * LHS2 + o);
                if constexpr (!F16) ares[mt][kb][2] = *(const lu32x4*)(as + 2 * R * LHS2 + o);
            }
    }
    constexpr bool APRE = !ARES && DFF_APRE;
    u32x4 apre[3];
    if constexpr (APRE) {
        apre[0] = *(const volatile lu32x4*)(as + rowoff[0]);
        apre[1] = *(const volatile lu32x4*)(as + R * LHS2 + rowoff[0]);
        if constexpr (!F16) apre[2] = *(const volatile lu32x4*)(as + 2 * R * LHS2 + rowoff[0]);
    }
    f32x4 cs[MT], cb[MT];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = e / NHALF, half = e % NHALF, slot = e % DR;
        if (half == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { cs[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; cb[mt] = cs[mt]; }
        }
        // a surplus tile (last round only) skips its products: a wave-uniform branch with no VMEM inside, so the
        // counts stay exact, and the wave gets out of the way of its SIMD partner
        if (NTN % NWV == 0 || i < CNT - 1 || wave + NWV * i < NTN) {
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int o = rowoff[mt] + 16 * (half * HB + kb);
                    u32x4 ah, am, al;
                    if constexpr (ARES) {
                        ah = ares[mt][half * HB + kb][0]; am = ares[mt][half * HB + kb][1]; al = ares[mt][half * HB + kb][F16 ? 1 : 2];
                    } else if constexpr (APRE) {
                        // this unit's fragments were requested a unit ago; request the next unit's (the first of the next
                        // entry after the last of this one: A is the same for every tile) before this unit's products
                        ah = apre[0]; am = apre[1]; al = apre[F16 ? 1 : 2];
                        constexpr int dummy = 0; (void)dummy;
                        const int mtn = (mt + 1) % MT, kbn = (mt + 1 == MT) ? kb + 1 : kb;
                        const int kabs = (kbn == HB) ? ((half + 1) % NHALF) * HB : half * HB + kbn;
                        const int on = rowoff[mtn] + 16 * kabs;
                        apre[0] = *(const volatile lu32x4*)(as + on);
                        apre[1] = *(const volatile lu32x4*)(as + R * LHS2 + on);
                        if constexpr (!F16) apre[2] = *(const volatile lu32x4*)(as + 2 * R * LHS2 + on);
                    } else {
                        ah = *(const lu32x4*)(as + o);
                        am = *(const lu32x4*)(as + R * LHS2 + o);
                        if constexpr (!F16) al = *(const lu32x4*)(as + 2 * R * LHS2 + o);
                    }
                    if constexpr (F16) {
                        cs[mt] = mfma_f16(b[slot][kb][0], am, cs[mt]);
                        cs[mt] = mfma_f16(b[slot][kb][1], ah, cs[mt]);
                        cb[mt] = mfma_f16(b[slot][kb][0], ah, cb[mt]);
                    } else {
                    cs[mt] = mfma_bf16(b[slot][kb][0], al, cs[mt]);
                    cb[mt] = mfma_bf16(b[slot][kb][0], am, cb[mt]);
                    cs[mt] = mfma_bf16(b[slot][kb][2], ah, cs[mt]);
                    cb[mt] = mfma_bf16(b[slot][kb][1], ah, cb[mt]);
                    cs[mt] = mfma_bf16(b[slot][kb][1], am, cs[mt]);
                    cb[mt] = mfma_bf16(b[slot][kb][0], ah, cb[mt]);
                    }
                }
            }
        }
        if (e + DR < NE) {
            fill(b[slot], e + DR);
            if ((e + DR) % NHALF == 0) pre(tile_of((e + DR) / NHALF), aux[((e + DR) / NHALF) % NA]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (half == NHALF - 1) {
            const bool valid = wave + NWV * i < NTN;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) epi(tile_of(i), mt, F16 ? cb[mt] + cs[mt] * DFF_F16_LINV : cb[mt] + cs[mt], aux[i % NA], valid, i);
        }
    }
}

// The same GEMM with TWO output tiles per wave side by side (k-block-major): an A fragment read from LDS feeds both tiles'
// products.  With three or four row tiles the A operand does not fit the register file (ARES), every tile re-reads all of
// it, and three ds_read_b128 per six 16-cycle products make the LDS as busy as the matrix pipes (8 waves x 24 LDS cycles
// against 2 waves x 96 pipe cycles per SIMD): the two limits add up instead of overlapping.  Pairs q = wave + 8 i cover
// tiles 2 q and 2 q + 1 (a surplus tile repeats NTN - 1 with valid = false); ring of two k-blocks (both tiles' pieces).
template <int MT, int KB32, int NTN, int NAUX, bool F16 = false, class Pre, class Epi>
DEVI void gemm_wide_split_k2(const lu32* as, int R, int rowsA, const unsigned* __restrict__ Wp, int nt0, Pre pre, Epi epi) {
    constexpr int NPC = F16 ? 2 : 3;   // pieces per weight
    constexpr int LHS2 = (32 * KB32 + DFF_SPAD) / 2, NP = (NTN + 1) / 2, CNT = (NP + DFF_NWAVES - 1) / DFF_NWAVES, DR = 2;
    static_assert(KB32 >= DR, "ring");
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kg = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * LHS2 + 4 * kg;
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int q = wave + DFF_NWAVES * i;
        const int t0 = min(2 * q, NTN - 1), t1 = min(2 * q + 1, NTN - 1);
        const bool v0 = 2 * q < NTN, v1 = 2 * q + 1 < NTN;
        u32x4 b[DR][2][3];
        float aux[2][NAUX];
        auto fill = [&](u32x4 (&slot)[2][3], int kb) {
#pragma unroll
            for (int p = 0; p < NPC; ++p) slot[0][p] = wp[(((size_t)(nt0 + t0) * KB32 + kb) * NPC + p) * 64];
#pragma unroll
            for (int p = 0; p < NPC; ++p) slot[1][p] = wp[(((size_t)(nt0 + t1) * KB32 + kb) * NPC + p) * 64];
        };
#pragma unroll
        for (int d = 0; d < DR; ++d) fill(b[d], d);
        if constexpr (!(DFF_AUXLATE && KB32 > DR)) {
            pre(t0, aux[0]);
            pre(t1, aux[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 cs[2][MT], cb[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { cs[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; cb[t][mt] = cs[t][mt]; }
        u32x4 apre[3];
        apre[0] = *(const volatile lu32x4*)(as + rowoff[0]);
        apre[1] = *(const volatile lu32x4*)(as + R * LHS2 + rowoff[0]);
        if constexpr (!F16) apre[2] = *(const volatile lu32x4*)(as + 2 * R * LHS2 + rowoff[0]);
        if (v0) {   // (wave-uniform; a wave without a pair skips the products)
#pragma unroll
            for (int kb = 0; kb < KB32; ++kb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x4 ah = apre[0], am = apre[1], al = apre[F16 ? 1 : 2];
                    const int mtn = (mt + 1) % MT, kbn = (mt + 1 == MT) ? (kb + 1) % KB32 : kb;
                    const int on = rowoff[mtn] + 16 * kbn;
                    apre[0] = *(const volatile lu32x4*)(as + on);
                    apre[1] = *(const volatile lu32x4*)(as + R * LHS2 + on);
                    if constexpr (!F16) apre[2] = *(const volatile lu32x4*)(as + 2 * R * LHS2 + on);
                    if constexpr (F16) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) cs[t][mt] = mfma_f16(b[kb % DR][t][0], am, cs[t][mt]);
#pragma unroll
                        for (int t = 0; t < 2; ++t) cs[t][mt] = mfma_f16(b[kb % DR][t][1], ah, cs[t][mt]);
#pragma unroll
                        for (int t = 0; t < 2; ++t) cb[t][mt] = mfma_f16(b[kb % DR][t][0], ah, cb[t][mt]);
                    } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cs[t][mt] = mfma_bf16(b[kb % DR][t][0], al, cs[t][mt]);
                        cb[t][mt] = mfma_bf16(b[kb % DR][t][0], am, cb[t][mt]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cs[t][mt] = mfma_bf16(b[kb % DR][t][2], ah, cs[t][mt]);
                        cb[t][mt] = mfma_bf16(b[kb % DR][t][1], ah, cb[t][mt]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cs[t][mt] = mfma_bf16(b[kb % DR][t][1], am, cs[t][mt]);
                        cb[t][mt] = mfma_bf16(b[kb % DR][t][0], ah, cb[t][mt]);
                    }
                    }
                }
                if (kb + DR < KB32) {
                    fill(b[kb % DR], kb + DR);
                    if constexpr (DFF_AUXLATE) if (kb == 0) { pre(t0, aux[0]); pre(t1, aux[1]); }   // (behind three k-blocks, see gemm_wide_split_st)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (DFF_AUXLATE && KB32 > DR) {
            pre(t0, aux[0]);
            pre(t1, aux[1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) epi(t0, mt, F16 ? cb[0][mt] + cs[0][mt] * DFF_F16_LINV : cb[0][mt] + cs[0][mt], aux[0], v0, i);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) epi(t1, mt, F16 ? cb[1][mt] + cs[1][mt] * DFF_F16_LINV : cb[1][mt] + cs[1][mt], aux[1], v1, i);
    }
}

// gemm_wide_units on the split operands (few output tiles: (tile, row-tile) units round-robin over the waves, loop-free).
// Up to two units per wave: all weights requested up front.  More (four row tiles: 5 tiles x 4 = 20 units, three per
// wave): a ring of two units -- three units of K = 128 weights at once are 144 registers.
template <int MT, int KB32, int NTN, bool F16 = false, class Epi>
DEVI void gemm_wide_units_split(const lu32* as, int R, int rowsA, const unsigned* __restrict__ Wp, int nt0, Epi epi) {
    constexpr int NPC = F16 ? 2 : 3;
    const int tid_ = tid_now();
    constexpr int NU = NTN * MT, DU = (NU + DFF_NWAVES - 1) / DFF_NWAVES, DRU = DU < 2 ? DU : 2, LHS2 = (32 * KB32 + DFF_SPAD) / 2;
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kg = lane >> 4, mm = lane & 15;
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
    u32x4 b[DRU][KB32][3];
    auto fill = [&](u32x4 (&slot)[KB32][3], int d) {
        const int nt = min(wave + DFF_NWAVES * d, NU - 1) / MT;
#pragma unroll
        for (int kb = 0; kb < KB32; ++kb)
#pragma unroll
            for (int p = 0; p < NPC; ++p) slot[kb][p] = wp[(((size_t)(nt0 + nt) * KB32 + kb) * NPC + p) * 64];
    };
#pragma unroll
    for (int d = 0; d < DRU; ++d) fill(b[d], d);
    if constexpr (DU > DRU) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DU; ++d) {
        const int u = wave + DFF_NWAVES * d;
        if (u < NU) {
            const int nt = u / MT, mt = u - nt * MT;
            const int ro = min(mt * 16 + mm, rowsA - 1) * LHS2 + 4 * kg;
            f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f}, cs2 = cs, cb2 = cs;
#pragma unroll
            for (int kb = 0; kb < KB32; ++kb) {
                const u32x4 ah = *(const lu32x4*)(as + ro + 16 * kb);
                const u32x4 am = *(const lu32x4*)(as + R * LHS2 + ro + 16 * kb);
                if constexpr (F16) {   // cs / cs2: the 2^11-scaled cross terms, cb: h.h
                    cs = mfma_f16(b[d % DRU][kb][0], am, cs);
                    cs2 = mfma_f16(b[d % DRU][kb][1], ah, cs2);
                    cb = mfma_f16(b[d % DRU][kb][0], ah, cb);
                } else {
                const u32x4 al = *(const lu32x4*)(as + 2 * R * LHS2 + ro + 16 * kb);
                cs = mfma_bf16(b[d % DRU][kb][0], al, cs);
                cb = mfma_bf16(b[d % DRU][kb][0], am, cb);
                cs2 = mfma_bf16(b[d % DRU][kb][2], ah, cs2);
                cb2 = mfma_bf16(b[d % DRU][kb][1], ah, cb2);
                cs = mfma_bf16(b[d % DRU][kb][1], am, cs);
                cb = mfma_bf16(b[d % DRU][kb][0], ah, cb);
                }
            }
            if constexpr (DU > DRU) {
                if (d + DRU < DU) { fill(b[d % DRU], d + DRU); __builtin_amdgcn_sched_barrier(0); }
            }
            if constexpr (F16) epi(nt, mt, cb + (cs + cs2) * DFF_F16_LINV);
            else epi(nt, mt, (cb + cb2) + (cs + cs2));
        }
    }
}

// The (tile, row-tile) units of a head group's G_ext GEMM (NT = 5 HGS column tiles x MT row tiles) on waves W0 .. W0 + NWV - 1 only,
// results parked in registers (backward head pipeline: the other waves are busy with dS meanwhile).  Unit u = w + NWV d.
template <int MT, int KB32, int NT, int W0, int NWV, bool F16 = false>
DEVI void gx_units_hold(const lu32* as, int R, int rowsA, const unsigned* __restrict__ Wp, int nt0,
                        f32x4 (&held)[(NT * MT + NWV - 1) / NWV]) {
    constexpr int NPC = F16 ? 2 : 3;
    constexpr int NU = NT * MT, DU = (NU + NWV - 1) / NWV, LHS2 = (32 * KB32 + DFF_SPAD) / 2;
    // ring of two entries; at K = 128 an entry is HALF a unit (two k-blocks, 24 registers): whole units -- 96 registers in
    // flight -- were what made trp-cage's shape spill with this pipeline
    constexpr int NHALF = (KB32 % 2 == 0 && KB32 >= 4) ? 2 : 1, KH = KB32 / NHALF, NE = DU * NHALF;
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6) - W0;
    const int kg = lane >> 4, mm = lane & 15;
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
    u32x4 b[2][KH][3];
    auto fill = [&](u32x4 (&slot)[KH][3], int e) {
        const int nt = min(wave + NWV * (e / NHALF), NU - 1) / MT;
#pragma unroll
        for (int kb = 0; kb < KH; ++kb)
#pragma unroll
            for (int p = 0; p < NPC; ++p) slot[kb][p] = wp[(((size_t)(nt0 + nt) * KB32 + (e % NHALF) * KH + kb) * NPC + p) * 64];
    };
    fill(b[0], 0);
    if (NE > 1) fill(b[1], 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 cs, cb, cs2, cb2;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int d = e / NHALF, h = e % NHALF;
        const int u = min(wave + NWV * d, NU - 1);
        const int mt = u - (u / MT) * MT;
        const int ro = min(mt * 16 + mm, rowsA - 1) * LHS2 + 4 * kg;
        if (h == 0) { cs = (f32x4){0.f, 0.f, 0.f, 0.f}; cb = cs; cs2 = cs; cb2 = cs; }
#pragma unroll
        for (int kb = 0; kb < KH; ++kb) {
            const int ka = h * KH + kb;
            const u32x4 ah = *(const lu32x4*)(as + ro + 16 * ka);
            const u32x4 am = *(const lu32x4*)(as + R * LHS2 + ro + 16 * ka);
            if constexpr (F16) {
                cs = mfma_f16(b[e % 2][kb][0], am, cs);
                cs2 = mfma_f16(b[e % 2][kb][1], ah, cs2);
                cb = mfma_f16(b[e % 2][kb][0], ah, cb);
            } else {
            const u32x4 al = *(const lu32x4*)(as + 2 * R * LHS2 + ro + 16 * ka);
            cs = mfma_bf16(b[e % 2][kb][0], al, cs);
            cb = mfma_bf16(b[e % 2][kb][0], am, cb);
            cs2 = mfma_bf16(b[e % 2][kb][2], ah, cs2);
            cb2 = mfma_bf16(b[e % 2][kb][1], ah, cb2);
            cs = mfma_bf16(b[e % 2][kb][1], am, cs);
            cb = mfma_bf16(b[e % 2][kb][0], ah, cb);
            }
        }
        if (e + 2 < NE) { fill(b[e % 2], e + 2); __builtin_amdgcn_sched_barrier(0); }
        if (h == NHALF - 1) held[d] = F16 ? cb + (cs + cs2) * DFF_F16_LINV : (cb + cb2) + (cs + cs2);
    }
}

// One element of a split A operand: three 16-bit stores (row-major bf16 pieces [piece][R][LS], LS in bf16 units).
template <bool F16 = false>
DEVI void store_split(lu16* as16, int R, int LS, int row, int col, float v) {
    if constexpr (F16) {   // two fp16 pieces (h, l'): dff_device.h split1h
        unsigned short hh, ll;
        split1h(v, hh, ll);
        as16[(0 * R + row) * LS + col] = hh;
        as16[(1 * R + row) * LS + col] = ll;
        return;
    }
    const unsigned uh = __float_as_uint(v) & 0xffff0000u;
    const float r = v - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(um);
    as16[(0 * R + row) * LS + col] = (unsigned short)(uh >> 16);
    as16[(1 * R + row) * LS + col] = (unsigned short)(um >> 16);
    as16[(2 * R + row) * LS + col] = (unsigned short)(__float_as_uint(r2) >> 16);
}
// Four consecutive columns (col % 4 == 0) of one row: one 8-byte store per piece (LS2 = dwords per piece row, even).
template <bool F16 = false>
DEVI void store_split4(lu32* as, int R, int LS2, int row, int col, const f32x4 v) {
    if constexpr (F16) {
        unsigned h0, l0, h1, l1;
        split2h(v[0], v[1], h0, l0);
        split2h(v[2], v[3], h1, l1);
        const int o = row * LS2 + (col >> 1);
        *(lu32x2*)(as + 0 * R * LS2 + o) = (u32x2){h0, h1};
        *(lu32x2*)(as + 1 * R * LS2 + o) = (u32x2){l0, l1};
        return;
    }
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned uh = __float_as_uint(v[q]) & 0xffff0000u;
        const float r = v[q] - __uint_as_float(uh);
        const unsigned um = __float_as_uint(r) & 0xffff0000u;
        const float r2 = r - __uint_as_float(um);
        hh[q] = uh; mm[q] = um; ll[q] = __float_as_uint(r2);
    }
    const int o = row * LS2 + (col >> 1);
    *(lu32x2*)(as + 0 * R * LS2 + o) = (u32x2){__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u)};
    *(lu32x2*)(as + 1 * R * LS2 + o) = (u32x2){__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u)};
    *(lu32x2*)(as + 2 * R * LS2 + o) = (u32x2){__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u)};
}
// The same with a compile-time k-block count: loop-free, so that the ring (D k-blocks ahead) is waited for exactly.
// (PRE: the ring's first D k-blocks were requested by tall_ring_fill before the barrier in front of this GEMM)
template <int NTW, int D, int MT = 1, bool F16 = false>
DEVI void tall_ring_fill(u32x4 (&b)[D][NTW][3], const unsigned* __restrict__ Wp, int KBtot, int kb0, int ntiles) {
    constexpr int NP = F16 ? 2 : 3;
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + DFF_NWAVES * i;
        const size_t tb = ((size_t)(nt < ntiles ? nt : 0) * KBtot + kb0) * NP;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int p = 0; p < NP; ++p) b[d][i][p] = wp[(tb + NP * d + p) * 64];
    }
    asm volatile("" ::: "memory");
}
// F16: two-piece fp16 operands (see gemm_wide_split_st); the 2^11-scaled cross terms collect in a second accumulator set that is
// folded into `acc` before the function returns (acc lives across head groups / FFN chunks in the callers).
template <int MT, int NTW, int NKB, bool PRE = false, bool F16 = false>
DEVI void gemm_tall_split_st_b(f32x4 (&acc)[NTW][MT], int LS2 /* dwords per piece row */, const lu32* as, int R, int rowsA,
                             const unsigned* __restrict__ Wp, int KBtot, int kb0, int ntiles, u32x4 (&b)[NKB < 4 ? NKB : 4][NTW][3]) {
    constexpr int NP = F16 ? 2 : 3;
    const int tid_ = tid_now();
    constexpr int D = NKB < 4 ? NKB : 4;
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kg = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * LS2 + 4 * kg;
    const WPtr<gu32x4, DFF_WMODE(MT)> wp((const gu32x4*)Wp, (unsigned)lane & 63u);
    size_t tbase[NTW];
    bool tok[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + DFF_NWAVES * i;
        tok[i] = nt < ntiles;
        tbase[i] = ((size_t)(tok[i] ? nt : 0) * KBtot + kb0) * NP;
    }
    if (!tok[0]) return;
    if (!PRE) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[d][i][p] = wp[(tbase[i] + NP * d + p) * 64];
    }
    f32x4 acc2[F16 ? NTW : 1][F16 ? MT : 1];
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);   // issue the ring's loads here (see gemm_wide_split_st)
    // TPRE (up to three row tiles: 24 .. 36 more registers; trp-cage -1.3 %, villin -0.8 %, BBA and protein G neutral): the A fragments of k-block kb + 1 are requested before the products of
    // k-block kb -- otherwise every k-block starts on the LDS latency of its own operands
    constexpr bool TPRE = MT <= DFF_TPRE_MT;
    u32x4 nh[TPRE ? MT : 1], nm[TPRE ? MT : 1], nl[TPRE ? MT : 1];
    auto a_load = [&](u32x4 (&xl)[TPRE ? MT : 1], u32x4 (&xh)[TPRE ? MT : 1], u32x4 (&xm)[TPRE ? MT : 1], int kb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if constexpr (!F16) xl[mt] = *(const volatile lu32x4*)(as + 2 * R * LS2 + rowoff[mt] + 16 * kb);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xh[mt] = *(const volatile lu32x4*)(as + rowoff[mt] + 16 * kb);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xm[mt] = *(const volatile lu32x4*)(as + R * LS2 + rowoff[mt] + 16 * kb);
    };
    if constexpr (TPRE) a_load(nl, nh, nm, 0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int d = kb % D;
        u32x4 ah[MT], am[MT], al[MT];
        // (requested in the order the products consume them -- l, h, m pieces -- so that the first product waits for one
        // read, not for nine)
        if constexpr (TPRE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { al[mt] = nl[mt]; ah[mt] = nh[mt]; am[mt] = nm[mt]; }
            if (kb + 1 < NKB) a_load(nl, nh, nm, kb + 1);
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if constexpr (!F16) al[mt] = *(const volatile lu32x4*)(as + 2 * R * LS2 + rowoff[mt] + 16 * kb);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = *(const volatile lu32x4*)(as + rowoff[mt] + 16 * kb);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) am[mt] = *(const volatile lu32x4*)(as + R * LS2 + rowoff[mt] + 16 * kb);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i)
            if (i == 0 || tok[i]) {
                if constexpr (F16) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = mfma_f16(b[d][i][0], am[mt], acc2[i][mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = mfma_f16(b[d][i][1], ah[mt], acc2[i][mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_f16(b[d][i][0], ah[mt], acc[i][mt]);
                } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], al[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][2], ah[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][1], am[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], am[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][1], ah[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], ah[mt], acc[i][mt]);
                }
            }
        if (kb + D < NKB) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[d][i][p] = wp[(tbase[i] + NP * (kb + D) + p) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] += acc2[i][mt] * DFF_F16_LINV;
    }
}

template <int MT, int NTW, int NKB, bool F16 = false>
DEVI void gemm_tall_split_st(f32x4 (&acc)[NTW][MT], int LS2, const lu32* as, int R, int rowsA,
                             const unsigned* __restrict__ Wp, int KBtot, int kb0, int ntiles) {
    u32x4 b[NKB < 4 ? NKB : 4][NTW][3];
    gemm_tall_split_st_b<MT, NTW, NKB, false, F16>(acc, LS2, as, R, rowsA, Wp, KBtot, kb0, ntiles, b);
}

#ifndef DFF_WOPRE
#define DFF_WOPRE 1
#endif
#ifndef DFF_L2W
#define DFF_L2W 1
#endif
#ifndef DFF_PSPLIT
#define DFF_PSPLIT 1   // co_dqkv_rows: dQ / dK as bf16 pieces
#endif
#ifndef DFF_DQKV_ROWS
#define DFF_DQKV_ROWS 1   // four row tiles: a wave owns a row tile's column tiles in the three-phase dV / dQ / dK products
#endif
#ifndef DFF_GXTILE0
#define DFF_GXTILE0 1
#endif
#ifndef DFF_GXTILE
#define DFF_GXTILE 1   // backward head pipeline: a spare wave parks a whole G_ext tile (all row tiles) where tiles == spare waves
#endif
#ifndef DFF_XFAST
#define DFF_XFAST 1   // PAIR: plain stores / L2-served loads when both blocks of a pair report the same XCD (0: always sc1)
#endif
#ifndef DFF_GXT
#define DFF_GXT 1
#endif
#ifndef DFF_K2
#define DFF_K2 1   // wide split GEMMs at four row tiles: two output tiles per wave off one A read (gemm_wide_split_k2; three row tiles: measured 0.6 % slower, 204 B of scratch)
#endif
#ifndef DFF_EXTPRE
#define DFF_EXTPRE 1   // extension-block weights of the tall GEMMs requested before the split GEMM (0: behind it, as until round 3)
#endif
#ifndef DFF_QSP
#define DFF_QSP 1   // dQ leaves co_ds as bf16 pieces (0: fp32, split by every wave of the back-projection)
#endif
// The knobs above are measured decisions (DESIGN.md / DESIGN_HISTORY.md name each A/B); a build that changes one is a development
// build and has to say so: -DDFF_EXPERIMENT, which dff_version() reports next to the flags.
#if !defined(DFF_EXPERIMENT) && (DFF_F16G != 15 || DFF_AUXLATE != 1 || DFF_APRE != 1 || DFF_PIPEB_128_2 != 1 || DFF_QTPRE != 1 || DFF_TPRE_MT != 3 || \
     DFF_K2_MT != 3 || DFF_ARES_LIM != 8 || DFF_ARES != 1 || DFF_WOPRE != 1 || DFF_L2W != 1 || DFF_PSPLIT != 1 || DFF_DQKV_ROWS != 1 || \
     DFF_GXTILE0 != 1 || DFF_GXTILE != 1 || DFF_XFAST != 1 || DFF_GXT != 1 || DFF_K2 != 1 || DFF_EXTPRE != 1 || DFF_QSP != 1)
#error "non-default tuning knobs: a development build -- add -DDFF_EXPERIMENT (dff_version() then says so)"
#endif
// L2 warm-up.  The weights of a phase are what all 32 workgroups of an XCD ask their L2 for at about the same time; they are
// 15 MB per step (villin) against 4 MB of L2, so whoever is first pays the trip to memory and the others queue behind the
// same lines: the convoy moves at the pace of a miss per phase.  Here each workgroup requests 1/32 of the NEXT phase's lines
// while the current phase runs (line i by the workgroup with (blockIdx.x >> 3) % 32 == i % 32: blocks b and b + 8 share an
// XCD under the observed placement -- used for speed only): one instruction of one wave covers 256 KiB of weights.  The
// requests are LDS-DMA loads into a 256-byte junk pad: no register waits for the data and nothing reads it.  (Touching
// whole regions from every workgroup was measured too: the texture unit takes a cycle per line, villin 577 -> 641 us.)
// Region: `ntiles` pieces of `lpt` 128-byte lines, `stride` bytes apart.
DEVI void l2_touch(unsigned junk_byte, const void* base, int ntiles, size_t stride, int lpt) {
    const int lane = tid_now() & 63;
    const int q = (int)(blockIdx.x >> 3) & 31;
    const int total = ntiles * lpt;
    for (int l0 = 0; l0 < total; l0 += 2048) {
        const int ln = l0 + lane * 32 + q;
        if (ln < total) {
            const int t = ln / lpt, w = ln - t * lpt;
            const char __attribute__((address_space(1)))* p =
                (const char __attribute__((address_space(1)))*)base + (size_t)t * stride + (size_t)w * 128;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" ::"s"(junk_byte), "v"(p) : "memory");
        }
    }
}

// QSP (round 4): dQ arrives as pieces as well (co_ds: [h | m] in place of the fp32 row of buffer regQ, l in `lsq`).
// F16 (round 5): two-piece fp16 operands -- the in-place pieces are [h | l'] (put_piece16), nothing lives in lsp / lsq, operands
// that arrive as fp32 rows are scaled by qs and split here (split8h); `acc` collects scaled units (the caller's row stage
// multiplies the inverse back).
template <int MT, int NTW, int HGS, bool KVS = false, bool VSP = false, int PRE = 0, bool QSP = false, bool F16 = false>
DEVI void gemm_tall_qkvT_split(f32x4 (&acc)[NTW][MT], const lfloat* Rg, int regQ, int RN, const unsigned* __restrict__ Ws,
                               int head0, int ntiles, const lu32* lsp, u32x4 (&b)[4][NTW][3], const lu32* lsq = nullptr, float qs = 1.0f) {
    constexpr int LQ = 80 * HGS + 4, NKB = 6 * HGS, D = 4, KBtot = 6 * DFF_HEADS, NPC = F16 ? 2 : 3;
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kg = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, RN - 1) * LQ + 8 * kg;
    const gu32x4* wp = (const gu32x4*)Ws + lane;
    size_t tbase[NTW];
    bool tok[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + DFF_NWAVES * i;
        tok[i] = nt < ntiles;
        tbase[i] = ((size_t)(tok[i] ? nt : 0) * KBtot + 6 * head0) * NPC;
    }
    if (!tok[0]) return;
    {
#pragma unroll
        for (int d = PRE; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int p = 0; p < NPC; ++p) b[d][i][p] = wp[(tbase[i] + NPC * d + p) * 64];
    }
    f32x4 acc2[F16 ? NTW : 1][F16 ? MT : 1];
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    // QTPRE: every operand arrives as pieces (dQ, dK and dV all split by their producers) and the shape has up to three row
    // tiles: the fragments of k-block kb + 1 are requested before the products of kb (as in gemm_tall_split_st_b)
    constexpr bool QTPRE = KVS && VSP && QSP && MT <= DFF_TPRE_MT && DFF_QTPRE;
    u32x4 nh[QTPRE ? MT : 1], nm[QTPRE ? MT : 1], nl[QTPRE ? MT : 1];
    auto p_load = [&](u32x4 (&xl)[QTPRE ? MT : 1], u32x4 (&xh)[QTPRE ? MT : 1], u32x4 (&xm)[QTPRE ? MT : 1], int kb) {
        constexpr int LSV = 32 * HGS + 4;
        const int hh = kb / 6, part = (kb % 6) / 2, half = kb % 2;
        const lu32* const hb = (const lu32*)(Rg + (part == 0 ? regQ : part) * RN * LQ + hh * 80) + 16 * half;
        const lu32* const lb = part == 1 ? (const lu32*)(Rg + (1 + half) * RN * LQ + hh * 80 + 64)
                                         : (part == 2 ? lsp : lsq) + hh * 32 + 16 * half;
        const int lmul = part == 1 ? LQ : LSV;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if constexpr (!F16) xl[mt] = *(const volatile lu32x4*)(lb + min(mt * 16 + mm, RN - 1) * lmul + 4 * kg);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xh[mt] = *(const volatile lu32x4*)(hb + rowoff[mt] - 4 * kg);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xm[mt] = *(const volatile lu32x4*)(hb + 32 + rowoff[mt] - 4 * kg);
    };
    if constexpr (QTPRE) p_load(nl, nh, nm, 0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int d = kb % D;
        const int hh = kb / 6, part = (kb % 6) / 2, half = kb % 2;
        const int aoff = (part == 0 ? regQ : part) * RN * LQ + hh * 80 + 32 * half;
        u32x4 ah[MT], am[MT], al[MT];
        if constexpr (QTPRE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { al[mt] = nl[mt]; ah[mt] = nh[mt]; am[mt] = nm[mt]; }
            if (kb + 1 < NKB) p_load(nl, nh, nm, kb + 1);
        } else if (KVS && (part == 1 || (VSP && part == 2) || (QSP && part == 0))) {
            constexpr int LSV = 32 * HGS + 4;
            const lu32* const hb = (const lu32*)(Rg + (part == 0 ? regQ : part) * RN * LQ + hh * 80) + 16 * half;
            const lu32* const lb = part == 1 ? (const lu32*)(Rg + (1 + half) * RN * LQ + hh * 80 + 64)
                                             : (part == 2 ? lsp : lsq) + hh * 32 + 16 * half;
            const int lmul = part == 1 ? LQ : LSV;
            // (l, h, m: the order the products consume them)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) if constexpr (!F16) al[mt] = *(const volatile lu32x4*)(lb + min(mt * 16 + mm, RN - 1) * lmul + 4 * kg);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ah[mt] = *(const volatile lu32x4*)(hb + rowoff[mt] - 4 * kg);   // row * LQ + 4 kg
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) am[mt] = *(const volatile lu32x4*)(hb + 32 + rowoff[mt] - 4 * kg);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const lfloat* ap = Rg + aoff + rowoff[mt];
                const f32x4 x0 = *(const lf32x4*)ap, x1 = *(const lf32x4*)(ap + 4);
                if constexpr (F16) split8h(x0 * qs, x1 * qs, ah[mt], am[mt]);
                else split8(x0, x1, ah[mt], am[mt], al[mt]);
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i)
            if (i == 0 || tok[i]) {
                if constexpr (F16) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = mfma_f16(b[d][i][0], am[mt], acc2[i][mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc2[i][mt] = mfma_f16(b[d][i][1], ah[mt], acc2[i][mt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_f16(b[d][i][0], ah[mt], acc[i][mt]);
                } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], al[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][2], ah[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][1], am[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], am[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][1], ah[mt], acc[i][mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma_bf16(b[d][i][0], ah[mt], acc[i][mt]);
                }
            }
        if (kb + D < NKB) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int p = 0; p < NPC; ++p) b[d][i][p] = wp[(tbase[i] + NPC * (kb + D) + p) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] += acc2[i][mt] * DFF_F16_LINV;
    }
}

// wide GEMM with few output tiles (NTN * MT (tile, row-tile) units <= a few per wave): the UNITS, not the tiles, go
// round-robin over the waves, so that all four SIMDs carry the same MFMA load; loop-free, all weights loaded up
// front, hook() as in gemm_wide.  epi(nt_local, mt, acc).
template <int MT, int KB, int NTN, class Epi, class Hook>
DEVI void gemm_wide_units(const lfloat* A, int lda, int rowsA, const float* __restrict__ Wp, int KBtot,
                          int kb0, int nt0, Epi epi, Hook hook) {
    const int tid_ = tid_now();
    constexpr int NU = NTN * MT, DU = (NU + DFF_NWAVES - 1) / DFF_NWAVES;
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kk = lane >> 4, mm = lane & 15;
    const WPtr<gf32x4, DFF_WMODE(MT)> wp((const gf32x4*)Wp + (size_t)kb0 * 64, (unsigned)lane & 63u);
    f32x4 b[DU][KB];
#pragma unroll
    for (int d = 0; d < DU; ++d) {
        const int u = wave + DFF_NWAVES * d;
        if (u < NU) {
            const int nt = u / MT;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b[d][kb] = wp[((size_t)(nt0 + nt) * KBtot + kb) * 64];
        }
    }
    hook();
#pragma unroll
    for (int d = 0; d < DU; ++d) {
        const int u = wave + DFF_NWAVES * d;
        if (u < NU) {
            const int nt = u / MT, mt = u - nt * MT;
            const lfloat* ap = A + min(mt * 16 + mm, rowsA - 1) * lda + 4 * kk;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; kb += 2) {
                const f32x4 a0 = *(const lf32x4*)(ap + 16 * kb), a1 = *(const lf32x4*)(ap + 16 * kb + 16);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[d][kb][s4], a0[s4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[d][kb + 1][s4], a1[s4], acc1, 0, 0, 0);
                }
            }
            epi(nt, mt, acc0 + acc1);
        }
    }
}

// gemm_tall_kb with a compile-time block count: loop-free (exact waits for the ring of D k-blocks)
template <int MT, int NTW, int XPER, int NKB, class KF>
DEVI void gemm_tall_kb_st(f32x4 (&acc)[NTW][MT], KF kf, const lfloat* A, int lda, int rowsA,
                          const float* __restrict__ Wp, int KBtot, int ntiles) {
    const int tid_ = tid_now();
    constexpr int D = NKB < 4 ? NKB : 4;
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kk = lane >> 4, mm = lane & 15;
    int rowoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowoff[mt] = min(mt * 16 + mm, rowsA - 1) * lda + 4 * kk;
    const WPtr<gf32x4, DFF_WMODE(MT)> wp((const gf32x4*)Wp, (unsigned)lane & 63u);
    size_t tbase[NTW];
    bool tok[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + DFF_NWAVES * i;
        tok[i] = nt < ntiles;
        tbase[i] = (size_t)(tok[i] ? nt : 0) * KBtot;
    }
    if (!tok[0]) return;
    f32x4 b[D][NTW];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        int aoff_, wkb;
        kf(d, aoff_, wkb);
#pragma unroll
        for (int i = 0; i < NTW; ++i) b[d][i] = wp[(tbase[i] + wkb) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ib = 0; ib < NKB; ++ib) {
        const int d = ib % D;
        int aoff, wkb;
        kf(ib, aoff, wkb);
        const bool ext = XPER > 0 && wkb % (XPER > 0 ? XPER : 1) == 4;
        if (ext) {
            float ax[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ax[mt] = A[aoff + rowoff[mt] - 3 * kk];
#pragma unroll
            for (int i = 0; i < NTW; ++i)
                if (i == 0 || tok[i]) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[d][i][0], ax[mt], acc[i][mt], 0, 0, 0);
                }
        } else {
            f32x4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *(const lf32x4*)(A + aoff + rowoff[mt]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < NTW; ++i)
                    if (i == 0 || tok[i]) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[d][i][s4], a[mt][s4], acc[i][mt], 0, 0, 0);
                    }
        }
        if (ib + D < NKB) {
            int aoff2, wkb2;
            kf(ib + D, aoff2, wkb2);
#pragma unroll
            for (int i = 0; i < NTW; ++i) b[d][i] = wp[(tbase[i] + wkb2) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The extension k-steps that follow a split tall GEMM (one fp32 k-step per head of the group, weights = the s = 0 slots of the
// fp32 image's extension blocks), in two halves: the weights are requested BEFORE the split GEMM and multiplied after it.
// As one call behind the GEMM (gemm_tall_kb) their trip to the L2 was fully exposed, once per head and layer in the forward
// output projection and again in the back-projection.
template <int NTW, int NH>
struct ExtW { float b[NH][NTW]; };
template <int NTW, int NH, int MT = 1, class WK>
DEVI void ext_fetch(ExtW<NTW, NH>& e, WK wk, const float* __restrict__ Wp, int KBtot, int ntiles) {
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const WPtr<gf32x4, DFF_WMODE(MT)> wp((const gf32x4*)Wp, (unsigned)lane & 63u);
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int nt = wave + DFF_NWAVES * t;
            e.b[i][t] = wp.first_float(((size_t)(nt < ntiles ? nt : 0) * KBtot + wk(i)) * 64);
        }
    asm volatile("" ::: "memory");
}
template <int MT, int NTW, int NH, class AO>
DEVI void ext_apply(f32x4 (&acc)[NTW][MT], const ExtW<NTW, NH>& e, AO aoff_of, const lfloat* A, int lda, int rowsA, int ntiles, float wscale = 1.0f) {
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int kk = lane >> 4, mm = lane & 15;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        float ax[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ax[mt] = A[aoff_of(i) + min(mt * 16 + mm, rowsA - 1) * lda + kk];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (t == 0 || wave + DFF_NWAVES * t < ntiles) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(e.b[i][t] * wscale, ax[mt], acc[t][mt], 0, 0, 0);
            }
    }
}

template <int MT, int NTW>
DEVI void acc_zero(f32x4 (&acc)[NTW][MT]) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// write the tall accumulators (+ optional bias) to an LDS row buffer
template <int MT, int NTW>
DEVI void store_tall(const f32x4 (&acc)[NTW][MT], float* out, int ld, int rows, int ntiles,
                     const float* __restrict__ bias) {
    const int tid_ = tid_now();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int quad = lane >> 4, rl = lane & 15;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + DFF_NWAVES * i;
        if (nt >= ntiles) continue;
        const f32x4 bv = bias ? *(const f32x4*)(bias + 16 * nt + 4 * quad) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + rl;
            if (row < rows) *(f32x4*)(out + row * ld + 16 * nt + 4 * quad) = acc[i][mt] + bv;
        }
    }
}


// 91-92,100-103) -> resbuf, stashed as nodes_in of layer 0.
template <int H, bool GEN>
DEVI void node_embed(const Ctx& c, const DffModelDev& m) {
    const int tid_ = tid_now();
    for (int idx = tid_; idx < c.rows * H; idx += DFF_NTHREADS) {
        const int row = idx / H, col = idx - row * H;
        const int g = row / c.N, i = row - g * c.N;
        float v = m.WnT[i * H + col] + c.tn[g] * m.WnT[(GEN ? m.wn_t : c.N) * H + col] + m.bn[col];
        if (GEN && m.in_abs) {   // nodes = [one-hot, x, t]  (graph_transformer.py:99-102)
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) v += c.xs[row * 4 + c3] * m.WnT[(c.N + c3) * H + col];
        }
        c.resbuf[row * (H + 4) + col] = v;
    }
}
// ------------------------------------------------------------------------------------------
// Row stages (LayerNorm, gates and their backward forms).  LP lanes share a row, HC = H / LP columns per lane, in groups of
// VW consecutive columns (element e = j VW + k of a lane is column VW sub + VW LP j + k): every LDS / stash / parameter
// access of a lane is a 16- (or 8-) byte vector, the LP lanes of a row cover 4 LP consecutive floats per access.  LP is
// a template argument (512 / LP rows per pass; the kernel uses 16), and each
// stage issues every global load it needs (stashed rows, gate weights, LayerNorm gains) BEFORE the first reduction: round 2's
// stages loaded a scalar, waited, used it, 16 to 100 times per pass (b_ln2+gate1: 36 k cycles per call on protein G).
// ------------------------------------------------------------------------------------------
template <int H, int LP>
struct RowMap {
    static constexpr int HC = H / LP, VW = (HC % 4 == 0) ? 4 : 2, NV = HC / VW, RPP = DFF_NTHREADS / LP;   // rows per pass
    static_assert(H % LP == 0 && HC % 2 == 0, "row layout");
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// HC values of a row-shaped array (p = the row's first element: LDS tile row, stash row or a parameter vector)
template <int H, int LP>
DEVI void rload(float (&x)[H / LP], const float* p, int sub) {
    using M = RowMap<H, LP>;
#pragma unroll
    for (int j = 0; j < M::NV; ++j) {
        const float* q = p + M::VW * sub + M::VW * LP * j;
        if constexpr (M::VW == 4) { const f32x4 v = *(const f32x4*)q; x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3]; }
        else { const f32x2 v = *(const f32x2*)q; x[2 * j] = v[0]; x[2 * j + 1] = v[1]; }
    }
}
template <int H, int LP>
DEVI void rstore(float* p, const float (&x)[H / LP], int sub) {
    using M = RowMap<H, LP>;
#pragma unroll
    for (int j = 0; j < M::NV; ++j) {
        float* q = p + M::VW * sub + M::VW * LP * j;
        if constexpr (M::VW == 4) *(f32x4*)q = (f32x4){x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
        else *(f32x2*)q = (f32x2){x[2 * j], x[2 * j + 1]};
    }
}
// A row stage's K = H GEMM input.  fp32 engine: the fp32 row into abuf.  Split engine (round 4): the three bf16 pieces
// straight into the split A operand (as[piece][row][LHS2], what split_rows used to make of abuf in a pass -- and a workgroup
// barrier -- of its own, four times per layer); abuf does not exist in those variants.
template <int H, int LP, bool SPW, bool F16 = false>
DEVI void rstore_a(const Ctx& c, int row, const float (&x)[H / LP], int sub) {
    using M = RowMap<H, LP>;
    if constexpr (!SPW) {
        rstore<H, LP>(c.abuf + row * (H + 4), x, sub);
    } else {
        constexpr int LHS2 = (H + DFF_SPAD) / 2;
#pragma unroll
        for (int j = 0; j < M::NV; ++j) {
            const int col = M::VW * sub + M::VW * LP * j;
            if constexpr (M::VW == 4) {
                store_split4<F16>(c.asp, c.RNa, LHS2, row, col, (f32x4){x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]});
            } else if constexpr (F16) {
                unsigned h, l;
                split2h(x[2 * j], x[2 * j + 1], h, l);
                const int o = row * LHS2 + (col >> 1);
                c.asp[0 * c.RNa * LHS2 + o] = h;
                c.asp[1 * c.RNa * LHS2 + o] = l;
            } else {
                unsigned hh[2], mm[2], ll[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float v = x[2 * j + q];
                    const unsigned uh = __float_as_uint(v) & 0xffff0000u;
                    const float r = v - __uint_as_float(uh);
                    const unsigned um = __float_as_uint(r) & 0xffff0000u;
                    hh[q] = uh; mm[q] = um; ll[q] = __float_as_uint(r - __uint_as_float(um));
                }
                const int o = row * LHS2 + (col >> 1);
                c.asp[0 * c.RNa * LHS2 + o] = __builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u);
                c.asp[1 * c.RNa * LHS2 + o] = __builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u);
                c.asp[2 * c.RNa * LHS2 + o] = __builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u);
            }
        }
    }
}
template <int LP>
DEVI float grp_max_lp(float v) {   // all-reduce (max) over the LP lanes of a row
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    if constexpr (LP == 16) v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
// fp16 engine, backward: scale a row-stage output row (a gradient: any magnitude) by the power of two that brings its maximum to
// [16, 32) before it is split into fp16 pieces; the inverse goes to c.rsc[row] for whoever takes the chain's result back
// (everything between two row stages is linear per row).  Exact.
template <int H, int LP>
DEVI void row_pow2_scale(const Ctx& c, int row, float (&x)[H / LP], int sub) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < H / LP; ++i) mx = fmaxf(mx, fabsf(x[i]));
    float sc, inv;
    pow2_scale(grp_max_lp<LP>(mx), sc, inv);
#pragma unroll
    for (int i = 0; i < H / LP; ++i) x[i] *= sc;
    if (sub == 0) c.rsc[row] = inv;
}
template <int LP>
DEVI float grp_sum_lp(float v) {   // all-reduce over the LP (8 or 16) lanes of a row
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    if constexpr (LP == 16) v += dpp_mov<0x140>(v);
    return v;
}
// absolute coordinates: dE/dx_i += d(nodes_0)_i . W_node[:, x columns]   (dn_0 is in resbuf after layer 0's backward)
template <int H, int LP>
DEVI void node_embed_bwd(const Ctx& c, const DffModelDev& m) {
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float dn[HC], w[3][HC];
        rload<H, LP>(dn, c.resbuf + row * LH, sub);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) rload<H, LP>(w[c3], m.WnT + (c.N + c3) * H, sub);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            float f = 0.f;
#pragma unroll
            for (int i = 0; i < HC; ++i) f += dn[i] * w[c3][i];
            f = grp_sum_lp<LP>(f);
            if (sub == 0) c.dxs[row * 4 + c3] += f;
        }
    }
}

// LayerNorm of one row held as HC values per lane of an LP-lane group
template <int H, int LP>
DEVI void ln_stats(const float (&x)[H / LP], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < H / LP; ++i) s += x[i];
    mean = grp_sum_lp<LP>(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < H / LP; ++i) { const float d = x[i] - mean; q += d * d; }
    const float var = grp_sum_lp<LP>(q) * (1.0f / H);
    rstd = fast_rsqrt(var + 1e-5f);
}

// gate = sigmoid(w . [x, res, x-res])   (graph_transformer.py:197-205); w = the three weight vectors, already in registers
template <int H, int LP>
DEVI float gate_value(const float (&x)[H / LP], const float (&res)[H / LP], const float (&w)[3][H / LP]) {
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < H / LP; ++i) z += x[i] * w[0][i] + res[i] * w[1][i] + (x[i] - res[i]) * w[2][i];
    return sigmoid_f(grp_sum_lp<LP>(z));
}
template <int H, int LP>
DEVI void gate_weights(float (&w)[3][H / LP], const float* g, int sub) {
    rload<H, LP>(w[0], g, sub); rload<H, LP>(w[1], g + H, sub); rload<H, LP>(w[2], g + 2 * H, sub);
}

// R0: nodes (resbuf) -> stash nodes_in ; LN1 -> abuf
template <int H, int LP, bool SPW, bool F16 = false>
DEVI void row_ln1(const Ctx& c, const DffLayerDev& lw, int l) {   // (F16: the GEMM behind it takes two-piece fp16 operands)
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    float* s_nodes = c.stash + (size_t)l * c.sl.layer_stride + c.sl.nodes_in;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float x[HC], gam[HC], bet[HC];
        rload<H, LP>(gam, lw.ln1_g, sub); rload<H, LP>(bet, lw.ln1_b, sub);
        rload<H, LP>(x, c.resbuf + row * LH, sub);
        rstore<H, LP>(s_nodes + row * H, x, sub);
        float mean, rstd;
        ln_stats<H, LP>(x, mean, rstd);
#pragma unroll
        for (int i = 0; i < HC; ++i) x[i] = (x[i] - mean) * rstd * gam[i] + bet[i];
        rstore_a<H, LP, SPW, F16>(c, row, x, sub);
    }
}

// R1: tbuf = attn_out, resbuf = nodes -> nodes1 (resbuf), stash attn_out, LN2 -> abuf
template <int H, int LP, bool SPW, bool F16 = false>
DEVI void row_gate1_ln2(const Ctx& c, const DffLayerDev& lw, int l, const float* tbuf) {
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    float* s_att = c.stash + (size_t)l * c.sl.layer_stride + c.sl.attn_out;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float x[HC], res[HC], n1[HC], w[3][HC], gam[HC], bet[HC];
        gate_weights<H, LP>(w, lw.g1, sub);
        rload<H, LP>(gam, lw.ln2_g, sub); rload<H, LP>(bet, lw.ln2_b, sub);
        rload<H, LP>(x, tbuf + row * LH, sub);
        rload<H, LP>(res, c.resbuf + row * LH, sub);
        rstore<H, LP>(s_att + row * H, x, sub);
        const float g = gate_value<H, LP>(x, res, w);
#pragma unroll
        for (int i = 0; i < HC; ++i) n1[i] = x[i] * g + res[i] * (1.0f - g);
        rstore<H, LP>(c.resbuf + row * LH, n1, sub);
        float mean, rstd;
        ln_stats<H, LP>(n1, mean, rstd);
#pragma unroll
        for (int i = 0; i < HC; ++i) n1[i] = (n1[i] - mean) * rstd * gam[i] + bet[i];
        rstore_a<H, LP, SPW, F16>(c, row, n1, sub);
    }
}

// R2: tbuf = ff, resbuf = nodes1 -> nodes2 (resbuf), stash ff.  Last layer: energy + dn = w_dec.
template <int H, int LP>
DEVI void row_gate2(const Ctx& c, const DffModelDev& m, const DffLayerDev& lw, int l, const float* tbuf,
                    bool last, float* energy_out) {
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    float* s_ff = c.stash + (size_t)l * c.sl.layer_stride + c.sl.ff;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float x[HC], res[HC], w[3][HC];
        gate_weights<H, LP>(w, lw.g2, sub);
        rload<H, LP>(x, tbuf + row * LH, sub);
        rload<H, LP>(res, c.resbuf + row * LH, sub);
        rstore<H, LP>(s_ff + row * H, x, sub);
        const float g = gate_value<H, LP>(x, res, w);
        if (last && !m.conservative) {
            // force head (graph_transformer.py:62-63,112-113): forces = node_decoder(nodes), no VJP;
            // the update stage takes dE/dx, so the negated forces go there
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
                float wd[HC], f = 0.f;
                rload<H, LP>(wd, m.wdec + c3 * H, sub);
#pragma unroll
                for (int i = 0; i < HC; ++i) f += (x[i] * g + res[i] * (1.0f - g)) * wd[i];
                f = grp_sum_lp<LP>(f);
                if (sub == 0) c.dxs[row * 4 + c3] = -(f + m.bdec3[c3]);
            }
            continue;
        }
        if (last) {
            float wd[HC], e = 0.f;
            rload<H, LP>(wd, m.wdec, sub);
#pragma unroll
            for (int i = 0; i < HC; ++i) e += (x[i] * g + res[i] * (1.0f - g)) * wd[i];
            rstore<H, LP>(c.resbuf + row * LH, wd, sub);   // d(sum_i e_i)/d nodes_L
            if (energy_out) {
                e = grp_sum_lp<LP>(e);
                if (sub == 0) energy_out[(size_t)c.b0 * c.N + row] = e + m.bdec;
            }
        } else {
#pragma unroll
            for (int i = 0; i < HC; ++i) x[i] = x[i] * g + res[i] * (1.0f - g);
            rstore<H, LP>(c.resbuf + row * LH, x, sub);
        }
    }
}

// RB1: dn (resbuf) through gate2 -> dff (abuf), dn1 partial (resbuf)
template <int H, int LP, bool SPW, bool F16 = false>
DEVI void rowb_gate2(const Ctx& c, const DffLayerDev& lw, int l) {
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    const float* sb = c.stash + (size_t)l * c.sl.layer_stride;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float ao[HC], nin[HC], n1[HC], ff[HC], dn[HC], w1[3][HC], w2[3][HC];
        rload<H, LP>(ao, sb + c.sl.attn_out + row * H, sub);
        rload<H, LP>(nin, (l == 0 ? c.l0 : sb) + c.sl.nodes_in + row * H, sub);
        rload<H, LP>(ff, sb + c.sl.ff + row * H, sub);
        gate_weights<H, LP>(w1, lw.g1, sub);
        gate_weights<H, LP>(w2, lw.g2, sub);
        rload<H, LP>(dn, c.resbuf + row * LH, sub);
        const float g1 = gate_value<H, LP>(ao, nin, w1);
#pragma unroll
        for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + nin[i] * (1.0f - g1);
        const float g2 = gate_value<H, LP>(ff, n1, w2);
        float dg = 0.f;
#pragma unroll
        for (int i = 0; i < HC; ++i) dg += dn[i] * (ff[i] - n1[i]);
        dg = grp_sum_lp<LP>(dg);
        const float dz = dg * g2 * (1.0f - g2);
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            ao[i] = dn[i] * g2 + dz * (w2[0][i] + w2[2][i]);
            nin[i] = dn[i] * (1.0f - g2) + dz * (w2[1][i] - w2[2][i]);
        }
        if constexpr (F16) row_pow2_scale<H, LP>(c, row, ao, sub);   // (the FFN backward chain runs in scaled units)
        rstore_a<H, LP, SPW, F16>(c, row, ao, sub);
        rstore<H, LP>(c.resbuf + row * LH, nin, sub);
    }
}

// RB2: tbuf = df ; dn1 = resbuf + LN2bwd(df) ; gate1 bwd -> dattn (abuf), dn_in partial (resbuf)
template <int H, int LP, bool SPW, bool FIN = false, bool FOUT = false>
DEVI void rowb_ln2_gate1(const Ctx& c, const DffLayerDev& lw, int l, const float* tbuf) {
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    const float* sb = c.stash + (size_t)l * c.sl.layer_stride;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float ao[HC], nin[HC], n1[HC], d1[HC], w[3][HC], gam[HC], df[HC], dnp[HC];
        rload<H, LP>(ao, sb + c.sl.attn_out + row * H, sub);
        rload<H, LP>(nin, (l == 0 ? c.l0 : sb) + c.sl.nodes_in + row * H, sub);
        gate_weights<H, LP>(w, lw.g1, sub);
        rload<H, LP>(gam, lw.ln2_g, sub);
        rload<H, LP>(df, tbuf + row * LH, sub);
        if constexpr (FIN) {   // fp16 engine: the FFN backward chain ran in this row's scaled units (rowb_gate2)
            const float rinv = c.rsc[row];
#pragma unroll
            for (int i = 0; i < HC; ++i) df[i] *= rinv;
        }
        rload<H, LP>(dnp, c.resbuf + row * LH, sub);
        const float g1 = gate_value<H, LP>(ao, nin, w);
#pragma unroll
        for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + nin[i] * (1.0f - g1);
        float mean, rstd;
        ln_stats<H, LP>(n1, mean, rstd);
        float s1 = 0.f, s2 = 0.f;
        float dyg[HC], xh[HC];
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            xh[i] = (n1[i] - mean) * rstd;
            dyg[i] = df[i] * gam[i];
            s1 += dyg[i];
            s2 += dyg[i] * xh[i];
        }
        s1 = grp_sum_lp<LP>(s1) * (1.0f / H);
        s2 = grp_sum_lp<LP>(s2) * (1.0f / H);
        float dg = 0.f;
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            d1[i] = dnp[i] + rstd * (dyg[i] - s1 - xh[i] * s2);
            dg += d1[i] * (ao[i] - nin[i]);
        }
        dg = grp_sum_lp<LP>(dg);
        const float dz = dg * g1 * (1.0f - g1);
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            ao[i] = d1[i] * g1 + dz * (w[0][i] + w[2][i]);
            nin[i] = d1[i] * (1.0f - g1) + dz * (w[1][i] - w[2][i]);
        }
        if constexpr (FOUT) row_pow2_scale<H, LP>(c, row, ao, sub);   // (dattn row-scaled: the G_ext GEMM's epilogues multiply the inverse back)
        rstore_a<H, LP, SPW, FOUT>(c, row, ao, sub);
        rstore<H, LP>(c.resbuf + row * LH, nin, sub);
    }
}

// RB3: tbuf = d(LN1 out) ; dn = resbuf + LN1bwd
template <int H, int LP>
DEVI void rowb_ln1(const Ctx& c, const DffLayerDev& lw, int l, const float* tbuf, float in_scale = 1.0f) {   // (in_scale: fp16 engine, CoGeo::qsi)
    const int tid_ = tid_now();
    constexpr int HC = H / LP, LH = H + 4;
    const int grp = tid_ / LP, sub = tid_ % LP;
    const float* sb = c.stash + (size_t)l * c.sl.layer_stride;
    for (int row = grp; row < c.rows; row += RowMap<H, LP>::RPP) {
        float nin[HC], gam[HC], dy[HC], dnp[HC];
        rload<H, LP>(nin, (l == 0 ? c.l0 : sb) + c.sl.nodes_in + row * H, sub);
        rload<H, LP>(gam, lw.ln1_g, sub);
        rload<H, LP>(dy, tbuf + row * LH, sub);
#pragma unroll
        for (int i = 0; i < HC; ++i) dy[i] *= in_scale;
        rload<H, LP>(dnp, c.resbuf + row * LH, sub);
        float mean, rstd;
        ln_stats<H, LP>(nin, mean, rstd);
        float s1 = 0.f, s2 = 0.f, dyg[HC], xh[HC];
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            xh[i] = (nin[i] - mean) * rstd;
            dyg[i] = dy[i] * gam[i];
            s1 += dyg[i];
            s2 += dyg[i] * xh[i];
        }
        s1 = grp_sum_lp<LP>(s1) * (1.0f / H);
        s2 = grp_sum_lp<LP>(s2) * (1.0f / H);
#pragma unroll
        for (int i = 0; i < HC; ++i) dnp[i] += rstd * (dyg[i] - s1 - xh[i] * s2);
        rstore<H, LP>(c.resbuf + row * LH, dnp, sub);
    }
}

// ------------------------------------------------------------------------------------------
// attention stages on MFMA.  Head-group buffers (R x LQ each, a head = 80 columns [64 | 16 ext]):
//   R0 = Q_ext = [q | u]   (u = W_c,h^T q: the edge term of the logits, 3 of 16 columns used)
//   R1 = K_ext = [k | x]   R2 = V_ext = [v | x]   R3 = G_ext = [dE/do | r] (backward)
//   (forward: o_ext overwrites the row tile's own Q_ext rows in R0 once its logits are done)
// so that logits = Q_ext K_ext^T, o_ext = P V_ext = [o | sum_j a_ij x_j], da = G_ext V_ext^T, and the
// extension columns of dQ_ext / dK_ext / dV_ext are du / dE/dx_j / dE/dx_j: every N x N contraction is
// a 16x16x4 tile product.  A workgroup's G proteins share the tiles; pairs from different proteins are
// masked in the softmax (exact zeros in P, hence in dS).
// The phases are cooperative: tiles go round-robin over the 8 waves, barriers between phases.
// ------------------------------------------------------------------------------------------
// acc[jt] = A[rows of tile it] . B[rows of tile jt]^T over K = 80
template <int MT>
DEVI void co_dot_rows(f32x4 (&acc)[MT], const lfloat* A, const lfloat* B, int ld, int RN, int it, int lane) {
    const int kk = lane >> 4, mm = lane & 15;
    const lfloat* ap = A + min(16 * it + mm, RN - 1) * ld + 4 * kk;
    f32x4 av[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) av[kb] = *(const lf32x4*)(ap + 16 * kb);
    // extension block: only its first 4 columns are ever non-zero (u | s, x | |x|^2, r | g_D): one k-step, exact
    const float ax = ap[64 - 3 * kk];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        const lfloat* bp = B + min(16 * jt + mm, RN - 1) * ld + 4 * kk;
        f32x4 bv[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bv[kb] = *(const lf32x4*)(bp + 16 * kb);
        const float bx = bp[64 - 3 * kk];
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; kb += 2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][s], bv[kb][s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb + 1][s], bv[kb + 1][s], c1, 0, 0, 0);
            }
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, c0, 0, 0, 0);
        acc[jt] = c0 + c1;
    }
}

// one 16x16 output tile  C[m][n] = sum_k Aop[16 mo + m][k] B[k][n],  k over all 16 MT rows:
// TRANS = false: Aop[i][k] = T[i][k] ; true: Aop[i][k] = T[k][i]   (T = a head's (16MT x PL) tile array)
// B points at column 0 of the wanted 16-column slice of a head-group buffer (rows clamped to RN-1:
// the matching T entries are exact zeros).
// k-step (kt, s) covers k = 16 kt + 4 s .. + 3 (lane: + kk); T's rows and columns at or beyond the workgroup's real
// `rows` are exact zeros (P, dS), so the k-steps that start there are skipped.
// PL_: leading dimension of T.  The default (16 MT + 4) tile arrays have 16 MT rows; a TIGHT one (LdsLayout, PL_ < 16 MT) has the
// workgroup's RN allocated rows and no columns beyond PL_: operand reads are clamped to row RN - 1 (they feed output rows /
// k-steps that are discarded / skipped), and a column index beyond PL_ wraps into the next row (finite, discarded likewise).
template <int MT, bool TRANS, int PL_ = 16 * MT + 4>
DEVI f32x4 co_mm(const lfloat* T, int mo, const lfloat* B, int ldb, int RN, int rows, int lane) {
    constexpr int PL = PL_;
    constexpr bool TIGHT = PL_ < 16 * MT;
    const int kk = lane >> 4, mm = lane & 15;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    // (the operands of row tile kt + 1 are requested before the products of tile kt, as volatile reads: the compiler sinks
    // plain ones into the row-count branches that use them -- see co_mm5)
    typedef const volatile lfloat* vlp;
    float an[4], bn[4];
    auto load = [&](int kt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * kt + 4 * s + kk;
            an[s] = TRANS ? *(vlp)(T + (TIGHT ? min(k, RN - 1) : k) * PL + 16 * mo + mm)
                          : *(vlp)(T + (TIGHT ? min(16 * mo + mm, RN - 1) : 16 * mo + mm) * PL + k);
            if (TIGHT && TRANS && k >= RN) an[s] = 0.f;   // a row the tight array does not have: zero, as the wide array's pad rows
            bn[s] = *(vlp)(B + min(k, RN - 1) * ldb + mm);
        }
    };
    load(0);
#pragma unroll
    for (int kt = 0; kt < MT; ++kt) {
        float as[4], bs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { as[s] = an[s]; bs[s] = bn[s]; }
        if (kt + 1 < MT) load(kt + 1);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[0], bs[0], c0, 0, 0, 0);
        if (16 * kt + 4 < rows) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[1], bs[1], c1, 0, 0, 0);
        if (16 * kt + 8 < rows) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[2], bs[2], c0, 0, 0, 0);
        if (16 * kt + 12 < rows) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[3], bs[3], c1, 0, 0, 0);
    }
    return c0 + c1;
}

// the same for the five 16-column tiles of a head at once (B points at the head's column 0): the A
// operand is read once and five independent accumulator chains keep the MFMA pipe busy while the
// next operands arrive.
// hook(step), step = 0 .. 4 MT - 1, runs once per k-step: a place to spread another phase's weight requests over this one's
// MFMAs (a burst of them stalls the wave at issue while the CU's one texture-address unit takes 16 cycles per KiB).
struct NoStepHook { DEVI void operator()(int) const {} };
// SWAP: the two operands change places in the MFMA, which transposes the output tile: lane (quad, col) then holds row 16 mo + col,
// columns 16 nt + 4 quad .. + 3 -- four CONSECUTIVE columns of one row, so an epilogue writes 8- or 16-byte LDS words instead of four
// 2- or 4-byte ones (round 5: the dQ / dK / dV pieces of the fp16 engine).
template <int MT, bool TRANS, int NT, int PL_ = 16 * MT + 4, class SH = NoStepHook, bool SWAP = false>
DEVI void co_mmN(f32x4 (&c)[NT], const lfloat* T, int mo, const lfloat* B, int ldb, int RN, int rows, int lane, SH hook = SH()) {
    constexpr int PL = PL_, NS = 4 * MT;
    constexpr bool TIGHT = PL_ < 16 * MT;   // (see co_mm)
    const int kk = lane >> 4, mm = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) c[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // k-step st contracts k = 4 st + kk.  The steps beyond the workgroup's rows are skipped by a (uniform) branch, and the
    // compiler will not lift an LDS read over a branch: the operands of step st + 1 are therefore requested BEFORE the
    // products of step st -- otherwise every product waits out the LDS latency of its own operands (3 of 5 did: the
    // phases built on this ran at 40 % of their MFMA time).  Reads of skipped steps are harmless (clamped rows).
    // (volatile: a plain read that is only used inside the next step's branch gets sunk into it again)
    typedef const volatile lfloat* vlp;
    auto a_of = [&](int st) {
        const int k = 4 * st + kk;
        const float v = TRANS ? *(vlp)(T + (TIGHT ? min(k, RN - 1) : k) * PL + 16 * mo + mm)
                              : *(vlp)(T + (TIGHT ? min(16 * mo + mm, RN - 1) : 16 * mo + mm) * PL + k);
        // (TRANS: k is a ROW of the tile array; the tight one has no pad rows, and a k-step that straddles the last real row --
        // row counts that are not a multiple of 4 -- must see zeros there, as it does in the wide array)
        return (TIGHT && TRANS && k >= RN) ? 0.f : v;
    };
    auto b_of = [&](int st) { return (vlp)(B + min(4 * st + kk, RN - 1) * ldb + mm); };
    float a_n = a_of(0), b_n[NT];
    {
        vlp bp = b_of(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_n[nt] = bp[16 * nt];
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const float a_c = a_n;
        float b_c[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_c[nt] = b_n[nt];
        if (st + 1 < NS) {
            a_n = a_of(st + 1);
            vlp bp = b_of(st + 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b_n[nt] = bp[16 * nt];
        }
        if (st % 4 == 0 || 4 * st < rows) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                c[nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(b_c[nt], a_c, c[nt], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x4f32(a_c, b_c[nt], c[nt], 0, 0, 0);
        }
        hook(st);
    }
}
template <int MT, bool TRANS, int PL_ = 16 * MT + 4, class SH = NoStepHook, bool SWAP = false>
DEVI void co_mm5(f32x4 (&c)[5], const lfloat* T, int mo, const lfloat* B, int ldb, int RN, int rows, int lane, SH hook = SH()) {
    co_mmN<MT, TRANS, 5, PL_, SH, SWAP>(c, T, mo, B, ldb, RN, rows, lane, hook);
}


// GEN variants (other input branches, oracle/kernel_model_gen.py): the 16-column head extension carries
//   Q_ext: [u - 2 s x_i (3) | s]    K_ext, V_ext: [x_j (3) | |x_j|^2]    o_ext: [xrel (3) | D]
//   G_ext: [r - 2 gD x_i (3) | gD]  with s = c_h.q (distance term of the logits), D = sum_j a_ij |x_i - x_j|^2.
// The GEMMs produce the x-independent parts ([u | s], [r | gD]); the x_i-dependent fix-ups are applied in LDS
// by the wave that owns the row tile, right before it uses them.
struct CoGeo {
    lfloat *Rg, *Pbuf, *dSbuf, *xs, *dxw;   // dxw: this wave's partial dE/dx
    lu32* lsp;                              // l pieces of dV (LdsLayout::lsplit)
    lu32* lsq;                              // l pieces of dQ (behind them)
    lfloat* m12;                            // GEN: [m1 | m2] rows of the head group (backward)
    const int __attribute__((address_space(3))) * prow;   // protein index of each row, -1 for pad rows
    int N, RN, rows;
    float qs, qsi;                          // fp16 engine: power-of-two scale of the dQ / dK / dV pieces of this layer (and its inverse)
};
// dQ / dK / dV (the QKV_ext^T GEMM's A operand) mix the rows of dattn, so they share ONE scale per workgroup and layer: the
// smallest of the row scales rowb_ln2_gate1 left in rsc[] (as inverses), times 2^-8 -- sums over up to 64 rows and the factors
// of q, k and P stay far from fp16's 65504, and with subnormals kept the pieces still resolve 3e-11 of the scaled values.
DEVI void block_pow2_scale(const lfloat* rsc, float& qs, float& qsi) {
    const int lane = tid_now() & 63;
    float mx = rsc[lane];
    mx = row16_max(mx);
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    int ex = (int)((__float_as_uint(mx) >> 23) & 255u);
    ex = ex < 20 ? 20 : (ex > 230 ? 230 : ex);
    qs = __uint_as_float((unsigned)(246 - ex) << 23);    // 2^-(ex - 127) 2^-8
    qsi = __uint_as_float((unsigned)(ex + 8) << 23);     // 2^(ex - 127) 2^8
}
// one element of dQ / dK / dV as the pieces the back-projection multiplies, in place of its fp32 row: [h | m] + l elsewhere (bf16), or
// -- fp16 engine -- [h | l'] of the scaled value and nothing elsewhere
DEVI void put_piece16(lu16* hm, float v, float qs) {
    unsigned short hh, ll;
    split1h(v * qs, hh, ll);
    hm[0] = hh; hm[64] = ll;
}
// ... four consecutive elements of a row at once (the transposed tiles of co_mmN<..., SWAP>): two 8-byte stores; hm2 = the dword that
// holds the first element's h piece (the l' pieces sit 64 elements = 32 dwords further on)
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
DEVI void put_piece16x4(lu32* hm2, const f32x4& v, float qs) {
    unsigned h0, l0, h1, l1;
    split2h(v[0] * qs, v[1] * qs, h0, l0);
    split2h(v[2] * qs, v[3] * qs, h1, l1);
    typedef u32x2_ __attribute__((address_space(3))) lu32x2;
    *(lu32x2*)hm2 = (u32x2_){h0, h1};
    *(lu32x2*)(hm2 + 32) = (u32x2_){l0, l1};
}

// K_ext / V_ext extension columns <- x (columns 3..15 zero)
template <int HGS, bool GEN>
DEVI void co_fill_x(const CoGeo& g) {
    const int tid_ = tid_now();
    constexpr int LQ = 80 * HGS + 4;
    const int total = g.rows * HGS * 16;
    for (int it = tid_; it < total; it += DFF_NTHREADS) {
        const int cc = it & 15, r2 = it >> 4;
        const int hh = r2 % HGS, row = r2 / HGS;
        float v = cc < 3 ? g.xs[row * 4 + cc] : 0.f;
        if (GEN && cc == 3) {
            const float x0 = g.xs[row * 4], x1 = g.xs[row * 4 + 1], x2 = g.xs[row * 4 + 2];
            v = x0 * x0 + x1 * x1 + x2 * x2;
        }
        lfloat* d = g.Rg + g.RN * LQ + row * LQ + hh * 80 + 64 + cc;
        d[0] = v;
        d[g.RN * LQ] = v;
    }
}

// GEN fix-ups of one row tile (lanes 0..15 take one row each)
template <int HGS>
DEVI void co_fix_q(const CoGeo& g, int hh, int it, int lane) {   // Q_ext ext: [u | s] -> [u - 2 s x_i | s]
    constexpr int LQ = 80 * HGS + 4;
    const int row = 16 * it + lane;
    if (lane < 16 && row < g.rows) {
        lfloat* q = g.Rg + row * LQ + hh * 80 + 64;
        const float sv = q[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) q[c3] = q[c3] - 2.0f * sv * g.xs[row * 4 + c3];
    }
}
template <int HGS>
DEVI void co_fix_g(const CoGeo& g, int hh, int it, int lane) {   // G_ext ext: [r | gD] -> [r - 2 gD x_i | gD]; dE/dx_i += gD (2 x_i - 2 m1_i)
    constexpr int LQ = 80 * HGS + 4;
    const int row = 16 * it + lane;
    if (lane < 16 && row < g.rows) {
        lfloat* gp = g.Rg + 3 * g.RN * LQ + row * LQ + hh * 80 + 64;
        const lfloat* mp = g.m12 + (hh * g.RN + row) * 4;
        const float gD = gp[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            const float xc = g.xs[row * 4 + c3];
            g.dxw[row * 4 + c3] += gD * (2.0f * xc - 2.0f * mp[c3]);
            gp[c3] = gp[c3] - 2.0f * gD * xc;
        }
    }
}

// logits + softmax:  a_ihj = softmax_j( scale (q_ih.k_jh + u_ih.x_j) ) over the beads j of i's own
// protein (graph_transformer.py:247-255 with the j-constant terms dropped) -> Pbuf (+ stash); then, by
// the same wave (its 16 probability rows are all it needs, so no barrier):
// o_ext = P V_ext -> R0 rows of this tile (Q_ext rows of the tile are dead after its logits);
// extension tile: xrel_i = sum_j a_ij x_j - x_i
// SPW: the 64 regular columns of o_ext are written as bf16 pieces ([piece][RN][64 HGS + 8], into R3 | R4, free in the
// forward pass) for gemm_tall_split; the extension columns stay fp32 in R0.
template <int MT, int HGS, bool GEN, bool SPW = false, int PL_ = 16 * MT + 4>
DEVI void co_softmax_pv(const CoGeo& g, gfloat* sP /* P block of head hg*HGS */, gfloat* sM /* m12 block of head hg*HGS (GEN) */,
                        lfloat* oxt = nullptr /* != null: the extension columns of o_ext go here (rows x 16 HGS) instead of R0 */) {
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int LQ = 80 * HGS + 4, PL = PL_, PT = 16 * MT * PL, PS = 16 * MT;
    constexpr bool TIGHT = PL_ < 16 * MT;   // P tile array of RN rows x PL columns (LdsLayout): no pad rows, no columns >= PL
    static_assert(!TIGHT || HGS == 1, "tight tile arrays: one head per group");
    const int quad = lane >> 4, col = lane & 15;
    int gj[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) gj[jt] = g.prow[16 * jt + col];
    for (int item = wave; item < HGS * MT; item += DFF_NWAVES) {
        const int hh = item / MT, it = item - hh * MT;
        if (GEN) co_fix_q<HGS>(g, hh, it, lane);
        {
            f32x4 acc[MT];
            co_dot_rows<MT>(acc, g.Rg + hh * 80, g.Rg + g.RN * LQ + hh * 80, LQ, g.RN, it, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * quad + r;
                const int gi0 = g.prow[i], gi = gi0 >= 0 ? gi0 : -2;   // a pad row matches nothing
                float s[MT], mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < MT; ++jt) {
                    s[jt] = gj[jt] == gi ? acc[jt][r] * 0.125f : -INFINITY;
                    mx = fmaxf(mx, s[jt]);
                }
                mx = row16_max(mx);
                float e[MT], den = 0.f;
#pragma unroll
                for (int jt = 0; jt < MT; ++jt) {
                    e[jt] = gj[jt] == gi ? fast_exp(s[jt] - mx) : 0.f;
                    den += e[jt];
                }
                den = row16_sum(den);
                lfloat* pl = g.Pbuf + hh * PT + i * PL + col;
                gfloat* ps = sP + ((size_t)hh * g.RN + min(i, g.RN - 1)) * PS + col;
                const float rden = fast_rcp(den);
#pragma unroll
                for (int jt = 0; jt < MT; ++jt) {
                    const float p = gi >= 0 ? e[jt] * rden : 0.f;
                    if (!TIGHT || (i < g.RN && 16 * jt + col < PL)) pl[16 * jt] = p;
                    if (gi >= 0) st_ntg<DFF_SITE_ST(MT, 4)>(ps + 16 * jt, p);
                }
            }
        }
        f32x4 o[5];
        co_mm5<MT, false, PL>(o, g.Pbuf + hh * PT, it, g.Rg + 2 * g.RN * LQ + hh * 80, LQ, g.RN, g.rows, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * it + 4 * quad + r;
            if (row < g.rows) {
                lfloat* d = g.Rg + row * LQ + hh * 80 + col;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if constexpr (SPW) store_split<(DFF_F16G & 1) != 0>((lu16*)(g.Rg + 3 * g.RN * LQ), g.RN, 64 * HGS + DFF_SPAD, row, hh * 64 + 16 * nt + col, o[nt][r]);
                    else d[16 * nt] = o[nt][r];
                }
                if (!GEN) {
                    const float xr = o[4][r] - g.xs[row * 4 + min(col, 3)];
                    if (oxt) oxt[row * (16 * HGS) + hh * 16 + col] = xr;
                    else d[64] = xr;
                }
            }
            if (GEN) {
                // extension tile [m1 (3) | m2]: xrel = m1 - x_i ; D = |x_i|^2 - 2 x_i.m1 + m2  (columns 0..3 = one DPP quad)
                const int rw = min(row, g.rows - 1);
                const float x0 = g.xs[rw * 4], x1 = g.xs[rw * 4 + 1], x2 = g.xs[rw * 4 + 2];
                const float xr = col == 0 ? x0 : col == 1 ? x1 : col == 2 ? x2 : 0.f;
                const float e = o[4][r];
                const float tq = col < 3 ? -2.0f * xr * e : col == 3 ? e + (x0 * x0 + x1 * x1 + x2 * x2) : 0.f;
                const float D = quad_sum(tq);
                if (row < g.rows) {
                    g.Rg[row * LQ + hh * 80 + 64 + col] = col < 3 ? e - xr : col == 3 ? D : 0.f;
                    if (col < 4) st_ntg<DFF_SITE_ST(MT, 4)>(sM + ((size_t)hh * g.RN + row) * 4 + col, e);
                }
            }
        }
    }
}

// The same stage for the shipped input branch (!GEN), with both products issued transposed so that nothing has to be
// re-laid-out between them.  Logits: S^T tiles with the K_ext rows permuted inside a tile (tile row m <-> bead
// j = 16 jt + 4 (m % 4) + m / 4), so lane (i = lane & 15, quad) holds s[i][16 jt + 4 r + quad], r = 0..3, jt < MT: the whole
// logit row of bead i sits in the four lanes i, i + 16, i + 32, i + 48 -- softmax = 4 MT in-lane steps + two cross-quad
// shuffles (the C-layout version needs eight DPP steps for each of a lane's four rows).  P V_ext: o^T tiles, whose
// second operand (row i, k = quad) at k-step (kt, r) is P[i][16 kt + 4 r + quad] -- exactly the register the lane already
// holds; the k-steps stay contiguous in j, so the ones beyond the workgroup's rows are still skipped.  P never touches LDS
// in the forward pass (the backward pass reloads it from the stash), and a lane ends up with 4 consecutive columns of o
// for its row: 16-byte LDS writes / 8-byte piece writes.  All stash stores are unconditional (junk slot for pad rows).
DEVI float xquad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
DEVI float xquad_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
template <int MT, int HGS, bool SPW>
DEVI void co_softmax_pv_t(const CoGeo& g, gfloat* sP /* P block of head hg*HGS */, lfloat* oxt, gfloat* junk) {
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int LQ = 80 * HGS + 4, PS = 16 * MT;
    const int quad = lane >> 4, rl = lane & 15;
    const int jperm = 4 * (rl & 3) + (rl >> 2);   // K_ext row of tile row rl
    for (int item = wave; item < HGS * MT; item += DFF_NWAVES) {
        const int hh = item / MT, it = item - hh * MT;
        const int i = 16 * it + rl;
        const int gi0 = g.prow[i], gi = gi0 >= 0 ? gi0 : -2;   // a pad row matches nothing
        f32x4 p[MT];
        {
            // S^T: first operand = K_ext rows (permuted), second = Q_ext rows of this tile
            const lfloat* qp = g.Rg + hh * 80 + min(i, g.RN - 1) * LQ + 4 * quad;
            f32x4 qv[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) qv[kb] = *(const lf32x4*)(qp + 16 * kb);
            const float qx = qp[64 - 3 * quad];
#pragma unroll
            for (int jt = 0; jt < MT; ++jt) {
                const lfloat* kp = g.Rg + g.RN * LQ + hh * 80 + min(16 * jt + jperm, g.RN - 1) * LQ + 4 * quad;
                f32x4 kv[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) kv[kb] = *(const lf32x4*)(kp + 16 * kb);
                const float kx = kp[64 - 3 * quad];
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; kb += 2)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[kb][s4], qv[kb][s4], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[kb + 1][s4], qv[kb + 1][s4], c1, 0, 0, 0);
                    }
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kx, qx, c0, 0, 0, 0);
                p[jt] = c0 + c1;
            }
        }
        // masked softmax of row i over j = 16 jt + 4 r + quad
        bool ok[MT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ok[jt][r] = g.prow[16 * jt + 4 * r + quad] == gi;
                p[jt][r] = ok[jt][r] ? p[jt][r] * 0.125f : -INFINITY;
                mx = fmaxf(mx, p[jt][r]);
            }
        mx = xquad_max(mx);
        float den = 0.f;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[jt][r] = ok[jt][r] ? fast_exp(p[jt][r] - mx) : 0.f;
                den += p[jt][r];
            }
        den = xquad_sum(den);
        const float rden = gi >= 0 ? fast_rcp(den) : 0.f;
        // The stash row of bead i: a lane holds every fourth probability (j = 16 jt + 4 r + quad), so writing them as they lie is 4 MT
        // dword stores per lane, each instruction scattering 64 dwords over 16 rows (measured: 1.5 - 1.7 k cycles of a head's 9.6 k on
        // the softmax waves, which are the critical path of the head pipeline).  A 4 x 4 transpose across the four lanes of a row --
        // v_permlane32_swap, then v_permlane16_swap: four instructions per tile -- leaves lane (i, quad) with j = 16 jt + 4 quad .. + 3:
        // one 16-byte store per tile, 64 contiguous bytes per row.  (P V_ext below keeps using the untransposed registers.)
        gfloat* const ps = gi >= 0 ? sP + ((size_t)hh * g.RN + i) * PS + 4 * quad : junk;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            p[jt] *= rden;
            const auto a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[jt][0]), __float_as_uint(p[jt][2]), false, false);
            const auto a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[jt][1]), __float_as_uint(p[jt][3]), false, false);
            const auto c01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
            const auto c23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
            const f32x4 row4 = {__uint_as_float(c01[0]), __uint_as_float(c01[1]), __uint_as_float(c23[0]), __uint_as_float(c23[1])};
            st_ntg4<DFF_SITE_ST(MT, 4)>(gi >= 0 ? ps + 16 * jt : junk, row4);
        }
        // o^T = (P V_ext)^T: first operand = V_ext (k = bead, column n), second = P from the registers
        f32x4 o[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const lfloat* const vb = g.Rg + 2 * g.RN * LQ + hh * 80 + rl;
        // (operands of k-step st + 1 requested before the products of step st: see co_mm5)
        float v_n[5];
        typedef const volatile lfloat* vlp;   // (volatile: see co_mm5)
        {
            vlp vp = vb + min(quad, g.RN - 1) * LQ;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) v_n[nt] = vp[16 * nt];
        }
#pragma unroll
        for (int st = 0; st < 4 * MT; ++st) {
            float v_c[5];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) v_c[nt] = v_n[nt];
            if (st + 1 < 4 * MT) {
                vlp vp = vb + min(4 * (st + 1) + quad, g.RN - 1) * LQ;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) v_n[nt] = vp[16 * nt];
            }
            if (st % 4 == 0 || 4 * st < g.rows) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_c[nt], p[st / 4][st % 4], o[nt], 0, 0, 0);
            }
        }
        if (i < g.rows) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if constexpr (SPW) store_split4<(DFF_F16G & 1) != 0>((lu32*)(g.Rg + 3 * g.RN * LQ), g.RN, (64 * HGS + DFF_SPAD) / 2, i, hh * 64 + 16 * nt + 4 * quad, o[nt]);
                else *(lf32x4*)(g.Rg + i * LQ + hh * 80 + 16 * nt + 4 * quad) = o[nt];
            }
            // extension tile: xrel_i = sum_j a_ij x_j - x_i (columns 0..2; the others as the C-layout version leaves them)
            const float x3 = g.xs[i * 4 + 3];
            f32x4 xr = o[4] - (f32x4){x3, x3, x3, x3};
            if (quad == 0) xr = o[4] - *(const lf32x4*)(g.xs + i * 4);
            if (oxt) *(lf32x4*)(oxt + i * (16 * HGS) + hh * 16 + 4 * quad) = xr;
            else *(lf32x4*)(g.Rg + i * LQ + hh * 80 + 64 + 4 * quad) = xr;
        }
    }
}

// GEN: extension tile of dQ_ext = dS K_ext holds [A (3) | B] = [sum_j dS x_j | sum_j dS |x_j|^2] in the columns
// 0..3 of a row: du = A ; ds = -2 x_i.A + B ; dE/dx_i += -2 s_i A.  Returns the value to store.
template <int HGS>
DEVI float co_dq_ext(const CoGeo& g, int hh, int row, int col, float e) {
    constexpr int LQ = 80 * HGS + 4;
    const int rw = min(row, g.rows - 1);
    const float xr = col < 3 ? g.xs[rw * 4 + col] : 0.f;
    const float tq = col < 3 ? -2.0f * xr * e : col == 3 ? e : 0.f;
    const float dsv = quad_sum(tq);
    if (row < g.rows && col < 3) g.dxw[row * 4 + col] += -2.0f * g.Rg[row * LQ + hh * 80 + 67] * e;
    return col == 3 ? dsv : e;
}
// GEN: extension tile of dV_ext / dK_ext holds [c (3) | w]: dE/dx_j += c + 2 x_j w
DEVI float co_dx_ext(const CoGeo& g, int row, int col, float e) {
    const float w = quad_bcast3(e);
    const int rw = min(row, g.rows - 1);
    return e + 2.0f * (col < 3 ? g.xs[rw * 4 + col] : 0.f) * w;
}

// backward: da = G_ext V_ext^T ; ds = scale a (da - sum_j a da) -> dSbuf.
// DQ: the same wave goes on with dQ_ext = dS K_ext for its row tile -> buffer 4 (no barrier needed: it
// only reads the dS rows it has just written).
template <int MT, int HGS, bool DQ, bool GEN, int PL_ = 16 * MT + 4, bool QSP = false>
DEVI void co_ds(const CoGeo& g) {
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int LQ = 80 * HGS + 4, PL = PL_, PT = 16 * MT * PL;
    constexpr bool TIGHT = PL_ < 16 * MT;   // (see co_softmax_pv)
    static_assert(!TIGHT || (HGS == 1 && !DQ), "tight tile arrays: one head per group, three-phase backward");
    const int quad = lane >> 4, col = lane & 15;
    for (int item = wave; item < HGS * MT; item += DFF_NWAVES) {
        const int hh = item / MT, it = item - hh * MT;
        if (GEN) { co_fix_q<HGS>(g, hh, it, lane); co_fix_g<HGS>(g, hh, it, lane); }
        {
            f32x4 acc[MT];
            co_dot_rows<MT>(acc, g.Rg + 3 * g.RN * LQ + hh * 80, g.Rg + 2 * g.RN * LQ + hh * 80, LQ, g.RN, it, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * quad + r;
                const lfloat* pl = g.Pbuf + hh * PT + (TIGHT ? min(i, g.RN - 1) : i) * PL + col;
                float p[MT], sm = 0.f;
#pragma unroll
                for (int jt = 0; jt < MT; ++jt) {
                    p[jt] = (!TIGHT || 16 * jt + col < PL) ? pl[16 * jt] : 0.f;
                    sm += p[jt] * acc[jt][r];
                }
                sm = row16_sum(sm);
                lfloat* dl = g.dSbuf + hh * PT + i * PL + col;
#pragma unroll
                for (int jt = 0; jt < MT; ++jt)
                    if (!TIGHT || (i < g.RN && 16 * jt + col < PL)) dl[16 * jt] = 0.125f * (p[jt] * (acc[jt][r] - sm));
            }
        }
        if constexpr (DQ && QSP && DFF_QT16 && !GEN) {
            // transposed tiles: this lane holds row 16 it + col, columns 16 nt + 4 quad .. + 3 of dQ_ext
            f32x4 dq[5];
            co_mm5<MT, false, 16 * MT + 4, NoStepHook, true>(dq, g.dSbuf + hh * PT, it, g.Rg + g.RN * LQ + hh * 80, LQ, g.RN, g.rows, lane);
            const int row = 16 * it + col;
            if (row < g.rows) {
                lfloat* const d = g.Rg + 4 * g.RN * LQ + row * LQ + hh * 80;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) put_piece16x4((lu32*)d + 8 * nt + 2 * quad, dq[nt], g.qs);
                *(lf32x4*)(d + 64 + 4 * quad) = dq[4];
            }
        } else
        if (DQ) {
            f32x4 dq[5];
            co_mm5<MT, false>(dq, g.dSbuf + hh * PT, it, g.Rg + g.RN * LQ + hh * 80, LQ, g.RN, g.rows, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * it + 4 * quad + r;
                if (GEN) dq[4][r] = co_dq_ext<HGS>(g, hh, row, col, dq[4][r]);
                if (row < g.rows) {
                    lfloat* d = g.Rg + 4 * g.RN * LQ + row * LQ + hh * 80 + col;
                    if constexpr (QSP) {
                        // the 64 regular columns leave as the bf16 pieces the back-projection multiplies (see co_dv_dk): split once
                        // here instead of by each of the eight waves that read them
                        constexpr int LSV = 32 * HGS + 4;
                        lu16* const hm = (lu16*)(g.Rg + 4 * g.RN * LQ + hh * 80) + col;
                        lu16* const lb0 = (lu16*)(g.lsq + hh * 32) + col;
                        lu16* const lb1 = (lu16*)(g.lsq + hh * 32 + 16) + col;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const float v = dq[nt][r];
                            if constexpr (DFF_QT16) { put_piece16(hm + row * 2 * LQ + 16 * nt, v, g.qs); continue; }
                            const unsigned uh = __float_as_uint(v) & 0xffff0000u;
                            const float r1 = v - __uint_as_float(uh);
                            const unsigned um = __float_as_uint(r1) & 0xffff0000u;
                            const float r2 = r1 - __uint_as_float(um);
                            hm[row * 2 * LQ + 16 * nt] = (unsigned short)(uh >> 16);
                            hm[row * 2 * LQ + 64 + 16 * nt] = (unsigned short)(um >> 16);
                            (nt < 2 ? lb0 : lb1)[row * 2 * LSV + 16 * (nt & 1)] = (unsigned short)(__float_as_uint(r2) >> 16);
                        }
                        d[64] = dq[4][r];
                    } else {
#pragma unroll
                    for (int nt = 0; nt < 5; ++nt) d[16 * nt] = dq[nt][r];
                    }
                }
            }
        }
    }
}

template <int MT, int HGS, bool EXT_ONLY, bool GEN, bool KVS = false, bool VSP = false, class SH = NoStepHook, int PL_ = 16 * MT + 4>
DEVI void co_dv_dk(const CoGeo& g, SH hook = SH()) {
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int LQ = 80 * HGS + 4, PL = PL_, PT = 16 * MT * PL;
    static_assert(PL_ == 16 * MT + 4 || EXT_ONLY, "tight tile arrays: three-phase backward (co_dqkv), only the layer-0 form runs here");
    const int quad = lane >> 4, col = lane & 15;
    if (EXT_ONLY) {
        for (int item = wave; item < 2 * HGS * MT; item += DFF_NWAVES) {
            const int which = item / (HGS * MT), r0 = item - which * (HGS * MT);   // 0: dV, 1: dK
            const int hh = r0 / MT, mo = r0 - hh * MT;
            const lfloat* T = (which ? g.dSbuf : g.Pbuf) + hh * PT;
            const f32x4 acc = co_mm<MT, true, PL>(T, mo, g.Rg + (which ? 0 : 3) * g.RN * LQ + hh * 80 + 64, LQ, g.RN, g.rows, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mo + 4 * quad + r;
                const float e = GEN ? co_dx_ext(g, row, col, acc[r]) : acc[r];
                if (row < g.rows && col < 3) g.dxw[row * 4 + col] += e;
            }
        }
        if (GEN) {   // the logits' distance term reaches x_i through Q_ext as well: dE/dx_i += -2 s_i sum_j dS_ij x_j
            for (int item = wave; item < HGS * MT; item += DFF_NWAVES) {
                const int hh = item / MT, it = item - hh * MT;
                const f32x4 acc = co_mm<MT, false, PL>(g.dSbuf + hh * PT, it, g.Rg + g.RN * LQ + hh * 80 + 64, LQ, g.RN, g.rows, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) (void)co_dq_ext<HGS>(g, hh, 16 * it + 4 * quad + r, col, acc[r]);
            }
        }
    } else {
        for (int item = wave; item < 2 * HGS * MT; item += DFF_NWAVES) {
            const int which = item / (HGS * MT), r0 = item - which * (HGS * MT);
            const int hh = r0 / MT, mo = r0 - hh * MT;
            const lfloat* T = (which ? g.dSbuf : g.Pbuf) + hh * PT;
            f32x4 acc[5];
            if constexpr (KVS && VSP && DFF_QT16 && !GEN) {
                // transposed tiles (see co_ds): row 16 mo + col, columns 16 nt + 4 quad .. + 3; the extension tile's columns 0..2 = quad 0
                co_mm5<MT, true, 16 * MT + 4, SH, true>(acc, T, mo, g.Rg + (which ? 0 : 3) * g.RN * LQ + hh * 80, LQ, g.RN, g.rows, lane, hook);
                const int row = 16 * mo + col;
                if (row < g.rows) {
                    lu32* const d2 = (lu32*)(g.Rg + (which ? 1 : 2) * g.RN * LQ + row * LQ + hh * 80);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) put_piece16x4(d2 + 8 * nt + 2 * quad, acc[nt], g.qs);
                    if (quad == 0) {
                        g.dxw[row * 4 + 0] += acc[4][0];
                        g.dxw[row * 4 + 1] += acc[4][1];
                        g.dxw[row * 4 + 2] += acc[4][2];
                    }
                }
                continue;
            }
            co_mm5<MT, true>(acc, T, mo, g.Rg + (which ? 0 : 3) * g.RN * LQ + hh * 80, LQ, g.RN, g.rows, lane, hook);
            lfloat* const dst = g.Rg + (which ? 1 : 2) * g.RN * LQ + hh * 80 + col;
            lu16* const hm = (lu16*)(g.Rg + (which ? 1 : 2) * g.RN * LQ + hh * 80) + col;
            constexpr int LSV = 32 * HGS + 4;
            // l pieces: columns 0..31 from lb0, 32..63 from lb1, row stride ls (16-bit units)
            lu16* const lb0 = (which ? (lu16*)(g.Rg + 1 * g.RN * LQ + hh * 80 + 64) : (lu16*)(g.lsp + hh * 32)) + col;
            lu16* const lb1 = (which ? (lu16*)(g.Rg + 2 * g.RN * LQ + hh * 80 + 64) : (lu16*)(g.lsp + hh * 32 + 16)) + col;
            const int ls = which ? 2 * LQ : 2 * LSV;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mo + 4 * quad + r;
                if (row < g.rows) {
                    if (KVS && (VSP || which)) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const float v = acc[nt][r];
                            if constexpr (DFF_QT16) { put_piece16(hm + row * 2 * LQ + 16 * nt, v, g.qs); continue; }
                            const unsigned uh = __float_as_uint(v) & 0xffff0000u;
                            const float r1 = v - __uint_as_float(uh);
                            const unsigned um = __float_as_uint(r1) & 0xffff0000u;
                            const float r2 = r1 - __uint_as_float(um);
                            hm[row * 2 * LQ + 16 * nt] = (unsigned short)(uh >> 16);
                            hm[row * 2 * LQ + 64 + 16 * nt] = (unsigned short)(um >> 16);
                            (nt < 2 ? lb0 : lb1)[row * ls + 16 * (nt & 1)] = (unsigned short)(__float_as_uint(r2) >> 16);
                        }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) dst[row * LQ + 16 * nt] = acc[nt][r];
                    }
                }
                const float e = GEN ? co_dx_ext(g, row, col, acc[4][r]) : acc[4][r];
                if (row < g.rows && col < 3) g.dxw[row * 4 + col] += e;
            }
        }
    }
}

// backward tile products without the fifth buffer (4 row tiles: LDS is full), one phase each:
// WHICH 0: dV_ext = P^T G_ext      (reads Pbuf, R3)  -> R2
// WHICH 1: dQ_ext = dS K_ext       (reads dSbuf, R1) -> R3
// WHICH 2: dK_ext = dS^T Q_ext     (reads dSbuf, R0) -> R1
// the extension tiles of dV_ext / dK_ext are dE/dx_j and go to this wave's dxw instead.
template <int MT, int HGS, int WHICH, bool GEN, int PL_ = 16 * MT + 4>
DEVI void co_dqkv(const CoGeo& g) {
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int LQ = 80 * HGS + 4, PL = PL_, PT = 16 * MT * PL;
    constexpr int SRC = WHICH == 0 ? 3 : WHICH == 1 ? 1 : 0;
    constexpr int DST = WHICH == 0 ? 2 : WHICH == 1 ? 3 : 1;
    const int quad = lane >> 4, col = lane & 15;
    const lfloat* T = WHICH == 0 ? g.Pbuf : g.dSbuf;
    for (int item = wave; item < HGS * MT * 5; item += DFF_NWAVES) {
        const int hh = item / (MT * 5), rem = item - hh * (MT * 5);
        const int mo = rem / 5, nt = rem - mo * 5;
        const f32x4 acc = co_mm<MT, WHICH != 1, PL>(T + hh * PT, mo, g.Rg + SRC * g.RN * LQ + hh * 80 + 16 * nt, LQ, g.RN, g.rows, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mo + 4 * quad + r;
            float e = acc[r];
            if (GEN && nt == 4) e = WHICH == 1 ? co_dq_ext<HGS>(g, hh, row, col, e) : co_dx_ext(g, row, col, e);
            if (row < g.rows) {
                if (WHICH != 1 && nt == 4) {
                    if (col < 3) g.dxw[row * 4 + col] += e;
                } else {
                    g.Rg[DST * g.RN * LQ + row * LQ + hh * 80 + 16 * nt + col] = e;
                }
            }
        }
    }
}

// The same three phases for exactly four row tiles on eight waves, shipped input branch (protein G, round 4): instead of 20
// single-tile items over 8 waves (three rounds of LDS-latency-bound 16 x 16 products: 24 scalar reads for 14 k-steps), wave w
// takes row tile w & 3 and runs its column tiles {0, 1, 2} (w < 4) or {3, 4} (w >= 4) side by side off ONE read of the
// tile-array operand: the two waves of a SIMD share a row tile's five products.
// PSPLIT (split engine): dQ and dK leave as the bf16 pieces the back-projection multiplies ([h | m] in place of the fp32 row; dQ's
// l pieces in g.lsq = the P tile array, dead after the dV phase; dK's in the extension columns of the R1 / R2 rows, as co_dv_dk
// does); dV, whose phase has no dead buffer to put l pieces in, stays fp32 and is split by the back-projection's waves.
template <int MT, int WHICH, int PL_, bool PSPLIT = false>
DEVI void co_dqkv_rows(const CoGeo& g) {
    static_assert(MT == 4 && DFF_NWAVES == 8, "two waves per row tile");
    constexpr int LQ = 84;
    constexpr int SRC = WHICH == 0 ? 3 : WHICH == 1 ? 1 : 0;
    constexpr int DST = WHICH == 0 ? 2 : WHICH == 1 ? 3 : 1;
    const int tid_ = tid_now(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int quad = lane >> 4, col = lane & 15, mo = wave & (MT - 1);
    const lfloat* const T = WHICH == 0 ? g.Pbuf : g.dSbuf;
    constexpr int LSV = 36;   // dwords per row of dQ's l pieces (LdsLayout::LSV at one head per group)
    auto put = [&](int nt, const f32x4& acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mo + 4 * quad + r;
            if (row < g.rows) {
                if (WHICH != 1 && nt == 4) { if (col < 3) g.dxw[row * 4 + col] += acc[r]; }
                else if (PSPLIT && WHICH != 0 && nt < 4) {
                    lu16* const hm = (lu16*)(g.Rg + DST * g.RN * LQ) + col;
                    lu16* const lb = (WHICH == 1 ? (lu16*)(g.lsq + (nt < 2 ? 0 : 16)) : (lu16*)(g.Rg + (nt < 2 ? 1 : 2) * g.RN * LQ + 64)) + col;
                    const int ls = WHICH == 1 ? 2 * LSV : 2 * LQ;
                    const float v = acc[r];
                    if constexpr (DFF_QT16) put_piece16(hm + row * 2 * LQ + 16 * nt, v, g.qs);
                    else {
                    const unsigned uh = __float_as_uint(v) & 0xffff0000u;
                    const float r1 = v - __uint_as_float(uh);
                    const unsigned um = __float_as_uint(r1) & 0xffff0000u;
                    const float r2 = r1 - __uint_as_float(um);
                    hm[row * 2 * LQ + 16 * nt] = (unsigned short)(uh >> 16);
                    hm[row * 2 * LQ + 64 + 16 * nt] = (unsigned short)(um >> 16);
                    lb[row * ls + 16 * (nt & 1)] = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                } else g.Rg[DST * g.RN * LQ + row * LQ + 16 * nt + col] = acc[r];
            }
        }
    };
    // fp16 engine: dQ / dK as transposed tiles (co_mmN<..., SWAP>): this lane holds row 16 mo + col, columns 16 nt + 4 quad .. + 3
    constexpr bool SW = PSPLIT && WHICH != 0 && DFF_QT16;
    auto put_t = [&](int nt, const f32x4& acc) {
        const int row = 16 * mo + col;
        if (row >= g.rows) return;
        lfloat* const d = g.Rg + DST * g.RN * LQ + row * LQ;
        if (nt < 4) put_piece16x4((lu32*)d + 8 * nt + 2 * quad, acc, g.qs);
        else if (WHICH == 1) *(lf32x4*)(d + 64 + 4 * quad) = acc;
        else if (quad == 0) { g.dxw[row * 4 + 0] += acc[0]; g.dxw[row * 4 + 1] += acc[1]; g.dxw[row * 4 + 2] += acc[2]; }
    };
    if (wave < MT) {
        f32x4 c[3];
        co_mmN<MT, WHICH != 1, 3, PL_, NoStepHook, SW>(c, T, mo, g.Rg + SRC * g.RN * LQ, LQ, g.RN, g.rows, lane);
        if constexpr (SW) { put_t(0, c[0]); put_t(1, c[1]); put_t(2, c[2]); }
        else { put(0, c[0]); put(1, c[1]); put(2, c[2]); }
    } else {
        f32x4 c[2];
        co_mmN<MT, WHICH != 1, 2, PL_, NoStepHook, SW>(c, T, mo, g.Rg + SRC * g.RN * LQ + 48, LQ, g.RN, g.rows, lane);
        if constexpr (SW) { put_t(3, c[0]); put_t(4, c[1]); }
        else { put(3, c[0]); put(4, c[1]); }
    }
}

// reload Q_ext, K, V (-> R0, R1, R2) and optionally P (-> Pbuf) of layer l / head group hg from the stash,
// split in two so that the loads fly while something else (the G_ext GEMM) runs:
// issue: global -> registers ; commit: registers -> LDS.
template <int MT, int HGS>
struct CoReload {
    static constexpr int NQ = (DFF_QKVW / 4) * HGS;          // float4 per row of the q|k|v blocks
    static constexpr int NP = 4 * MT * HGS;                  // float4 per row of the P blocks
    static constexpr int U = (16 * MT * (NQ + NP) + DFF_NTHREADS - 1) / DFF_NTHREADS;
    f32x4 t[U];
    // per item: [0,14) source offset (float4 units, from the head group's q|k|v or P block), [14,28) LDS offset
    // (float4 units, from Rg or Pbuf), bit 28: q|k|v item (else P), bit 29: in range
    unsigned code[U];
};
// The item -> address maps depend on the thread and the workgroup's row count only: planned once per hg loop
// (the divisions are the expensive part), then every issue / commit is an unpack and an add.
template <int MT, int HGS, int PL_ = 16 * MT + 4>
DEVI void co_reload_plan(CoReload<MT, HGS>& rl, const CoGeo& g, bool need_p, int tid) {
    using RL = CoReload<MT, HGS>;
    constexpr int PS = 16 * MT, LQ = 80 * HGS + 4, PL = PL_, PT = 16 * MT * PL;
    const int nq = g.rows * RL::NQ, total = nq + (need_p ? g.rows * RL::NP : 0);
#pragma unroll
    for (int u = 0; u < RL::U; ++u) {
        const int it0 = u * DFF_NTHREADS + tid;
        const int it = min(it0, total - 1);       // out-of-range items re-read the last one and store nothing
        unsigned code;
        if (it < nq) {
            const int row = it / RL::NQ, r2 = it - row * RL::NQ;
            const int hh = r2 / (DFF_QKVW / 4), c4 = r2 - hh * (DFF_QKVW / 4);
            const int colq = 4 * c4;
            const int reg = (colq >= 80) + (colq >= 144);
            const int cl = colq - 80 * reg + 16 * (reg >> 1);
            const unsigned src = (unsigned)((hh * g.RN + row) * (DFF_QKVW / 4) + c4);
            const unsigned dst = (unsigned)(reg * g.RN * LQ + row * LQ + hh * 80 + cl) >> 2;
            code = src | dst << 14 | 1u << 28;
        } else {
            const int ip = it - nq;
            const int row = ip / RL::NP, r2 = ip - row * RL::NP;
            const int hh = r2 / (4 * MT), c4 = r2 - hh * (4 * MT);
            const unsigned src = (unsigned)((hh * g.RN + row) * (PS / 4) + c4);
            const unsigned dst = (unsigned)(hh * PT + row * PL + 4 * c4) >> 2;
            code = src | dst << 14;
            if (PL_ < 16 * MT && 4 * c4 >= PL) { rl.code[u] = code; continue; }   // tight tile array: these columns do not exist (pad beads)
        }
        rl.code[u] = code | (it0 < total ? 1u << 29 : 0u);
    }
}
// every lane issues all U loads: a fixed number of loads in flight lets the compiler count vmcnt exactly
template <int MT, int HGS>
DEVI void co_reload_issue(CoReload<MT, HGS>& rl, const gfloat* sqkv /* head hg*HGS */, const gfloat* sP) {
    using RL = CoReload<MT, HGS>;
#pragma unroll
    for (int u = 0; u < RL::U; ++u) {
        const unsigned code = rl.code[u];
        const gfloat* const base = (code >> 28 & 1u) ? sqkv : sP;
        rl.t[u] = ld_ntg4<DFF_SITE_LD(MT, 1)>(base + 4 * (code & 0x3fffu));
    }
    // without this the compiler sinks the loads down to their use in co_reload_commit (no global store in between)
    asm volatile("" ::: "memory");
}
template <int MT, int HGS>
DEVI void co_reload_commit(const CoReload<MT, HGS>& rl, const CoGeo& g) {
    using RL = CoReload<MT, HGS>;
#pragma unroll
    for (int u = 0; u < RL::U; ++u) {
        const unsigned code = rl.code[u];
        lfloat* const base = (code >> 28 & 1u) ? g.Rg : g.Pbuf;
        if (code >> 29 & 1u) *(lf32x4*)(base + 4 * (code >> 14 & 0x3fffu)) = rl.t[u];
    }
}


// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// Workgroup barrier.  With SPILL the residual stream lives in global memory (the stash) and is
// handed between different threads of the workgroup, so the barrier must also wait for this
// wave's outstanding global stores (hipcc's plain __syncthreads() does not emit that wait).
template <bool SPILL>
DEVI void wg_sync() {
    if (SPILL) __threadfence_block();
    __syncthreads();
}

// PAIR: TWO workgroups per protein, for batches that leave half the CUs idle (protein G at 128 per GPU).  Block b and
// block b + 8 (same XCD under the observed b % 8 placement -- used for speed only) own heads 0..3 / 4..7 and one of the two
// FFN chunks each; everything row-wise (LayerNorm, gates, the integrator) runs redundantly in both.  The four H-wide GEMM
// outputs per layer (attention out, FFN out and their backward counterparts) and the final dE/dx are partial sums: each
// workgroup publishes its partial tile (plain stores -> barrier -> agent-scope release fence -> flag, MI355X_MICROARCH.md
// "inter-workgroup visibility"), waits for its partner's flag (one lane polls, agent-scope acquire, barrier) and adds the
// partner's tile -- a + b == b + a, so both workgroups continue with bit-identical values and stay in lockstep without
// any further communication: 13 exchanges of <= 29 KB per step.  Both blocks of a pair must be resident at once: the host
// only launches this variant with at most one block per CU.  A bounded spin (xflag error word) replaces a hang.
template <int H, int MT, int HGS, bool SPILL, bool GEN, bool SPW, bool PAIR = false>
__global__ __launch_bounds__(DFF_NTHREADS) void dff_fused_kernel(const DffModelDev m, const DffRunArgs a) {
    using LL = LdsLayout<H, MT, HGS, SPILL>;
    static_assert(!PAIR || (!GEN && DFF_HEADS / HGS % 2 == 0 && 4 * H / LL::FC >= 2), "PAIR splits head groups and FFN chunks in two");
    constexpr int LH = LL::LH, LQ = LL::LQ, F = LL::F, FC = LL::FC, LF = LL::LF;
    constexpr int NT_H = H / 16;                       // output tiles of an H-wide GEMM
    constexpr int NTW = (NT_H + DFF_NWAVES - 1) / DFF_NWAVES;
    constexpr int NHG = DFF_HEADS / HGS;
    constexpr int NCH = F / FC;
    // lanes per row of the row stages (RowMap).  16: up to 32 rows per pass; 8 lanes (64 rows per pass: villin's 35 and
    // protein G's 56 rows in one) was built and measured: 16 values per lane and array spill (412 B of scratch on villin's
    // variant), villin +5 %, protein G no better than the two 16-lane passes.
    constexpr int LPG = 16;
    constexpr int PLT = LL::template pl<SPW>();   // leading dimension of the P / dS tile arrays (TIGHT: 16 MT - 4, see LdsLayout)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    Ctx c;
    c.N = m.N; c.G = a.G; c.L = m.L;
    // PAIR: blocks b and b + 8 form pair (b & 7) + 8 (b >> 4); hf = which half of the heads / FFN chunks this block owns
    // (tests, a.xslow == 2: partners are blocks b and b + 1 instead -- under the round-robin placement they sit on DIFFERENT XCDs,
    // the XCC-ID handshake below finds that out by itself and the exchanges run the agent-scope protocol where it is needed)
    const bool adj = PAIR && a.xslow == 2;
    const int hf = PAIR ? (int)(adj ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1)) : 0;
    const int unit = PAIR ? (int)(adj ? (blockIdx.x >> 1) : ((blockIdx.x & 7) + 8 * (blockIdx.x >> 4))) : (int)blockIdx.x;
    c.b0 = a.b_base + unit * a.G;
    c.gcnt = min(a.G, a.B - c.b0);
    if (c.gcnt <= 0) return;   // (both blocks of a pair leave together)
    c.rows = c.gcnt * c.N;
    c.NP = c.N <= 8 ? 8 : c.N <= 16 ? 16 : c.N <= 32 ? 32 : 64;
    const LL ll(c.N, c.G, SPW);
    lu32* const asplit = (lu32*)(smem + ll.asplit);
    const unsigned junk_b = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lfloat*)((lfloat*)smem + ll.junk));
    // The weights of the NEXT GEMM phase are pulled into the XCD's L2 while the current phase runs (l2_touch).  One lambda per
    // weight image; the fp32 images are [tile][k-block of 16] x 1 KiB, the split ones [tile][k-block of 32] x 3 KiB.
    // The last wave asks.  When matters: vmcnt retires in order, so the wave's next wait for a load of its own also waits
    // for the (missing) warm-up lines -- in the pipelined attention phases it asks after its own work, in the slack the
    // row-tile waves leave it before the barrier.
    constexpr size_t UB = SPW ? 3072 : 1024;          // bytes of one (tile, k-block) unit
    constexpr bool FWD16 = SPW && (DFF_F16G & 1);     // forward weight GEMMs on the two-piece fp16 format
    constexpr size_t UBF = FWD16 ? 2048 : UB;         // ... whose images have 2 KB units
    constexpr bool FFB16 = SPW && (DFF_F16G & 2);     // FFN backward (W2T, W1T) likewise, on row-scaled operands
    constexpr size_t UBB2 = FFB16 ? 2048 : UB;
    constexpr bool GX16 = SPW && (DFF_F16G & 4);      // G_ext (WoxT) likewise: dattn row-scaled, unscaled by the GEMM's epilogues
    constexpr size_t UBG = GX16 ? 2048 : UB;
    lfloat* const rscl = (lfloat*)smem + ll.rsc;
    constexpr bool QT16 = SPW && DFF_QT16 && GX16;    // QKV_ext^T likewise (needs the row scales of G_ext's input to derive its block scale)
    constexpr size_t UBQ = QT16 ? 2048 : UB;
    constexpr int KQ = SPW ? 32 : 16;                 // rows of a k-block
    const int wave_l2 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    auto l2w = [&](const void* base, size_t off, int ntiles, size_t stride, size_t bytes) {
        if constexpr (SPW && DFF_L2W)   // (the fp32 variants are MFMA-bound: protein G measured 578.8 with, 578.6 us without)
            if (wave_l2 == DFF_NWAVES - 1) l2_touch(junk_b, (const char*)base + off, ntiles, stride, (int)(bytes >> 7));
    };
    auto l2w_flat = [&](const void* base, size_t off, size_t bytes) { l2w(base, off, 8, bytes / 8, bytes / 8); };
    constexpr size_t QKV_HG = (size_t)HGS * 13 * (H / KQ) * UBF;    // Wqkvx: a head group's 13 HGS tiles, contiguous
    constexpr size_t GX_HG = (size_t)HGS * 5 * (H / KQ) * UBG;      // WoxT: its 5 HGS tiles
    constexpr size_t FFW_CH = (size_t)(LL::FC / 16) * (H / KQ) * UBB2;  // W2T: a chunk's tiles
    constexpr size_t FFW1_CH = (size_t)(LL::FC / 16) * (H / KQ) * UBF;  // W1: likewise
    auto l2_wqkv = [&](const DffLayerDev& w, int hg) { l2w_flat(SPW ? (const void*)w.Wqkvx_s : (const void*)w.Wqkvx_p, hg * QKV_HG, QKV_HG); };
    auto l2_wgx = [&](const DffLayerDev& w, int hg) { l2w_flat(SPW ? (const void*)w.WoxT_s : (const void*)w.WoxT_p, hg * GX_HG, GX_HG); };
    auto l2_w1 = [&](const DffLayerDev& w, int ch) { l2w_flat(SPW ? (const void*)w.W1_s : (const void*)w.W1_p, ch * FFW1_CH, FFW1_CH); };
    auto l2_w2t = [&](const DffLayerDev& w, int ch) { l2w_flat(SPW ? (const void*)w.W2T_s : (const void*)w.W2T_p, ch * FFW_CH, FFW_CH); };
    // "tall" images (Nout = H): a few k-blocks of every one of the H / 16 tiles
    auto l2_tall = [&](const void* base, int kb0, int nkb, int kbtot, size_t ub = (SPW ? 3072 : 1024)) { l2w(base, (size_t)kb0 * ub, H / 16, (size_t)kbtot * ub, (size_t)nkb * ub); };
    auto l2_w2 = [&](const DffLayerDev& w, int ch) { l2_tall(SPW ? (const void*)w.W2_s : (const void*)w.W2_p, ch * (LL::FC / KQ), LL::FC / KQ, LL::F / KQ, UBF); };
    auto l2_w1t = [&](const DffLayerDev& w, int ch) { l2_tall(SPW ? (const void*)w.W1T_s : (const void*)w.W1T_p, ch * (LL::FC / KQ), LL::FC / KQ, LL::F / KQ, UBB2); };
    constexpr int WO_KB = SPW ? 2 : 5, WQT_KB = SPW ? 6 : 13;       // k-blocks per head (split: the 64 regular rows only)
    auto l2_wo = [&](const DffLayerDev& w, int hg) { l2_tall(SPW ? (const void*)w.Wox_s : (const void*)w.Wox_p, hg * HGS * WO_KB, HGS * WO_KB, DFF_HEADS * WO_KB, UBF); };
    auto l2_wqkvT = [&](const DffLayerDev& w, int hg) { l2_tall(SPW ? (const void*)w.WqkvxT_s : (const void*)w.WqkvxT_p, hg * HGS * WQT_KB, HGS * WQT_KB, DFF_HEADS * WQT_KB, UBQ); };
    c.xst = smem + ll.xst; c.xs = smem + ll.xs; c.dxs = smem + ll.dxs; c.vst = smem + ll.vst;
    c.cm = smem + ll.cm; c.tn = smem + ll.tn;
    c.abuf = smem + ll.abuf;
    c.rsc = smem + ll.rsc;
    c.asp = asplit; c.RNa = c.G * c.N;
    c.Pbuf = smem + ll.Pbuf; c.dSbuf = smem + ll.dSbuf; c.Rg = smem + ll.Rg;
    c.sl = dff_stash_layout(c.N, c.G, H, m.L, MT);
    c.stash = a.stash + (size_t)blockIdx.x * a.stash_stride;
    c.resbuf = SPILL ? (c.stash + c.sl.dn_spill) : (smem + ll.resbuf);
    float* tbuf = c.Rg;  // GEMM outputs of width H alias the start of the head-group region
    const int tid = threadIdx.x;
    constexpr int HG0 = 0;
    const int hg_lo = PAIR ? hf * (NHG / 2) : 0, hg_hi = PAIR ? (hf + 1) * (NHG / 2) : NHG;   // this block's head groups
    const int ch_lo = PAIR ? hf * ((NCH + 1) / 2) : 0, ch_hi = PAIR && hf == 0 ? (NCH + 1) / 2 : NCH;   // ... and FFN chunks (H = 96: 2 + 1)
    (void)HG0;
    // ---- PAIR: partial-tile exchange (see the comment above the kernel) ----
    unsigned xseq = 0;   // exchanges done so far in this launch (both blocks of a pair count alike)
    bool xfast = false;   // both blocks of the pair run on one XCD (decided once per launch, below): the exchange stays in its L2
    auto pair_exchange = [&](float* tile /* LDS, rows x ncols floats, leading dimension ld */, int ncols, int ld) {
        if constexpr (PAIR) {
            const int tid_ = tid_now();
            const size_t slot_floats = (size_t)(c.G * c.N) * (H + 4);
            float* const mine = a.xchg + ((size_t)(2 * unit + hf) * 2 + (xseq & 1)) * slot_floats;
            const float* const theirs = a.xchg + ((size_t)(2 * unit + (1 - hf)) * 2 + (xseq & 1)) * slot_floats;
            unsigned* const flags = a.xflag + 1 + 2 * unit;   // (word 0 is the sticky error word)
            const int n4 = ncols / 4;   // ncols % 4 == 0, ld % 4 == 0
            wg_sync<SPILL>();           // the tile is complete
            // The tile travels as agent-scope (sc1) 16-byte stores and loads: such a store is written through to the
            // device's coherence point and such a load bypasses the non-coherent caches, so publishing needs no
            // release / acquire FENCE -- at agent scope those write back / invalidate the XCD's whole L2, which is full of
            // dirty stash lines (measured: ~10 us per exchange, 17 % of the protein G step).  Order is kept at the ISA
            // level: every thread waits for its own stores (s_waitcnt vmcnt(0)) before the barrier that precedes the flag
            // store, and the tile is read only after the barrier that follows the flag poll.
            for (int it = tid_; it < c.rows * n4; it += DFF_NTHREADS) {
                const int row = it / n4, c4 = it - row * n4;
                const f32x4 v = *(const f32x4*)(tile + row * ld + 4 * c4);
                if (xfast) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine + (size_t)row * ncols + 4 * c4), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + (size_t)row * ncols + 4 * c4), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid_ == 0) {
                if (xfast) asm volatile("global_store_dword %0, %1, off" ::"v"(flags + hf), "v"(xseq + 1) : "memory");   // (stays in the shared L2)
                else __hip_atomic_store(flags + hf, xseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // never hang the GPU: the spin is bounded (~1 s), and once any pair has given up (error word set) nobody spins
                unsigned* const err = a.xflag;
                unsigned spins = 0;
                while (__hip_atomic_load(flags + (1 - hf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < xseq + 1) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023u) == 0 &&
                        (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        atomicOr(err, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            {   // all of this thread's loads in flight at once (at most 4: rows <= 64, ncols <= 128), one wait, then the adds
                f32x4 pv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int it = tid_ + k * DFF_NTHREADS;
                    const int row = it / n4, c4 = it - row * n4;
                    pv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (it < c.rows * n4)
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[k]) : "v"(theirs + (size_t)row * ncols + 4 * c4) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3])::"memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int it = tid_ + k * DFF_NTHREADS;
                    const int row = it / n4, c4 = it - row * n4;
                    if (it < c.rows * n4) {
                        float* const t = tile + row * ld + 4 * c4;
                        *(f32x4*)t = *(const f32x4*)t + pv[k];
                    }
                }
            }
            wg_sync<SPILL>();
            ++xseq;
        }
    };
    const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = c.N, RN = c.G * N, rows = c.rows;
    const lfloat* const abufL = (const lfloat*)smem + ll.abuf;
    CoGeo geo;
    geo.qs = 1.0f; geo.qsi = 1.0f;
    {
        lfloat* const sm = (lfloat*)smem;
        geo.Rg = sm + ll.Rg; geo.Pbuf = sm + ll.Pbuf; geo.dSbuf = sm + ll.dSbuf; geo.xs = sm + ll.xs;
        geo.dxw = sm + ll.dxw + wave_ * RN * 4;
        geo.m12 = sm + ll.m12;
        geo.lsp = (lu32*)(sm + ll.lsplit);
        geo.lsq = geo.lsp + RN * LL::LSV;
        if constexpr (LL::TIGHT_OK && SPW) geo.lsq = (lu32*)(sm + ll.Pbuf);   // (tight layout: the P tile array, dead after the dV phase)
        geo.prow = (const int __attribute__((address_space(3)))*)(sm + ll.prow);
        geo.N = N; geo.RN = RN; geo.rows = rows;
    }

    // zero LDS once (pad columns of the 32-wide buffers must be 0; pad rows must be finite)
    for (int i = tid; i < (int)ll.total; i += DFF_NTHREADS) smem[i] = 0.f;
    wg_sync<SPILL>();
    if constexpr (PAIR) {
        // A launch on top of a failed one (sticky error word set: an earlier PAIR launch lost a partner) leaves at once
        // -- the host does not look at the word on the launch path (that would synchronise the stream), it reads it at its
        // own synchronisation points (dff_model_status).  One thread reads, everybody agrees through LDS.
        if (tid == 0) ((unsigned*)smem)[ll.junk] = __hip_atomic_load(a.xflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned failed = ((const unsigned*)smem)[ll.junk];
        __syncthreads();
        if (failed) {
            // ... but not silently: a caller that never reaches a status check (Model.score() hands out a
            // torch.empty buffer) must not read uninitialised memory as forces -- this launch's OUTPUTS become NaN: forces
            // (+ energies), or the frames and kinetic energies of a Langevin launch, or the samples of a reverse chain.
            // The Langevin state (x, v) is left as it was: after the host's status check has reported and cleared the
            // word the caller can run the same steps again.
            const float qnan = __builtin_nanf("");
            const int nr3 = c.rows * 3;
            const size_t o3 = (size_t)c.b0 * c.N * 3;
            if (a.mode == DFF_MODE_SCORE) {
                for (int i = tid; i < nr3; i += DFF_NTHREADS) a.force_out[o3 + i] = qnan;
                if (a.energy_out) for (int i = tid; i < c.rows; i += DFF_NTHREADS) a.energy_out[(size_t)c.b0 * c.N + i] = qnan;
            } else if (a.mode == DFF_MODE_LANGEVIN) {
                const int nf = a.n_steps / (a.save_interval > 0 ? a.save_interval : 1);
                for (int f = 0; f < nf; ++f) {
                    if (a.frames) for (int i = tid; i < nr3; i += DFF_NTHREADS) a.frames[(size_t)f * a.B * c.N * 3 + o3 + i] = qnan;
                    if (a.ke && tid < c.gcnt) a.ke[(size_t)f * a.B + c.b0 + tid] = qnan;
                }
            } else {
                for (int i = tid; i < nr3; i += DFF_NTHREADS) a.x_io[o3 + i] = qnan;
            }
            return;
        }
    }
    // PAIR, same-XCD fast path (round 4).  The two blocks of a pair exchange 13 partial tiles per step through agent-scope
    // (sc1) stores and loads, which are written through to / served from memory because the XCDs' L2s are not coherent with
    // each other.  Blocks b and b + 8 are observed to share an XCD -- NOT a guarantee -- so each block publishes the XCD it
    // actually runs on (s_getreg_b32 HW_REG_XCC_ID) and reads its partner's, once per launch: when they agree, the one L2
    // they share IS their coherence point, and the tiles and flags travel as plain stores (acknowledged by that L2:
    // s_waitcnt vmcnt(0)) and L1-bypassing loads that hit it; otherwise the sc1 protocol runs as before.
    if constexpr (PAIR && DFF_XFAST) {
        unsigned* const xw = a.xflag + 1 + 2 * a.xpairs + 2 * unit;
        if (tid == 0) {
            const unsigned my_xcc = (__builtin_amdgcn_s_getreg(20 | ((4 - 1) << 11)) & 15u) + 1u;
            __hip_atomic_store(xw + hf, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* const err = a.xflag;
            unsigned theirs = 0, spins = 0;
            while ((theirs = __hip_atomic_load(xw + (1 - hf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0 &&
                    (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    atomicOr(err, 1u);
                    break;
                }
            }
            ((unsigned*)smem)[ll.junk] = (theirs == my_xcc && a.xslow != 1) ? 1u : 0u;
        }
        __syncthreads();
        xfast = ((const unsigned*)smem)[ll.junk] != 0u;
        __syncthreads();
    }
    if (tid < 64) ((int*)smem)[ll.prow + tid] = tid < rows ? tid / N : -1;

    Prof pf;
    pf.on = (a.prof != nullptr) && blockIdx.x == 0 && tid == 0;
    pf.acc = (unsigned long long*)(smem + ll.prof);
    pf.last = __builtin_readcyclecounter();

    // ---- load state ----
    {
        const float* xin = (a.mode == DFF_MODE_SCORE) ? a.x_in : a.x_io;
        if (tid < rows * 4) {
            const int row = tid >> 2, cc = tid & 3;
            float xv = 0.f, vv = 0.f;
            if (cc < 3) {
                const size_t gi = ((size_t)c.b0 * N + row) * 3 + cc;
                if (a.mode == DFF_MODE_DDPM && a.init_prior) {
                    xv = philox_normal(a.seed, a.item_offset + c.b0 + row / N, 0xFFFFFFFFull, row % N, cc);
                } else {
                    xv = xin[gi];
                }
                if (a.mode == DFF_MODE_LANGEVIN && !a.overdamped) vv = a.v_io[gi];
            }
            c.xst[tid] = xv;
            c.vst[tid] = vv;
        }
        if (tid < c.gcnt) c.tn[tid] = (a.mode == DFF_MODE_SCORE) ? a.tnorm[c.b0 + tid] : a.t_norm;
        wg_sync<SPILL>();
        if (a.mode == DFF_MODE_DDPM && a.init_prior) {  // x_T = center_zero(randn)  ddpm.py:242
            bead_mean(c, c.xst, c.cm);
            wg_sync<SPILL>();
            if (tid < rows * 4) c.xst[tid] -= c.cm[(tid >> 2) / N * 4 + (tid & 3)];
            wg_sync<SPILL>();
        }
    }

    for (int step = 0; step < a.n_steps; ++step) {
        int t_int = 0;
        if (a.mode == DFF_MODE_DDPM) {
            t_int = a.t_start - step;
            if (tid < c.gcnt) c.tn[tid] = (1.0f * (float)t_int) / (float)m.T;  // ddpm.py:203
        }
        // ---- centring: Langevin re-centres the state itself (langevin_cgnet.py:739); the score
        // op always centres its own input (graph_transformer.py:87) ----
        bead_mean(c, c.xst, c.cm);
        wg_sync<SPILL>();
        if (tid < rows * 4) {
            const float xc = c.xst[tid] - c.cm[(tid >> 2) / N * 4 + (tid & 3)];
            if (a.mode == DFF_MODE_LANGEVIN) c.xst[tid] = xc;
            c.xs[tid] = xc;
            c.dxs[tid] = 0.f;
#pragma unroll
            for (int w = 0; w < DFF_NWAVES; ++w) (smem + ll.dxw)[w * RN * 4 + tid] = 0.f;
        }
        wg_sync<SPILL>();
        if (a.mode == DFF_MODE_LANGEVIN) {  // second (no-op-sized) centring inside the score op
            bead_mean(c, c.xs, c.cm);
            wg_sync<SPILL>();
            if (tid < rows * 4) c.xs[tid] -= c.cm[(tid >> 2) / N * 4 + (tid & 3)];
            wg_sync<SPILL>();
        }

        pf.tick(0);
        // =============================== forward ===============================
        // Langevin: t is fixed, so the x-independent layer-0 inputs (node features, LN1, u, q, k, v)
        // are the same every step: computed on step 0, re-read from the stash afterwards.
        // with a precomputed table entry for this step's noise level (ensure_l0_table) layer 0 never runs its
        // QKV GEMM; without one, Langevin (fixed t) re-reads what step 0 left in this workgroup's own stash
        const bool tab = a.l0_tab != nullptr;
        c.l0 = tab ? a.l0_tab + (size_t)(a.mode == DFF_MODE_DDPM ? t_int : 0) * c.sl.layer_stride : c.stash;
        // absolute coordinates make layer 0 x-dependent: no caching, and the VJP runs through layer 0 and the embedding
        const bool full0 = GEN && m.in_abs;
        const bool cached0 = !full0 && (tab || ((a.mode == DFF_MODE_LANGEVIN) && step > 0));
        if (!cached0) {
            node_embed<H, GEN>(c, m);
            wg_sync<SPILL>();
        }
        for (int l = 0; l < m.L; ++l) {
            const DffLayerDev& lw = m.layer[l];
            float* sb = c.stash + (size_t)l * c.sl.layer_stride;
            const bool cached = cached0 && l == 0;
            gfloat* const sqkv = (gfloat*)sb + c.sl.qkvx;
            const gfloat* const sqkv_r = (const gfloat*)(l == 0 ? c.l0 : sb) + c.sl.qkvx;   // cached reads
            gfloat* const sPl = (gfloat*)sb + c.sl.P;
            if (cached) {
                for (int it = tid; it < rows * H; it += DFF_NTHREADS) {
                    const int row = it / H, col = it - row * H;
                    c.resbuf[row * LH + col] = ld_nt(c.l0 + c.sl.nodes_in + it);
                }
            } else {
                l2_wqkv(lw, hg_lo);
                row_ln1<H, LPG, SPW, FWD16>(c, lw, l);
                wg_sync<SPILL>();
            }
            pf.tick(1);
            f32x4 acc_o[NTW][MT];
            acc_zero<MT, NTW>(acc_o);
            // The split variants run the head loop as a two-stage pipeline: the logits / softmax / PV of a head group keep
            // only HGS * MT waves busy (a wave owns a row tile), so the other waves compute QKV_ext of the NEXT group
            // meanwhile and park its tiles in registers; they write them (LDS + stash) next to the W_o GEMM of the
            // current group, once the barrier has retired its q | k | v.  o_ext's extension columns live in dSbuf
            // (idle in the forward pass) so that R0 is free by then.
            constexpr bool PIPE = SPW && !GEN && HGS == 1 && MT < 4;   // HGS = 2: the parked tiles (7 x MT) do not fit the register file; MT = 4: 16 parked tiles neither
            lfloat* const oxt = PIPE ? geo.dSbuf : nullptr;
            constexpr int DWO = 2 * HGS < 4 ? 2 * HGS : 4;
            auto wo_gemm = [&](int hg, auto pre, u32x4 (&bw)[DWO][NTW][3]) {
                ExtW<NTW, HGS> ew;
                if constexpr (DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 5 + 4; }, lw.Wox_p, DFF_HEADS * 5, NT_H);
                gemm_tall_split_st_b<MT, NTW, 2 * HGS, decltype(pre)::value, FWD16>(acc_o, (64 * HGS + DFF_SPAD) / 2, (const lu32*)(geo.Rg + 3 * RN * LQ), RN, RN,
                                                     lw.Wox_s, 2 * DFF_HEADS, hg * HGS * 2, NT_H, bw);
                if constexpr (!DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 5 + 4; }, lw.Wox_p, DFF_HEADS * 5, NT_H);
                if (oxt) ext_apply<MT, NTW, HGS>(acc_o, ew, [=](int i) { return i * 16; }, oxt, 16 * HGS, RN, NT_H);
                else ext_apply<MT, NTW, HGS>(acc_o, ew, [=](int i) { return i * 80 + 64; }, geo.Rg, LQ, RN, NT_H);
            };
            if constexpr (PIPE) if (!cached) {
                constexpr int NI = HGS * MT, NWH = DFF_NWAVES - NI, NTQ = HGS * 13, CNTH = (NTQ + NWH - 1) / NWH;
                const int tid = tid_now();
                lfloat* const Rl = geo.Rg;
                gfloat* const junk = (gfloat*)c.stash + c.sl.junk + 4 * (tid & 63);
                auto mk_pre = [=](int hg) {
                    const gfloat* bq = (const gfloat*)lw.bqkvx + hg * HGS * DFF_QKVW;
                    return [=](int nt, float (&aux)[4]) { ld4_aux(aux, bq + nt * 16 + 4 * ((tid & 63) >> 4)); };
                };
                auto mk_epi = [=](int hg) {
                    gfloat* const sq = sqkv + (size_t)hg * HGS * RN * DFF_QKVW;
                    return [=](int nt, int mt, const f32x4& acc, const float (&aux)[4], bool valid, int) {
                        const int lane = tid & 63, c4 = 4 * (lane >> 4), row = mt * 16 + (lane & 15);
                        const int hh = nt / 13, tt = nt - 13 * hh;
                        const int reg = (tt >= 5) + (tt >= 9);
                        const bool ok = valid && row < rows;
                        const f32x4 v = acc + (f32x4){aux[0], aux[1], aux[2], aux[3]};
                        if (ok) *(lf32x4*)(Rl + reg * RN * LQ + row * LQ + hh * 80 + 16 * (tt - 5 * reg + (reg >> 1)) + c4) = v;
                        st_ntg4<DFF_SITE_ST(MT, 2)>(ok ? sq + (size_t)hh * RN * DFF_QKVW + (size_t)row * DFF_QKVW + 16 * tt + c4 : junk, v);
                    };
                };
                gemm_wide_split_st<MT, H / 32, NTQ, 4, 0, DFF_NWAVES, 3, FWD16>(asplit, RN, RN, lw.Wqkvx_s, hg_lo * NTQ, mk_pre(hg_lo), mk_epi(hg_lo));
                co_fill_x<HGS, GEN>(geo);   // once per layer: the forward pass never overwrites the K_ext / V_ext extension columns
                wg_sync<SPILL>();
                pf.tick(3);
                for (int hg = hg_lo; hg < hg_hi; ++hg) {
                    const bool more = hg + 1 < hg_hi;
                    f32x4 held[CNTH][MT];   // per iteration: nothing to keep alive on the softmax waves' path
                    if (wave_ < NI) {
                        co_softmax_pv_t<MT, HGS, SPW>(geo, sPl + (size_t)hg * HGS * RN * c.sl.PS, oxt, junk);
                    } else if (more) {
                        gemm_wide_split_st<MT, H / 32, NTQ, 1, NI, NWH, 2, FWD16>(asplit, RN, RN, lw.Wqkvx_s, (hg + 1) * NTQ,
                            [](int, float (&)[1]) {},
                            [&](int, int mt, const f32x4& acc, const float (&)[1], bool, int i) { held[i][mt] = acc; });
                    }
                    // (the GEMM waves are done before the softmax waves: the last one asks for the next phases' weights
                    // in the slack, and its next wait for a load of its own is a barrier away)
                    l2_wo(lw, hg);
                    if (hg + 2 < hg_hi) l2_wqkv(lw, hg + 2);
                    else if (!more) l2_w1(lw, ch_lo);
                    u32x4 bw[DWO][NTW][3];
                    constexpr bool WOPRE = DFF_WOPRE;   // W_o's operands cross the barrier in registers
                    if constexpr (WOPRE) tall_ring_fill<NTW, DWO, MT, FWD16>(bw, lw.Wox_s, 2 * DFF_HEADS, hg * HGS * 2, NT_H);
                    wg_sync<SPILL>();
                    pf.tick(4);
                    float bias4[CNTH][4];
                    if (more && wave_ >= NI) {   // the parked tiles' biases: requested before W_o's weights, used after its products
                        auto pre = mk_pre(hg + 1);
#pragma unroll
                        for (int i = 0; i < CNTH; ++i) pre(min(wave_ - NI + NWH * i, NTQ - 1), bias4[i]);
                    }
                    wo_gemm(hg, std::integral_constant<bool, WOPRE>(), bw);
                    if (more && wave_ >= NI) {
                        auto epi = mk_epi(hg + 1);
#pragma unroll
                        for (int i = 0; i < CNTH; ++i) {
                            const int wv = wave_ - NI + NWH * i;
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) epi(min(wv, NTQ - 1), mt, held[i][mt], bias4[i], wv < NTQ, i);
                        }
                    }
                    wg_sync<SPILL>();
                    pf.tick(6);
                }
            }
            for (int hg = hg_lo; hg < ((PIPE && !cached) ? hg_lo : hg_hi); ++hg) {
                // [q|u|k|v] of HGS heads -> R0,R1,R2 (+ stash)
                if (cached) {
                    const int tid = tid_now();
                    CoReload<MT, HGS> rl;
                    co_reload_plan<MT, HGS, PLT>(rl, geo, false, tid);
                    co_reload_issue<MT, HGS>(rl, sqkv_r + (size_t)hg * HGS * RN * DFF_QKVW, sPl);
                    co_reload_commit<MT, HGS>(rl, geo);
                }
                else {
                    const int tid = tid_now();
                    const gfloat* bq = (const gfloat*)lw.bqkvx + hg * HGS * DFF_QKVW;
                    gfloat* const sq = sqkv + (size_t)hg * HGS * RN * DFF_QKVW;
                    lfloat* const Rl = geo.Rg;
                    auto qkv_pre = [=](int nt, float (&aux)[4]) { ld4_aux(aux, bq + nt * 16 + 4 * ((tid & 63) >> 4)); };
                    gfloat* const junk = (gfloat*)c.stash + c.sl.junk + 4 * (tid & 63);
                    auto qkv_epi = [=](int nt, int mt, const f32x4& acc, const float (&aux)[4], bool valid, int) {
                            const int lane = tid & 63, c4 = 4 * (lane >> 4), row = mt * 16 + (lane & 15);
                            const int hh = nt / 13, tt = nt - 13 * hh;
                            const int reg = (tt >= 5) + (tt >= 9);
                            const bool ok = valid && row < rows;
                            const f32x4 v = acc + (f32x4){aux[0], aux[1], aux[2], aux[3]};
                            if (ok) *(lf32x4*)(Rl + reg * RN * LQ + row * LQ + hh * 80 + 16 * (tt - 5 * reg + (reg >> 1)) + c4) = v;
                            st_ntg4<DFF_SITE_ST(MT, 2)>(ok ? sq + (size_t)hh * RN * DFF_QKVW + (size_t)row * DFF_QKVW + 16 * tt + c4 : junk, v);
                        };
                    if constexpr (SPW)
                        if constexpr (MT >= DFF_K2_MT && DFF_K2) gemm_wide_split_k2<MT, H / 32, HGS * 13, 4, FWD16>(asplit, RN, RN, lw.Wqkvx_s, hg * HGS * 13, qkv_pre, qkv_epi);
                        else gemm_wide_split_st<MT, H / 32, HGS * 13, 4, 0, DFF_NWAVES, 3, FWD16>(asplit, RN, RN, lw.Wqkvx_s, hg * HGS * 13, qkv_pre, qkv_epi);
                    else
                        gemm_wide_st<MT, NT_H, HGS * 13, 4>(abufL, LH, RN, lw.Wqkvx_p, hg * HGS * 13, qkv_pre, qkv_epi);
                }
                co_fill_x<HGS, GEN>(geo);
                wg_sync<SPILL>();
                pf.tick(3);
                l2_wo(lw, hg);
                if constexpr (GEN)   // (MT = 4, protein G, ran the C-layout version until the transposed one's stash stores were vectorised: 377.6 -> 373.1 us at 128 per GPU, 648 -> 634 at 256)
                    co_softmax_pv<MT, HGS, GEN, SPW, PLT>(geo, sPl + (size_t)hg * HGS * RN * c.sl.PS,
                                                     (gfloat*)sb + c.sl.m12 + (size_t)hg * HGS * RN * 4, oxt);
                else
                    co_softmax_pv_t<MT, HGS, SPW>(geo, sPl + (size_t)hg * HGS * RN * c.sl.PS, oxt,
                                                  (gfloat*)c.stash + c.sl.junk + 4 * (tid_now() & 63));
                wg_sync<SPILL>();
                pf.tick(4);
                // attn_out += o_ext [W_o ; W_oc]   (K = 80 per head)
                if (hg + 1 < hg_hi) { if (!cached) l2_wqkv(lw, hg + 1); }
                else l2_w1(lw, ch_lo);
                if constexpr (SPW) {   // 64 regular rows per head on the split path, the extension block on the fp32 one
                    { u32x4 bw[DWO][NTW][3]; wo_gemm(hg, std::false_type(), bw); }
                } else
                gemm_tall_kb_st<MT, NTW, 5, 5 * HGS>(acc_o,
                    [=](int i, int& aoff, int& wkb) {
                        const int hh = i / 5, kb = i - 5 * hh;
                        aoff = hh * 80 + 16 * kb;
                        wkb = (hg * HGS + hh) * 5 + kb;
                    },
                    geo.Rg, LQ, RN, lw.Wox_p, DFF_HEADS * 5, NT_H);
                wg_sync<SPILL>();
                pf.tick(6);
            }
            store_tall<MT, NTW>(acc_o, tbuf, LH, rows, NT_H, hf == 0 ? lw.bo : nullptr);
            pair_exchange(tbuf, H, LH);
            wg_sync<SPILL>();
            row_gate1_ln2<H, LPG, SPW, FWD16>(c, lw, l, tbuf);
            wg_sync<SPILL>();
            pf.tick(7);
            // FFN: Linear(H,4H) -> GELU(erf) -> Linear(4H,H)   (graph_transformer.py:264-267)
            f32x4 acc_f[NTW][MT];
            acc_zero<MT, NTW>(acc_f);
            for (int ch = ch_lo; ch < ch_hi; ++ch) {
                {
                    const int tid = tid_now();
                    const gfloat* const b1g = (const gfloat*)lw.b1 + ch * FC;
                    gfloat* const shp = (gfloat*)sb + c.sl.h_pre + ch * FC;
                    lfloat* const hl = geo.Rg;
                    auto w1_pre = [=](int nt, float (&aux)[4]) { ld4_aux(aux, b1g + 16 * nt + 4 * ((tid & 63) >> 4)); };
                    gfloat* const junk = (gfloat*)c.stash + c.sl.junk + 4 * (tid & 63);
                    auto w1_epi = [=](int nt, int mt, const f32x4& acc, const float (&aux)[4], bool valid, int) {
                            const int lane = tid & 63, cl = 16 * nt + 4 * (lane >> 4), row = mt * 16 + (lane & 15);
                            const bool ok = valid && row < rows;
                            f32x4 gv, gp;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { float v_, p_; gelu_both(acc[r] + aux[r], v_, p_); gv[r] = v_; gp[r] = p_; }
                            st_ntg4<DFF_SITE_ST(MT, 8)>(ok ? shp + (size_t)row * F + cl : junk, gp);   // the slot "h_pre" holds gelu'(h_pre)
                            if (ok) {
                                if constexpr (SPW) store_split4<FWD16>((lu32*)hl, RN, (FC + DFF_SPAD) / 2, row, cl, gv);
                                else *(lf32x4*)(hl + row * LF + cl) = gv;
                            }
                        };
                    l2_w2(lw, ch);
                    if constexpr (SPW)
                        if constexpr (MT >= DFF_K2_MT && DFF_K2) gemm_wide_split_k2<MT, H / 32, FC / 16, 4, FWD16>(asplit, RN, RN, lw.W1_s, ch * (FC / 16), w1_pre, w1_epi);
                        else gemm_wide_split_st<MT, H / 32, FC / 16, 4, 0, DFF_NWAVES, 3, FWD16>(asplit, RN, RN, lw.W1_s, ch * (FC / 16), w1_pre, w1_epi);
                    else
                        gemm_wide_st<MT, NT_H, FC / 16, 4>(abufL, LH, RN, lw.W1_p, ch * (FC / 16), w1_pre, w1_epi);
                }
                wg_sync<SPILL>();
                pf.tick(8);
                if (ch + 1 < ch_hi) l2_w1(lw, ch + 1);
                else if (l == m.L - 1 && m.conservative) l2_w2t(lw, ch_lo);
                if constexpr (SPW)
                    gemm_tall_split_st<MT, NTW, FC / 32, FWD16>(acc_f, (FC + DFF_SPAD) / 2, (const lu32*)geo.Rg, RN, RN, lw.W2_s, F / 32, ch * (FC / 32), NT_H);
                else
                gemm_tall_kb_st<MT, NTW, 0, FC / 16>(acc_f,
                    [=](int i, int& aoff, int& wkb) { aoff = 16 * i; wkb = ch * (FC / 16) + i; },
                    geo.Rg, LF, RN, lw.W2_p, F / 16, NT_H);
                wg_sync<SPILL>();
                pf.tick(9);
            }
            store_tall<MT, NTW>(acc_f, tbuf, LH, rows, NT_H, hf == 0 ? lw.b2 : nullptr);
            pair_exchange(tbuf, H, LH);
            wg_sync<SPILL>();
            row_gate2<H, LPG>(c, m, lw, l, tbuf, l == m.L - 1, a.energy_out);
            wg_sync<SPILL>();
            pf.tick(10);
        }

        // =============================== backward ===============================
        for (int l = m.conservative ? m.L - 1 : -1; l >= 0; --l) {
            const DffLayerDev& lw = m.layer[l];
            const float* sb = c.stash + (size_t)l * c.sl.layer_stride;
            if (l < m.L - 1) l2_w2t(lw, ch_lo);
            rowb_gate2<H, LPG, SPW, FFB16>(c, lw, l);
            wg_sync<SPILL>();
            pf.tick(11);
            // dh = dff W2 ; dh_pre = dh * gelu'(h_pre) ; df = dh_pre W1
            f32x4 acc_f[NTW][MT];
            acc_zero<MT, NTW>(acc_f);
            for (int ch = ch_lo; ch < ch_hi; ++ch) {
                {
                    const int tid = tid_now();
                    const gfloat* const shp = (const gfloat*)sb + c.sl.h_pre + ch * FC;
                    lfloat* const hl = geo.Rg;
                    auto w2t_pre = [=](int nt, float (&aux)[4 * MT]) {
                            const int lane = tid & 63, cl = 16 * nt + 4 * (lane >> 4);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                ld4_aux<DFF_SITE_LD(MT, 16)>(aux + 4 * mt, shp + (size_t)min(mt * 16 + (lane & 15), rows - 1) * F + cl);
                        };
                    auto w2t_epi = [=](int nt, int mt, const f32x4& acc, const float (&aux)[4 * MT], bool valid, int) {
                            const int lane = tid & 63, cl = 16 * nt + 4 * (lane >> 4), row = mt * 16 + (lane & 15);
                            if (valid && row < rows) {
                                const f32x4 v = acc * (f32x4){aux[mt * 4], aux[mt * 4 + 1], aux[mt * 4 + 2], aux[mt * 4 + 3]};
                                if constexpr (SPW) store_split4<FFB16>((lu32*)hl, RN, (FC + DFF_SPAD) / 2, row, cl, v);
                                else *(lf32x4*)(hl + row * LF + cl) = v;
                            }
                        };
                    l2_w1t(lw, ch);
                    if constexpr (SPW)
                        if constexpr (MT >= DFF_K2_MT && DFF_K2) gemm_wide_split_k2<MT, H / 32, FC / 16, 4 * MT, FFB16>(asplit, RN, RN, lw.W2T_s, ch * (FC / 16), w2t_pre, w2t_epi);
                        else gemm_wide_split_st<MT, H / 32, FC / 16, 4 * MT, 0, DFF_NWAVES, 3, FFB16>(asplit, RN, RN, lw.W2T_s, ch * (FC / 16), w2t_pre, w2t_epi);
                    else
                        gemm_wide_st<MT, NT_H, FC / 16, 4 * MT>(abufL, LH, RN, lw.W2T_p, ch * (FC / 16), w2t_pre, w2t_epi);
                }
                wg_sync<SPILL>();
                pf.tick(12);
                if (ch + 1 < ch_hi) l2_w2t(lw, ch + 1);
                else l2_wgx(lw, hg_lo);
                if constexpr (SPW)
                    gemm_tall_split_st<MT, NTW, FC / 32, FFB16>(acc_f, (FC + DFF_SPAD) / 2, (const lu32*)geo.Rg, RN, RN, lw.W1T_s, F / 32, ch * (FC / 32), NT_H);
                else
                gemm_tall_kb_st<MT, NTW, 0, FC / 16>(acc_f,
                    [=](int i, int& aoff, int& wkb) { aoff = 16 * i; wkb = ch * (FC / 16) + i; },
                    geo.Rg, LF, RN, lw.W1T_p, F / 16, NT_H);
                wg_sync<SPILL>();
                pf.tick(13);
            }
            store_tall<MT, NTW>(acc_f, tbuf, LH, rows, NT_H, nullptr);
            pair_exchange(tbuf, H, LH);
            wg_sync<SPILL>();
            // The stashed q|k|v|P rows of head group hg + 1 are requested as soon as those of hg have been committed
            // to LDS, a whole group's worth of phases before they are needed (group 0: before the row stage below):
            // by the next weight load they have landed, so they delay nothing (vmcnt retires in order).
            const gfloat* const sqkv = (const gfloat*)(l == 0 ? c.l0 : sb) + c.sl.qkvx;
            const gfloat* const sPl = (const gfloat*)sb + c.sl.P;
            CoReload<MT, HGS> rl;
            co_reload_plan<MT, HGS, PLT>(rl, geo, true, tid_now());
            co_reload_issue<MT, HGS>(rl, sqkv + (size_t)hg_lo * HGS * RN * DFF_QKVW, sPl + (size_t)hg_lo * HGS * RN * c.sl.PS);
            rowb_ln2_gate1<H, LPG, SPW, FFB16, GX16>(c, lw, l, tbuf);
            wg_sync<SPILL>();
            if constexpr (QT16) block_pow2_scale(rscl, geo.qs, geo.qsi);   // (this layer's dQ / dK / dV scale, from dattn's row scales)
            pf.tick(14);
            f32x4 acc_a[NTW][MT];
            acc_zero<MT, NTW>(acc_a);
            pf.tick(15);
            // Backward head pipeline of the variants that have one in the forward pass (SPW, !GEN, HGS = 1): dS / dQ keep only
            // MT waves busy, so the others compute G_ext of the NEXT head meanwhile (5 MT units, parked in registers) and
            // write it once dV / dK and the back-projection of the current head are done with R3.
            // Measured per shape: (128,3,1) villin 631 -> 586 us, (96,2,2) BBA 338 -> 329; (128,2,2) trp-cage 363 -> 391 (the 20 parked
            // units + two K = 128 weight tiles in flight spill: 248 B of scratch), so that shape keeps the serial loop.
            constexpr bool PIPEB = SPW && !GEN && MT < 4 && HGS * MT < DFF_NWAVES && (5 * HGS * MT) % (DFF_NWAVES - HGS * MT) == 0 &&
                                   (DFF_PIPEB_128_2 || !(H == 128 && HGS == 2));
            if constexpr (PIPEB) {
                constexpr int NI = HGS * MT, NWH = DFF_NWAVES - NI, NTG = 5 * HGS, DU = NTG * MT / NWH;
                const int tid = tid_now();
                lfloat* const Gl = geo.Rg + 3 * RN * LQ;
                lfloat* const dxw = geo.dxw;
                auto gx_epi = [=](int nt, int mt, const f32x4& acc0) {
                        const int lane = tid & 63, quad = lane >> 4, row = mt * 16 + (lane & 15);
                        const int hh = nt / 5, tt = nt - 5 * hh;
                        const f32x4 acc = GX16 ? acc0 * rscl[row] : acc0;   // (fp16 engine: back to true units)
                        if (row < rows) {
                            *(lf32x4*)(Gl + row * LQ + hh * 80 + 16 * tt + 4 * quad) = acc;
                            if (tt == 4 && quad == 0) {   // r = dE/dxrel: columns 64..66 of the head
                                dxw[row * 4 + 0] -= acc[0];
                                dxw[row * 4 + 1] -= acc[1];
                                dxw[row * 4 + 2] -= acc[2];
                            }
                        }
                    };
                auto commit_issue = [&](int hg) {   // rows of head hg -> LDS, request those of hg + 1
                    co_reload_commit<MT, HGS>(rl, geo);
                    if (hg + 1 < hg_hi)
                        co_reload_issue<MT, HGS>(rl, sqkv + (size_t)(hg + 1) * HGS * RN * DFF_QKVW,
                                                 sPl + (size_t)(hg + 1) * HGS * RN * c.sl.PS);
                    co_fill_x<HGS, GEN>(geo);
                };
                if constexpr (NTG == NWH && DFF_GXTILE0)   // (the first head's G_ext, all waves: whole tiles as well)
                    gemm_wide_split_st<MT, H / 32, NTG, 1, 0, DFF_NWAVES, 3, GX16>(asplit, RN, RN, lw.WoxT_s, hg_lo * NTG, [](int, float (&)[1]) {},
                        [&](int nt, int mt, const f32x4& acc, const float (&)[1], bool valid, int) { if (valid) gx_epi(nt, mt, acc); });
                else
                gemm_wide_units_split<MT, H / 32, NTG, GX16>(asplit, RN, RN, lw.WoxT_s, hg_lo * NTG, gx_epi);
                commit_issue(hg_lo);
                wg_sync<SPILL>();
                pf.tick(16);
                const bool deep = l > 0 || full0;
                for (int hg = hg_lo; hg < hg_hi; ++hg) {
                    const bool more = hg + 1 < hg_hi;
                    // (GXTILE: as many spare waves as G_ext has tiles -- villin's shape -- : a wave takes a whole TILE, all row tiles,
                    // so its 12 KB of weights cross the CU once; as (tile, row-tile) units three waves stream each tile)
                    constexpr bool GXTILE = (NTG == NWH || DFF_GXTILE > 1) && DFF_GXTILE;
                    constexpr int CNTG = (NTG + NWH - 1) / NWH;   // tiles per spare wave (GXTILE)
                    f32x4 gheld[GXTILE ? CNTG * MT : DU];
                    if (wave_ < NI) {
                        if (deep) co_ds<MT, HGS, true, GEN, PLT, LL::KVS && DFF_QSP>(geo);
                        else co_ds<MT, HGS, false, GEN>(geo);
                    } else if (more) {
                        if constexpr (GXTILE)
                            gemm_wide_split_st<MT, H / 32, NTG, 1, NI, NWH, 2, GX16>(asplit, RN, RN, lw.WoxT_s, (hg + 1) * NTG, [](int, float (&)[1]) {},
                                [&](int, int mt, const f32x4& acc, const float (&)[1], bool, int i) { gheld[i * MT + mt] = acc; });
                        else
                        gx_units_hold<MT, H / 32, NTG, NI, NWH, GX16>(asplit, RN, RN, lw.WoxT_s, (hg + 1) * NTG, gheld);
                    }
                    if (deep) l2_wqkvT(lw, hg);
                    if (hg + 2 < hg_hi) l2_wgx(lw, hg + 2);
                    else if (!more && l > 0) l2_w2t(m.layer[l - 1], ch_lo);
                    wg_sync<SPILL>();
                    pf.tick(17);
                    if (deep) {
                        u32x4 bq[4][NTW][3];
                        co_dv_dk<MT, HGS, false, GEN, LL::KVS, LL::VSP>(geo);
                        wg_sync<SPILL>();
                        pf.tick(18);
                        ExtW<NTW, HGS> ew;
                        if constexpr (DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 13 + 4; }, lw.WqkvxT_p, DFF_HEADS * 13, NT_H);
                        gemm_tall_qkvT_split<MT, NTW, HGS, LL::KVS, LL::VSP, 0, LL::KVS && DFF_QSP, QT16>(acc_a, geo.Rg, 4, RN, lw.WqkvxT_s, hg * HGS, NT_H, geo.lsp, bq, geo.lsq, geo.qs);
                        if constexpr (!DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 13 + 4; }, lw.WqkvxT_p, DFF_HEADS * 13, NT_H);
                        ext_apply<MT, NTW, HGS>(acc_a, ew, [=](int i) { return 4 * RN * LQ + i * 80 + 64; }, geo.Rg, LQ, RN, NT_H, QT16 ? geo.qs : 1.0f);
                    } else {
                        co_dv_dk<MT, HGS, true, GEN>(geo);
                    }
                    wg_sync<SPILL>();
                    pf.tick(19);
                    if (more) {
                        if (wave_ >= NI) {
#pragma unroll
                            for (int d = 0; d < (GXTILE ? CNTG * MT : DU); ++d) {
                                if constexpr (GXTILE) {
                                    const int nt = wave_ - NI + NWH * (d / MT);
                                    if (nt < NTG) gx_epi(nt, d % MT, gheld[d]);
                                } else {
                                    const int u = wave_ - NI + NWH * d;
                                    gx_epi(u / MT, u - (u / MT) * MT, gheld[d]);
                                }
                            }
                        }
                        commit_issue(hg + 1);
                        wg_sync<SPILL>();
                        pf.tick(16);
                    }
                }
            }
            for (int hg = hg_lo; hg < (PIPEB ? hg_lo : hg_hi); ++hg) {
                // G_ext = dattn [W_o ; W_oc]^T (dE/do | r = dE/dxrel) for the heads of this group -> R3 ;
                // dE/dx_i -= r_i
                {
                    const int tid = tid_now();
                    lfloat* const Gl = geo.Rg + 3 * RN * LQ;
                    lfloat* const dxw = geo.dxw;
                    auto gx_epi = [=](int nt, int mt, const f32x4& acc0) {
                            const int lane = tid & 63, quad = lane >> 4, row = mt * 16 + (lane & 15);
                            const int hh = nt / 5, tt = nt - 5 * hh;
                            const f32x4 acc = GX16 ? acc0 * rscl[row] : acc0;   // (fp16 engine: back to true units)
                            if (row < rows) {
                                *(lf32x4*)(Gl + row * LQ + hh * 80 + 16 * tt + 4 * quad) = acc;
                                if (tt == 4 && quad == 0) {   // r = dE/dxrel: columns 64..66 of the head
                                    dxw[row * 4 + 0] -= acc[0];
                                    dxw[row * 4 + 1] -= acc[1];
                                    dxw[row * 4 + 2] -= acc[2];
                                }
                            }
                        };
                    if (l > 0 || full0) l2_wqkvT(lw, hg);
                    if constexpr (SPW && MT == 4 && DFF_GXT)
                        // four row tiles: a wave takes a whole TILE (all row tiles), so its 12 KB of weights come through the CU
                        // once; as (tile, row-tile) units four waves each stream the same tile
                        gemm_wide_split_st<MT, H / 32, HGS * 5, 1, 0, DFF_NWAVES, 3, GX16>(asplit, RN, RN, lw.WoxT_s, hg * HGS * 5, [](int, float (&)[1]) {},
                            [&](int nt, int mt, const f32x4& acc, const float (&)[1], bool valid, int) { if (valid) gx_epi(nt, mt, acc); });
                    else if constexpr (SPW)
                        gemm_wide_units_split<MT, H / 32, HGS * 5, GX16>(asplit, RN, RN, lw.WoxT_s, hg * HGS * 5, gx_epi);
                    else
                        gemm_wide_units<MT, NT_H, HGS * 5>(abufL, LH, RN, lw.WoxT_p, NT_H, 0, hg * HGS * 5, gx_epi, NoHook());
                    co_reload_commit<MT, HGS>(rl, geo);
                    if (GEN) {   // [m1 | m2] rows of this head group (contiguous in the stash and in LDS); before the
                                 // next group's rows are requested: a load issued after them would wait for them
                        const gfloat* const sM = (const gfloat*)sb + c.sl.m12 + (size_t)hg * HGS * RN * 4;
                        for (int i2 = tid; i2 < HGS * RN * 4; i2 += DFF_NTHREADS) geo.m12[i2] = ld_ntg<DFF_SITE_LD(MT, 1)>(sM + i2);
                    }
                    if (hg + 1 < hg_hi)
                        co_reload_issue<MT, HGS>(rl, sqkv + (size_t)(hg + 1) * HGS * RN * DFF_QKVW,
                                                 sPl + (size_t)(hg + 1) * HGS * RN * c.sl.PS);
                }
                co_fill_x<HGS, GEN>(geo);
                wg_sync<SPILL>();
                pf.tick(16);
                constexpr bool FIVE = LL::NREG == 5;
                if (l > 0 || full0) {
                    co_ds<MT, HGS, FIVE, GEN, PLT, SPW && LL::KVS && DFF_QSP>(geo);
                    wg_sync<SPILL>();
                    pf.tick(17);
                    if (FIVE) {
                        co_dv_dk<MT, HGS, false, GEN, SPW && LL::KVS, SPW && LL::VSP>(geo);
                    } else {
                        if constexpr (MT == 4 && HGS == 1 && !GEN && DFF_DQKV_ROWS) {
                            co_dqkv_rows<MT, 0, PLT>(geo);
                            wg_sync<SPILL>();
                            co_dqkv_rows<MT, 1, PLT, SPW && DFF_PSPLIT>(geo);
                            wg_sync<SPILL>();
                            co_dqkv_rows<MT, 2, PLT, SPW && DFF_PSPLIT>(geo);
                        } else {
                        co_dqkv<MT, HGS, 0, GEN, PLT>(geo);
                        wg_sync<SPILL>();
                        co_dqkv<MT, HGS, 1, GEN, PLT>(geo);
                        wg_sync<SPILL>();
                        co_dqkv<MT, HGS, 2, GEN, PLT>(geo);
                        }
                    }
                    wg_sync<SPILL>();
                    pf.tick(18);
                    // d(LN1 out) += [dq|du] W_qu + dk W_k + dv W_v   (K order per head [q64|u16|k64|v64])
                    if (hg + 1 < hg_hi) l2_wgx(lw, hg + 1);
                    else if (l > 0) l2_w2t(m.layer[l - 1], ch_lo);
                    if constexpr (SPW) {
                        u32x4 bq[4][NTW][3];
                        ExtW<NTW, HGS> ew;
                        if constexpr (DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 13 + 4; }, lw.WqkvxT_p, DFF_HEADS * 13, NT_H);
                        {
                            constexpr bool RP = MT == 4 && HGS == 1 && !GEN && DFF_DQKV_ROWS && DFF_PSPLIT;   // co_dqkv_rows<..., PSPLIT>
                            gemm_tall_qkvT_split<MT, NTW, HGS, LL::KVS || RP, LL::VSP, 0, (LL::KVS && DFF_QSP) || RP, QT16>(acc_a, geo.Rg, FIVE ? 4 : 3, RN, lw.WqkvxT_s, hg * HGS, NT_H, geo.lsp, bq, geo.lsq, geo.qs);
                        }
                        if constexpr (!DFF_EXTPRE) ext_fetch<NTW, HGS, MT>(ew, [=](int i) { return (hg * HGS + i) * 13 + 4; }, lw.WqkvxT_p, DFF_HEADS * 13, NT_H);
                        ext_apply<MT, NTW, HGS>(acc_a, ew, [=](int i) { return (FIVE ? 4 : 3) * RN * LQ + i * 80 + 64; }, geo.Rg, LQ, RN, NT_H, QT16 ? geo.qs : 1.0f);
                    } else
                    gemm_tall_kb_st<MT, NTW, 13, 13 * HGS>(acc_a,
                        [=](int i, int& aoff, int& wkb) {
                            const int hh = i / 13, tt = i - 13 * hh;
                            const int part = (tt >= 5) + (tt >= 9);
                            const int reg = part == 0 ? (FIVE ? 4 : 3) : part;   // dQ_ext in buffer 4 (or R3), dK in R1, dV in R2
                            aoff = reg * RN * LQ + hh * 80 + 16 * (tt - 5 * part + (part >> 1));
                            wkb = (hg * HGS + hh) * 13 + tt;
                        },
                        geo.Rg, LQ, RN, lw.WqkvxT_p, DFF_HEADS * 13, NT_H);
                } else {
                    co_ds<MT, HGS, false, GEN, PLT>(geo);
                    wg_sync<SPILL>();
                    pf.tick(17);
                    co_dv_dk<MT, HGS, true, GEN, false, false, NoStepHook, PLT>(geo);
                }
                wg_sync<SPILL>();
                pf.tick(19);
            }
            if (l > 0 || full0) {
                store_tall<MT, NTW>(acc_a, tbuf, LH, rows, NT_H, nullptr);
                pair_exchange(tbuf, H, LH);
                wg_sync<SPILL>();
                rowb_ln1<H, LPG>(c, lw, l, tbuf, QT16 ? geo.qsi : 1.0f);
                wg_sync<SPILL>();
                pf.tick(20);
            }
        }
        // dE/dx = sum of the per-wave partials (the force head wrote dxs itself)
        if (tid < rows * 4 && m.conservative) {
            float sdx = 0.f;
#pragma unroll
            for (int w = 0; w < DFF_NWAVES; ++w) sdx += (smem + ll.dxw)[w * RN * 4 + tid];
            c.dxs[tid] = sdx;
        }
        if constexpr (PAIR) {   // this block's heads' share of dE/dx + the partner's (rows x 4 floats)
            if (m.conservative) pair_exchange(c.dxs, 4, 4);
        }
        wg_sync<SPILL>();
        if (full0 && m.conservative) {
            node_embed_bwd<H, LPG>(c, m);
            wg_sync<SPILL>();
        }

        // =============================== update ===============================
        // dxs = d(sum E)/dx ; the score op returns -dxs (graph_transformer.py:159)
        if (a.mode == DFF_MODE_SCORE) {
            if (tid < rows * 4 && (tid & 3) < 3)
                a.force_out[((size_t)c.b0 * N + (tid >> 2)) * 3 + (tid & 3)] = -c.dxs[tid];
        } else if (a.mode == DFF_MODE_LANGEVIN) {
            const bool save = ((step + 1) % a.save_interval) == 0;
            const int fi = (step + 1) / a.save_interval - 1;
            if (tid < rows * 4 && (tid & 3) < 3) {
                const int row = tid >> 2, cc = tid & 3;
                const int g = row / N, i = row - g * N;
                const size_t item = (size_t)c.b0 + g;
                float xi;
                if (a.noise) xi = a.noise[(((size_t)step * a.B + item) * N + i) * 3 + cc];
                else xi = philox_normal(a.seed, a.item_offset + item, a.step_offset + step, i, cc);
                // forces = -GNN(x) / kbt_inv / sigma_t = dxs * force_scale   (langevin.py:79-87)
                const float f = c.dxs[tid] * a.force_scale;
                const float x = c.xst[tid];
                float xn, vn = 0.f;
                if (a.overdamped) {  // langevin_cgnet.py:481-500
                    xn = x + f * a.dtau + a.brown_sigma * xi;
                } else {             // BAOA(F)B, langevin_cgnet.py:447-479
                    vn = c.vst[tid] + (a.dt * f) / a.mass[i];
                    xn = x + (vn * a.dt) / 2.0f;
                    const float nz = a.noise_sigma[i] * xi;
                    vn = vn * a.vscale;
                    vn = vn + a.noisescale * nz;
                    xn = xn + (vn * a.dt) / 2.0f;
                }
                c.xst[tid] = xn;
                c.vst[tid] = vn;
                if (save && a.frames) a.frames[(((size_t)fi * a.B + item) * N + i) * 3 + cc] = xn;
            }
            wg_sync<SPILL>();
            if (save && a.ke && !a.overdamped && tid < c.gcnt) {  // langevin_cgnet.py:538-542
                float ke = 0.f;
                for (int i = 0; i < N; ++i) {
                    const float* vp = c.vst + (tid * N + i) * 4;
                    ke += a.mass[i] * (vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]);
                }
                a.ke[(size_t)fi * a.B + c.b0 + tid] = 0.5f * ke;
            }
        } else {  // DDPM p_sample (ddpm.py:195-232) + clamp/centre of p_sample_loop (:248-251)
            const bool act = tid < rows * 4 && (tid & 3) < 3;
            const int row = tid >> 2, cc = tid & 3;
            const int g = act ? row / N : 0, i = act ? row - g * N : 0;
            const size_t item = (size_t)c.b0 + g;
            float eps = act ? -c.dxs[tid] : 0.f;
            float xi = 0.f;
            if (act) {
                if (a.noise) xi = a.noise[(((size_t)step * a.B + item) * N + i) * 3 + cc];
                else xi = philox_normal(a.seed, a.item_offset + item, (uint64_t)t_int, i, cc);
            }
            // centre eps and noise (two bead-means)
            if (tid < rows * 4) { c.dxs[tid] = eps; c.xs[tid] = xi; }
            wg_sync<SPILL>();
            bead_mean(c, c.dxs, c.cm);
            bead_mean(c, c.xs, c.cm + 64);
            wg_sync<SPILL>();
            const float x = act ? c.xst[tid] : 0.f;
            float x0 = 0.f;
            if (act) {
                eps -= c.cm[g * 4 + cc];
                xi -= c.cm[64 + g * 4 + cc];
                x0 = m.sqrt_recip_ac[t_int] * x - m.sqrt_recipm1_ac[t_int] * eps;  // ddpm.py:140-147
            }
            wg_sync<SPILL>();
            if (tid < rows * 4) c.dxs[tid] = x0;
            wg_sync<SPILL>();
            bead_mean(c, c.dxs, c.cm);
            wg_sync<SPILL>();
            float xn = 0.f;
            if (act) {
                x0 -= c.cm[g * 4 + cc];
                const float mean = m.post_c1[t_int] * x0 + m.post_c2[t_int] * x;  // ddpm.py:149-161
                const float nzm = (t_int == 0) ? 0.f : 1.f;
                xn = mean + nzm * expf(0.5f * m.post_logvar[t_int]) * xi;
                if (xn > 1000.f || xn < -1000.f) {
                    if (a.clamp_flag) atomicOr(a.clamp_flag, 1);
                    xn = fminf(fmaxf(xn, -1000.f), 1000.f);
                }
            }
            wg_sync<SPILL>();
            if (tid < rows * 4) c.dxs[tid] = xn;
            wg_sync<SPILL>();
            bead_mean(c, c.dxs, c.cm);
            wg_sync<SPILL>();
            if (act) c.xst[tid] = xn - c.cm[g * 4 + cc];
        }
        wg_sync<SPILL>();
        // end of a reverse chain: the reference's assert_center_zero(mol) (models/ddpm.py:252, utils.py:73-86) on the device:
        // bit 1 of the flag word reports a centre of mass >= 1e-3 (bit 0: the +-1000 clamp); the host raises on it
        if (a.mode == DFF_MODE_DDPM && step == a.n_steps - 1 && a.clamp_flag) {
            bead_mean(c, c.xst, c.cm);
            wg_sync<SPILL>();
            if (tid < c.gcnt * 4 && (tid & 3) < 3 && !(fabsf(c.cm[tid]) < 1e-3f)) atomicOr(a.clamp_flag, 2);
        }
        pf.tick(21);
    }
    if (pf.on)
        for (int i = 0; i < DFF_NPROF; ++i) a.prof[i] = pf.acc[i];

    // ---- write back state ----
    if (a.mode != DFF_MODE_SCORE && tid < rows * 4 && (tid & 3) < 3) {
        const size_t gi = ((size_t)c.b0 * N + (tid >> 2)) * 3 + (tid & 3);
        a.x_io[gi] = c.xst[tid];
        if (a.mode == DFF_MODE_LANGEVIN && !a.overdamped) a.v_io[gi] = c.vst[tid];
    }
}

// ------------------------------------------------------------------------------------------
// debug GEMM kernel: one wide GEMM stage through the same device routine and packing
// ------------------------------------------------------------------------------------------
template <int KB>
__global__ __launch_bounds__(DFF_NTHREADS) void dff_debug_gemm_kernel(const float* A, const float* Wp, int M,
                                                                     int Nout, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 16 * KB, LD = K + 4;
    for (int i = threadIdx.x; i < 64 * LD + 64; i += DFF_NTHREADS) smem[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < M * K; i += DFF_NTHREADS) smem[(i / K) * LD + (i % K)] = A[i];
    __syncthreads();
    gemm_wide<4, KB, 1>((const lfloat*)smem, LD, M, Wp, KB, 0, 0, Nout / 16,
        [=](int, float (&)[1]) {},
        [=](int nt, int mt, const f32x4& acc, const float (&)[1], bool, int) {
            const int lane = threadIdx.x & 63, col = 16 * nt + 4 * (lane >> 4), row = mt * 16 + (lane & 15);
            if (row < M) *(f32x4*)(out + row * Nout + col) = acc;
        });
}

// ------------------------------------------------------------------------------------------
// variant table handed to the host dispatcher (dff_host.hip); taking the kernels' addresses instantiates them
// ------------------------------------------------------------------------------------------
// how the split variants name their engine: every weight GEMM on the two-piece fp16 format, or (any other DFF_F16G) "split_bf16"
#if DFF_F16G == 15
#define DFF_SPN "split_f16"
#else
#define DFF_SPN "split_bf16"
#endif
template <int H, int MT, int HGS, bool SP, bool SPW>
static unsigned lds_floats_of(int N, int G) {
    // The TIGHT tile arrays (LdsLayout: 16 MT - 4 = 60 columns, no pad rows) are correct for at most 16 MT - 4 bead rows:
    // the k-step over columns 60..63 is skipped because it would only hold pad beads.  Today the LDS budget already turns
    // away 57+ rows; say so explicitly instead of relying on it: "does not fit", the dispatcher moves on.
    using LL = LdsLayout<H, MT, HGS, SP>;
    if (SPW && LL::TIGHT_OK && G * N > LL::PL_TIGHT) return 0x3fffffffu;   // (x 4 bytes still fits 32 bits)
    return LL(N, G, SPW).total;
}
#define VAR(H, MT, HGS, SP)                                                                                     \
    { H, MT, HGS, SP, false, false, (const void*)&dff_fused_kernel<H, MT, HGS, SP, false, false>,               \
      &lds_floats_of<H, MT, HGS, SP, false>, "dff_fused_kernel<" #H "," #MT "," #HGS "," #SP ">" },             \
    { H, MT, HGS, SP, true, false, (const void*)&dff_fused_kernel<H, MT, HGS, SP, true, false>,                 \
      &lds_floats_of<H, MT, HGS, SP, false>, "dff_fused_kernel<" #H "," #MT "," #HGS "," #SP ",gen>" }
#define VAR_SPW(H, MT, HGS)                                                                                     \
    { H, MT, HGS, false, false, true, (const void*)&dff_fused_kernel<H, MT, HGS, false, false, true>,           \
      &lds_floats_of<H, MT, HGS, false, true>, "dff_fused_kernel<" #H "," #MT "," #HGS ",false," DFF_SPN ">" },  \
    { H, MT, HGS, false, true, true, (const void*)&dff_fused_kernel<H, MT, HGS, false, true, true>,             \
      &lds_floats_of<H, MT, HGS, false, true>, "dff_fused_kernel<" #H "," #MT "," #HGS ",false,gen," DFF_SPN ">" }
#define VAR_SPW_SPILL(H, MT, HGS)                                                                               \
    { H, MT, HGS, true, false, true, (const void*)&dff_fused_kernel<H, MT, HGS, true, false, true>,             \
      &lds_floats_of<H, MT, HGS, true, true>, "dff_fused_kernel<" #H "," #MT "," #HGS ",true," DFF_SPN ">" }
#define VAR_PAIR_SPW_SPILL(H, MT, HGS)                                                                            \
    { H, MT, HGS, true, false, true, (const void*)&dff_fused_kernel<H, MT, HGS, true, false, true, true>,         \
      &lds_floats_of<H, MT, HGS, true, true>, "dff_fused_kernel<" #H "," #MT "," #HGS ",true," DFF_SPN ",pair>", true }
#define VAR_PAIR(H, MT, HGS, SP)                                                                                  \
    { H, MT, HGS, SP, false, false, (const void*)&dff_fused_kernel<H, MT, HGS, SP, false, false, true>,           \
      &lds_floats_of<H, MT, HGS, SP, false>, "dff_fused_kernel<" #H "," #MT "," #HGS "," #SP ",pair>", true }
#define VAR_PAIR_SPW(H, MT, HGS)                                                                                  \
    { H, MT, HGS, false, false, true, (const void*)&dff_fused_kernel<H, MT, HGS, false, false, true, true>,       \
      &lds_floats_of<H, MT, HGS, false, true>, "dff_fused_kernel<" #H "," #MT "," #HGS ",false," DFF_SPN ",pair>", true }
static const Variant g_variants[] = {
#ifndef DFF_FAST_BUILD
    VAR(64, 1, 4, false),  VAR(64, 2, 2, false),  VAR(96, 1, 4, false),  VAR(96, 2, 2, false),
    VAR(128, 1, 4, false), VAR(128, 2, 2, false), VAR(128, 3, 1, false), VAR(128, 4, 1, true),
    // hidden = 256 (the reference's own smoke test, models/graph_transformer.py:332-359; no shipped checkpoint): fp32 engine,
    // up to 32 bead rows (the second shape keeps the residual stream in the stash to fit the LDS)
    VAR(256, 1, 4, false), VAR(256, 2, 1, true),
    VAR_SPW(96, 2, 2), VAR_SPW(128, 2, 2), VAR_SPW(128, 3, 1), VAR_SPW_SPILL(128, 4, 1),
    VAR_SPW(64, 1, 4), VAR_SPW(96, 1, 4), VAR_SPW(128, 1, 4),
    VAR_PAIR(128, 4, 1, true), VAR_PAIR_SPW(128, 3, 1), VAR_PAIR_SPW(128, 2, 2), VAR_PAIR_SPW(96, 2, 2), VAR_PAIR_SPW_SPILL(128, 4, 1), VAR_PAIR_SPW(96, 1, 4), VAR_PAIR_SPW(128, 1, 4),
#elif defined(DFF_ONLY)   // development builds: one named variant, e.g. -DDFF_ONLY="VAR_SPW(128,3,1)"
    DFF_ONLY,
#else   // development builds: one variant, so that the <= 16-row kernel can be iterated on quickly
    VAR(64, 1, 4, false),
#endif
};
int dff_fused_f16_mask() { return DFF_F16G; }
const Variant* dff_fused_variants(int* count) {
    *count = (int)(sizeof(g_variants) / sizeof(g_variants[0]));
    return g_variants;
}
int dff_debug_gemm_launch(int K, const float* dA, const float* dW, int M, int Nout, float* dO, size_t lds) {
    if (K == 64) hipLaunchKernelGGL(dff_debug_gemm_kernel<4>, dim3(1), dim3(DFF_NTHREADS), lds, 0, dA, dW, M, Nout, dO);
    else hipLaunchKernelGGL(dff_debug_gemm_kernel<8>, dim3(1), dim3(DFF_NTHREADS), lds, 0, dA, dW, M, Nout, dO);
    return (int)hipGetLastError();
}
