// dff_small.hip -- fast path of the fused sampler kernel for workgroups of <= 16 bead rows
// (chignolin 10 beads, alanine dipeptide 5 beads x up to 3 proteins, any N <= 16 model).
//
// Same algorithm and stash contract as dff_fused_kernel (dff_kernels.hip); what changes is the
// work decomposition, chosen so that nothing but the row-wise LayerNorm / gate stages needs a
// workgroup barrier:
//
//   * HEADS ARE WAVE-PRIVATE.  With NW = 8 waves wave w owns head w, with NW = 4 heads {w, w+4}: it
//     computes that head's q|u|k|v tiles, the logits, softmax, the P.V product and the head's slice of
//     the output projection entirely out of its private LDS region; the waves' partial output
//     projections (a K-split by head) are summed by the next row stage.  The FFN is split the same way (each wave owns
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
// The attention products run on v_mfma_f32_16x16x4_f32, the weight GEMMs of the SPW variants on v_mfma_f32_16x16x32_bf16
// through an exact three-way bf16 split (split engine below); VALU work is softmax (in the MFMA C layout), LayerNorm,
// gates, GELU and the integrator update.  One kernel per sampler mode (MODE template argument).
// The FOLD variant (hidden == head dimension, <= 3 layers: chignolin) keeps everything a step produces in LDS / registers
// in the sampling loops: no stash traffic (SmallLds<..., FOLD>, KEEPROWS).
#include "dff_device.h"

#ifdef DFF_MARKS   // development: stage markers ("; ##MARK <tick id>") in the ISA listing, to bin spills / waits by stage
#define DFF_MARK(n) asm volatile("; ##MARK " #n)
#else
#define DFF_MARK(n) ((void)0)
#endif
// DFF_F16 (round 5): the split variants run their weight GEMMs on the TWO-piece fp16 split (dff_device.h split8h): stream kind -1 =
// host-split fp16 image, 4 B per weight, 2 KB units, three v_mfma_f32_16x16x32_f16 per unit.  1: the FOLD variant only (chignolin:
// the headline), 2: every split variant of this kernel, 0: the three-piece bf16 engine of rounds 2-4.
#ifndef DFF_F16
#define DFF_F16 2
#endif
#define DFF_F16_ON(FOLD_) (DFF_F16 >= 2 || (DFF_F16 == 1 && (FOLD_)))
#ifndef DFF_SDR
#define DFF_SDR 4   // split-ring depth in units (SPW variants)
#endif
// NW = waves per workgroup.  NW = 4: one wave per SIMD, two heads per wave, 16-row head buffers.
// NW = 8: two waves per SIMD (each hides the other's stalls), one head per wave; to fit 8 wave
// regions in 160 KB the head buffers hold RLA = 11 rows (10 real + 1 dummy row that absorbs the
// pad lanes' stores; MFMA operand reads of rows 11..15 run into the next buffer, which is finite
// data multiplied by exact zeros of P / dS or landing in discarded output rows).
// NWR: the waves that are really there.  PAIR variant (round 6: two workgroups per protein, FOLD kernel only): NW = 8 stays the
// DECOMPOSITION (eight heads, eight FFN slices, the 11-row layout), each workgroup runs four of them on NWR = 4 waves.
template <int H, int NW, bool FOLD = false, int NWR = NW>
struct SmallLds {
    static constexpr int LH = H + 4;
    static constexpr int RLA = NW == 4 ? 16 : 11;
    // Row stride of the head buffers.  84 floats: 80 + one 16-byte pad slot.  FOLD: 88 -- 8 dwords mod 64, which makes the
    // ds_read_b128 operand fragments of the attention products conflict-free (a read group is rows {0-3, 12-15} of one
    // k-group and rows {4-11} of the next: bank slots (2 row + kg) mod 16 all even | all odd; with 84 row 11 of one k-group
    // lands on row 12 of the other) -- and leaves 20 spare columns per row, where the saved P tiles of layers 0 and L - 2 live.
    static constexpr int XLD = FOLD ? 88 : DFF_XLD;
    static constexpr int RS = RLA * XLD;             // floats of one head buffer (Q / K / V / G)
    static constexpr int PT = 16 * DFF_PLD;          // ... of one P / dS tile
    // FOLD variant (keys = values = the shared LayerNorm rows, dff_small_kernel below): no K / V regions.  A wave region is
    //   [Q | P | G | dS | Qsave | GP0 | GP1]
    // Qsave: q' of the layer before the last one parked between its forward and backward attention blocks (KEEP2), and, in
    // the spare columns 68.. of its rows, the 10 x 10 real entries of that layer's and of layer 0's softmax tiles; GP0 | GP1: GELU'(h_pre) of this wave's FFN hidden slice for the two layers before the last one, GPR rows x
    // GPS floats each (row GPR - 1 absorbs the pad lanes' stores) -- the last layer's tile lives behind the wave's 11-row
    // partial-sum tile inside G | dS, so that in the sampling loops of a <= 3-layer model GELU' never leaves the LDS (it was
    // 57 % of the kernel's stash traffic, profiles/r02).  The one shared copy of the LayerNorm rows (K_ext = V_ext) has its
    // own region `nx`; the fp32 K = H GEMM input `abuf` does not exist (the row stages write bf16 pieces into `asp`).
    static constexpr int GPR = 11, GPS = 32;
    static constexpr unsigned GPT = GPR * GPS;
    static constexpr unsigned GP_LAST = RLA * LH;   // offset of the last layer's tile inside G | dS (behind the partial-sum tile)
    static_assert(!FOLD || GP_LAST + GPT <= RS + PT, "the last layer's GELU' tile must fit G | dS behind the partial sums");
    static constexpr unsigned WREG = FOLD ? 3 * RS + 2 * PT + 2 * GPT : 4 * RS + 2 * PT;   // floats per wave region
    static_assert(RS + PT >= 16 * (H + 4) || !FOLD, "G | dS must hold a 16 x (H+4) tile");
    static_assert(WREG >= 16 * (H + 4), "wave region must hold a 16 x (H+4) partial-sum tile");
    static constexpr int DMA_N = FOLD ? 6 : 13;      // head_dma: global_load_lds instructions per head ([Q | P] or [Q | K | V | P])
    static constexpr unsigned xst = 0, xs = 64, dxs = 128, vst = 192, cm = 256, tn = 384, prof = 400,
                              dxw = 448,                      // [NW][128] per-wave dx partials (+ dummies)
                              abuf = 448 + NWR * 128, resbuf = abuf + (FOLD ? 0 : 16 * LH),
                              // SPW variants (8 waves): the K = H GEMM input as bf16 pieces, written by the row stages:
                              // [piece 3][k-block H/32][kg 4][row 16][4 dwords] -- a wave's ds_read_b128 of (row, kg)
                              // then touches 16 rows x 4 dwords = every bank once, whatever the lane group
                              asp = resbuf + (FOLD ? RLA : 16) * LH, asp_size = (NW == 8 && H == 64) ? ((FOLD && DFF_F16_ON(true)) ? 2 * (H / 32) * 256 + 64 : 3 * (H / 32) * 256) : 0,   // (only H = 64 has SPW variants; the fp16 engine has two pieces + the row scales)
                              // source-offset table of the LDS-DMA head fetch (head_dma): DMA_N instructions x 64 lanes
                              dmatab = asp + asp_size, dmatab_size = (NW == 8 && H == 64) ? DMA_N * 64 : 0,
                              // FOLD: the fp16 pieces of the LayerNorm rows `nx` holds in fp32 ([h | l'], the layout of `asp`): the A operand
                              // of the QKV' GEMM and an operand of the logits (forward) and of dA (backward) -- kept apart from `asp`,
                              // which the other row stages overwrite before the backward attention block of the layer comes round
                              nsp = dmatab + dmatab_size, nsp_size = FOLD ? 2 * (H / 32) * 256 : 0,
                              // ... and the same rows' 64 regular columns once more, TRANSPOSED: [h | l'][16-column tile][4 rows kg][column][row & 3] --
                              // lane (column m, kg) of a v_mfma_f32_16x16x16_f16 reads its four contraction-index values (rows 4 kg ..) as
                              // 8 bytes: the operand of O^T = V^T P^T and of dQ^T = K^T dS^T on the fp16 pipe (round 6)
                              nst = nsp + nsp_size, nst_size = (FOLD && DFF_F16_ON(true)) ? 2 * (H / 16) * 4 * 16 * 2 : 0,
                              // (nx stays in front of the wave regions: operand reads of its rows 11..15 land in wave 0's Q region, fp32 data)
                              nx = nst + nst_size, nx_size = FOLD ? RS : 0,
                              wreg = nx + nx_size, total = wreg + NWR * WREG + 64;
    static_assert(!FOLD || total * 4 <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------- wave-private MFMA engine
// One wave per SIMD means nothing hides a stall except the wave's own instruction stream, so the
// per-wave GEMM streams are written to be straight-line: tile loops fully unrolled, epilogues
// branch-free (pad lanes store to a dummy stash row / harmless LDS pad rows), and the weight
// operands come out of a register ring of DFF_DR entries that is kept full ACROSS GEMMs: the tail
// of one GEMM refills the ring with the head of the next, and every block ends by prefetching the
// first entries of the block that follows the barrier.  Every GEMM of this kernel has either
// K = H (wide: one entry = one 16-column output tile = H/16 k-blocks) or Nout = H (tall: one
// entry = one k-block for all H/16 output tiles), so all ring entries are E = H/16 float4.
// Lane-derived address offsets are loop-invariant, so LICM hoists every one of them out of the
// step loop and keeps hundreds of VGPRs live for the whole kernel (-> scratch spills).  Reading
// the lane id through an opaque asm makes each block re-derive its offsets locally (a few VALU
// ops) instead.
DEVI int lane_id() {
    int x = threadIdx.x & 63;
    asm volatile("" : "+v"(x));
    return x;
}
DEVI int tid_id() {
    int x = threadIdx.x;
    asm volatile("" : "+v"(x));
    return x;
}

template <int E, int DR>
struct Ring {
    f32x4 b[DR][E];
};
struct WStream {
    const gf32x4* base; // wave-uniform
    int estep;          // f32x4 stride between consecutive entries
    int pstride;        // f32x4 stride between the E parts of one entry
};
// base stays wave-uniform (SGPR); the lane offset is added at the load
DEVI WStream wide_stream(const float* Wp, int KBtot, int nt0) {
    return WStream{(const gf32x4*)Wp + (size_t)nt0 * KBtot * 64, KBtot * 64, 64};
}
DEVI WStream tall_stream(const float* Wp, int KBtot, int kw0) {
    return WStream{(const gf32x4*)Wp + (size_t)kw0 * 64, 64, KBtot * 64};
}
template <int E>
DEVI void ring_fill(f32x4 (&slot)[E], const WStream& w, int j, int lane) {
    // uniform 64-bit base (SALU) + zero-extended 32-bit lane offset: the saddr form of global_load, no 64-bit VALU adds
    const gf32x4* p = w.base + (size_t)j * w.estep;
    const unsigned lo = (unsigned)lane & 63u;   // known bits: lo << 4 cannot wrap
#pragma unroll
    for (int e = 0; e < E; ++e) slot[e] = (p + (size_t)e * w.pstride)[lo];
}
template <int E, int DR>
DEVI void ring_prefetch(Ring<E, DR>& r, const WStream& w, int lane) {
#pragma unroll
    for (int j = 0; j < DR; ++j) ring_fill<E>(r.b[j], w, j, lane);
}

// wide GEMM of one wave: N output tiles (entries), A fragments `a` (K = 16 E) preloaded.
// Ring phase PH = slot of entry 0.  Tail refills come from `wn`, the next GEMM's stream.
// aux[slot][..]: per-tile epilogue operands (bias / stashed values); the caller preloads the first
// DFF_DR tiles' worth, pre(t, aux_slot) refills.  The tile loop is ONE ring revolution unrolled
// (static slots) inside a rolled loop, which keeps the live register set small.  Pre / Epi are
// by-value functors (capture with [=]): a closure that ends up in memory costs a scratch reload
// per tile, and a scratch reload is a VMEM op -- the compiler then waits vmcnt(0) and drains the
// whole ring every tile.
template <int SLOT, int N, int E, int NAUX, int DR, class Pre, class Epi>
DEVI void wide_tile(Ring<E, DR>& ring, float (&aux)[DR][NAUX], const f32x4 (&a)[E], const WStream& w,
                    const WStream& wn, int lane, int t, const Pre& pre, const Epi& epi) {
    f32x4 (&b)[E] = ring.b[SLOT];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < E; kb += 2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb][s], b[kb][s], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb + 1][s], b[kb + 1][s], acc2, 0, 0, 0);
        }
    float auxc[NAUX];
#pragma unroll
    for (int q = 0; q < NAUX; ++q) auxc[q] = aux[SLOT][q];
    if (t + DR < N) { ring_fill<E>(b, w, t + DR, lane); pre(t + DR, aux[SLOT]); }
    else ring_fill<E>(b, wn, t + DR - N, lane);
    epi(t, acc + acc2, auxc);
}
// NOTE: aux is indexed by ring SLOT (callers preload aux[(PH + d) % DR] for tile d).  DR (ring depth,
// 2 or 4) is deduced from the ring argument.
template <int PH, int N, int E, int NAUX, int DR, class Pre, class Epi>
DEVI void wide_run(Ring<E, DR>& ring, float (&aux)[DR][NAUX], const f32x4 (&a)[E], const WStream& w,
                   const WStream& wn, int lane, const Pre pre, const Epi epi) {
    static_assert(E % 2 == 0, "E even");
    static_assert(DR == 2 || DR == 4, "ring depth");
    constexpr int NREV = N / DR, REM = N % DR;
#pragma unroll 1
    for (int rev = 0; rev < NREV; ++rev) {
        const int t0 = rev * DR;
        wide_tile<(PH + 0) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, t0, pre, epi);
        wide_tile<(PH + 1) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, t0 + 1, pre, epi);
        if constexpr (DR == 4) {
            wide_tile<(PH + 2) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, t0 + 2, pre, epi);
            wide_tile<(PH + 3) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, t0 + 3, pre, epi);
        }
    }
    if constexpr (REM > 0) wide_tile<(PH + 0) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, NREV * DR, pre, epi);
    if constexpr (REM > 1) wide_tile<(PH + 1) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, NREV * DR + 1, pre, epi);
    if constexpr (REM > 2) wide_tile<(PH + 2) % DR, N, E, NAUX>(ring, aux, a, w, wn, lane, NREV * DR + 2, pre, epi);
}

// tall GEMM of one wave: acc[nt] += A(:, k-block kb) . W(entry kb, tile nt), kb = 0..N-1
// The A fragment of k-block kb + 1 is read from LDS BEFORE the MFMAs of k-block kb are issued (`a` is the
// fragment of this step, loaded one step ago): otherwise every step starts with an exposed LDS round trip.
// XKB >= 0: k-block XKB (> 0) is a head's extension block, whose weight rows 4..15 are zero: one k-step over its
// columns 0..3 instead of four (the packed image holds row lane >> 4 in k-step 0, dff_host.hip pack_b).
template <int SLOT, int N, int E, int XKB, int DR, class FA>
DEVI void tall_step(Ring<E, DR>& ring, f32x4 (&acc)[E], f32x4& a, const FA& fa, const WStream& w, const WStream& wn, int lane, int kb) {
    f32x4 (&b)[E] = ring.b[SLOT];
    const lfloat* const pn = fa(kb + 1 < N ? kb + 1 : N - 1);
    f32x4 an;
    if (XKB >= 0 && kb + 1 == XKB) an[0] = pn[-3 * (lane >> 4)];   // extension block: column lane >> 4 (see pack_b)
    else an = *(const lf32x4*)pn;
#pragma unroll
    for (int nt = 0; nt < E; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[nt][0], acc[nt], 0, 0, 0);
    if (XKB < 0 || kb != XKB) {
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < E; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[nt][s], acc[nt], 0, 0, 0);
    }
    if (kb + DR < N) ring_fill<E>(b, w, kb + DR, lane);
    else ring_fill<E>(b, wn, kb + DR - N, lane);
    a = an;
}
template <int PH, int N, int E, int XKB, int DR, class FA>
DEVI void tall_run(Ring<E, DR>& ring, f32x4 (&acc)[E], const FA fa, const WStream& w, const WStream& wn, int lane) {
    constexpr int NREV = N / DR, REM = N % DR;
    f32x4 a = *(const lf32x4*)fa(0);
#pragma unroll 1
    for (int rev = 0; rev < NREV; ++rev) {
        const int k0 = rev * DR;
        tall_step<(PH + 0) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, k0);
        tall_step<(PH + 1) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, k0 + 1);
        if constexpr (DR == 4) {
            tall_step<(PH + 2) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, k0 + 2);
            tall_step<(PH + 3) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, k0 + 3);
        }
    }
    if constexpr (REM > 0) tall_step<(PH + 0) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, NREV * DR);
    if constexpr (REM > 1) tall_step<(PH + 1) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, NREV * DR + 1);
    if constexpr (REM > 2) tall_step<(PH + 2) % DR, N, E, XKB>(ring, acc, a, fa, w, wn, lane, NREV * DR + 2);
}

// ---------------------------------------------------------------- split-bf16 engine (SPW variants)
// The weight GEMMs on the bf16 matrix pipe at fp32 accuracy: an fp32 value is the exact sum of three bf16 pieces
// (truncation split h | m | l) and  a.b ~ ah.bh + (am.bh + ah.bm) + (al.bh + ah.bl + am.bm)  drops only terms of relative
// order 2^-24 (tools_ubench/split_bf16.hip: error vs fp64 <= that of v_mfma_f32_16x16x4_f32).  Six
// v_mfma_f32_16x16x32_bf16 (16 cycles each, and they leave the SIMD's vector port to the sibling wave) replace eight
// v_mfma_f32_16x16x4_f32 (32 cycles each).  The weights are split on the host (dff_host.hip pack_units); the A operand
// stays fp32 in LDS, exactly where the fp32 engine reads it, and is split in registers by the wave that consumes it
// (split8: ~44 VALU per 32-column block, amortised over the 4..13 tiles that use the block).
// One ring entry = one UNIT = (16-column output tile, 32-row k-block) = three pieces x 16 B per lane; every stream is
// a linear sequence of units ([tile][k-block] for the K = H GEMMs, [k-block][tile] for the Nout = H ones), the ring holds
// DR = H/16 units, and the last DR refills of a GEMM fetch the first units of the stream that follows, placed so that
// the following GEMM starts at ring phase PHN (blocks separated by a barrier always start at phase 0).
template <int DR>
struct SRing {
    u32x4 b[DR][3];
};
struct SStream {
    const gu32x4* base;   // wave-uniform; unit j at base + 192 j, piece p at + 64 p, lane at + lane
};
DEVI SStream sstream(const unsigned* Wp, int unit0, int ust = 192) { return SStream{(const gu32x4*)Wp + (size_t)unit0 * ust}; }
DEVI void sfill(u32x4 (&slot)[3], const gu32x4* p, int lane) {
    const unsigned lo = (unsigned)lane & 63u;
#pragma unroll
    for (int q = 0; q < 3; ++q) slot[q] = (p + 64 * q)[lo];
}
// The weights a wave needs between two workgroup barriers form ONE sequence of units: N0 units of stream s0 followed by
// N1 units of s1 (e.g. QKV_ext then [W_o;W_oc]; W1 then W2), and after them the first DR units of the NEXT block (M0
// units of n0, then units of n1), which the last DR refills of this block request -- so they are in flight during the row
// stage in between.  Unit i of the block lives in ring slot i % DR; all indices are compile-time (every GEMM below is
// fully unrolled, which also lets the compiler count the outstanding loads exactly: in a rolled loop the epilogue operands
// requested two tiles ahead are loop-carried, and the s_waitcnt that joins them drains the whole ring every revolution).
// eight fp32 values of a 32-column block (x0: columns 4 kg .. + 3, x1: columns 16 + 4 kg .. + 3 -- what two ds_read_b128
// of the fp32 engine's A pattern deliver) -> the three bf16 piece operands; element j sits in half j & 1 of dword j >> 1
template <int KB32>
DEVI void split_afrag(const f32x4 (&a)[2 * KB32], u32x4 (&ah)[KB32], u32x4 (&am)[KB32], u32x4 (&al)[KB32]) {
#pragma unroll
    for (int kb = 0; kb < KB32; ++kb) split8(a[2 * kb], a[2 * kb + 1], ah[kb], am[kb], al[kb]);
}
// A stream is one of two KINDS.  Kind 0: a host-split image (three bf16 pieces per weight, 6 B; dff_host.hip pack_units),
// units linear.  Kind > 0: the fp32 image of the fp32 engine (4 B per weight; pack_b), whose two 16-row blocks of a unit
// the consuming wave splits in registers (split8, ~44 VALU per unit): fewer bytes for the GEMMs that are bound by the
// L2 -> CU weight stream (QKV_ext, its transpose, [W_o;W_oc]), where the SIMDs idle anyway.  Kind 1: K = H image
// ([tile][16-row block]: unit j at + 128 j); kind KBtot (> 1): Nout = H image with KBtot 16-row blocks per tile, unit
// (kb, nt) at + nt KBtot 64 + 64 blk(kb), blk(kb) = 2 kb, or 2 kb + 1 from kb = 2 on in the QKV_ext^T image (KBtot =
// 104), whose heads are [q 0..3 | ext 4 | k 5..8 | v 9..12].
template <int KIND, int E>
DEVI const gu32x4* unit_addr(const SStream& w, int j) {
    if constexpr (KIND == 0) return w.base + (size_t)j * 192;
    else if constexpr (KIND == 1 || KIND == -1) return w.base + (size_t)j * 128;   // (-1: fp16 image, two pieces per unit)
    else {
        const int kb = j / E, nt = j - kb * E;
        return w.base + (size_t)nt * KIND * 64 + (size_t)(2 * kb + (KIND == DFF_HEADS * 13 && kb >= 2 ? 1 : 0)) * 64;
    }
}
template <int KIND>
DEVI void sfill_k(u32x4 (&slot)[3], const gu32x4* p, int lane) {
    const unsigned lo = (unsigned)lane & 63u;
    slot[0] = p[lo];
    slot[1] = (p + 64)[lo];
    if constexpr (KIND == 0) slot[2] = (p + 128)[lo];
}
// a unit of a head's EXTENSION output tile ([u | s | 0 ...], [r | g_D | 0 ...]): only output columns 0..3 have weights, i.e.
// only lanes with (lane & 15) < 4 hold anything but zeros -- they alone load (256 B instead of 1 KiB per piece: these
// tiles are 1/13 of the QKV_ext stream and 1/5 of the [W_o;W_oc]^T stream)
template <int KIND = 0>
DEVI void sfill_ext(u32x4 (&slot)[3], const gu32x4* p, int lane) {
    const unsigned lo = (unsigned)lane & 63u;
    const bool has = (lane & 15) < 4;
#pragma unroll
    for (int q = 0; q < (KIND < 0 ? 2 : 3); ++q) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (has) v = (p + 64 * q)[lo];
        slot[q] = v;
    }
}
// the three piece operands of the unit held by a slot
template <int KIND>
DEVI void unit_pieces(const u32x4 (&slot)[3], u32x4& bh, u32x4& bm, u32x4& bl) {
    if constexpr (KIND < 0) { bh = slot[0]; bm = slot[1]; bl = slot[1]; }   // (fp16: h | l', no third piece)
    else if constexpr (KIND == 0) { bh = slot[0]; bm = slot[1]; bl = slot[2]; }
    else split8(__builtin_bit_cast(f32x4, slot[0]), __builtin_bit_cast(f32x4, slot[1]), bh, bm, bl);
}
template <int N0_, int N1_, int M0_, int K0_, int K1_, int KN0_, int KN1_, int E_, int XU_ = -1>
struct SSeq {
    static constexpr int N0 = N0_, N1 = N1_, M0 = M0_, K0 = K0_, K1 = K1_, KN0 = KN0_, KN1 = KN1_, E = E_;
    static constexpr int XU = XU_;   // units XU, XU + 1 of s0 (kind 0) are an extension output tile (sfill_ext), or -1
    SStream s0, s1, n0, n1;
    static constexpr int kind(int i) { return i < N0 ? K0 : K1; }   // of unit i of this block
};
// The ring's refill loads must be ISSUED where they are written: left alone, the scheduler sinks a load whose slot is free
// down to the products that consume it a ring's depth later (it saves the slot's live range -- and exposes an L2 latency
// per unit).  A scheduling barrier that everything but VMEM may cross pins the load without fencing the products.
#ifndef DFF_PIN
#define DFF_PIN 1
#endif
#if DFF_PIN
#define DFF_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x078F)
#else
#define DFF_PIN_VMEM() ((void)0)
#endif
template <int DR, int I, class Q>
DEVI void seq_refill(SRing<DR>& ring, const Q& q, int lane) {   // slot of unit I <- unit I + DR (or the next block's)
    constexpr int slot = I % DR, J = I + DR;
    if constexpr (J < Q::N0 && Q::K0 <= 0 && Q::XU >= 0 && (J == Q::XU || J == Q::XU + 1)) sfill_ext<Q::K0>(ring.b[slot], unit_addr<Q::K0, Q::E>(q.s0, J), lane);
    else if constexpr (J < Q::N0) sfill_k<Q::K0>(ring.b[slot], unit_addr<Q::K0, Q::E>(q.s0, J), lane);
    else if constexpr (J < Q::N0 + Q::N1) sfill_k<Q::K1>(ring.b[slot], unit_addr<Q::K1, Q::E>(q.s1, J - Q::N0), lane);
    else if constexpr (slot < Q::M0) sfill_k<Q::KN0>(ring.b[slot], unit_addr<Q::KN0, Q::E>(q.n0, slot), lane);
    else sfill_k<Q::KN1>(ring.b[slot], unit_addr<Q::KN1, Q::E>(q.n1, slot - Q::M0), lane);
    DFF_PIN_VMEM();
}
// first DR units of a block (step start), all from one stream of kind KIND
template <int DR, int KIND, int E>
DEVI void sring_prefetch(SRing<DR>& r, const SStream& n0, int lane) {
#pragma unroll
    for (int j = 0; j < DR; ++j) sfill_k<KIND>(r.b[j], unit_addr<KIND, E>(n0, j), lane);
}
// wide GEMM (K = H = 32 KB32) of one wave on split operands: N output tiles, KB32 units each, the first one unit I0 of
// the block.  aux[t % 2][..]: epilogue operands of the tiles in flight (the caller preloads tiles 0 and 1,
// pre(t + 2, aux[t % 2]) requests the next ones right after tile t's were copied out).
template <int I0, int T, int N, int KB32, int NAUX, int DR, class Q, class Pre, class Epi>
DEVI void swide_from(SRing<DR>& ring, float (&aux)[2][NAUX], const u32x4 (&ah)[KB32], const u32x4 (&am)[KB32],
                     const u32x4 (&al)[KB32], const Q& q, int lane, const Pre& pre, const Epi& epi) {
    if constexpr (T < N) {
        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};   // small terms / big terms
#pragma unroll
        for (int kb = 0; kb < KB32; ++kb) {
            u32x4 bh, bm, bl;
            unit_pieces<Q::kind(I0 + T * KB32)>(ring.b[(I0 + T * KB32 + kb) % DR], bh, bm, bl);
            // a slot whose weights were split into temporaries is refilled before its MFMAs are issued
            if constexpr (Q::kind(I0 + T * KB32) > 0) {
                if (kb == 0) seq_refill<DR, I0 + T * KB32 + 0>(ring, q, lane);
                if (kb == 1) seq_refill<DR, I0 + T * KB32 + (KB32 > 1 ? 1 : 0)>(ring, q, lane);
            }
            if constexpr (Q::kind(I0 + T * KB32) < 0) {   // fp16: am / bm hold the l' pieces; cs collects the 2^11-scaled cross terms
                cs = mfma_f16(am[kb], bh, cs);
                cs = mfma_f16(ah[kb], bm, cs);
                cb = mfma_f16(ah[kb], bh, cb);
            } else {
            cs = mfma_bf16(al[kb], bh, cs);
            cb = mfma_bf16(am[kb], bh, cb);
            cs = mfma_bf16(ah[kb], bl, cs);
            cb = mfma_bf16(ah[kb], bm, cb);
            cs = mfma_bf16(am[kb], bm, cs);
            cb = mfma_bf16(ah[kb], bh, cb);
            }
        }
        if constexpr (Q::kind(I0 + T * KB32) <= 0) {
            if constexpr (KB32 >= 1) seq_refill<DR, I0 + T * KB32 + 0>(ring, q, lane);
            if constexpr (KB32 >= 2) seq_refill<DR, I0 + T * KB32 + 1>(ring, q, lane);
        }
        float auxc[NAUX];
#pragma unroll
        for (int i = 0; i < NAUX; ++i) auxc[i] = aux[T % 2][i];
        if constexpr (T + 2 < N) pre(T + 2, aux[T % 2]);
        if constexpr (Q::kind(I0 + T * KB32) < 0) epi(T, cb + cs * DFF_F16_LINV, auxc);
        else epi(T, cb + cs, auxc);
        swide_from<I0, T + 1, N, KB32, NAUX>(ring, aux, ah, am, al, q, lane, pre, epi);
    }
}
template <int I0, int N, int KB32, int NAUX, int DR, class Q, class Pre, class Epi>
DEVI void swide_run(SRing<DR>& ring, float (&aux)[2][NAUX], const u32x4 (&ah)[KB32], const u32x4 (&am)[KB32],
                    const u32x4 (&al)[KB32], const Q& q, int lane, const Pre pre, const Epi epi) {
    static_assert(KB32 <= 2, "refill list");
    swide_from<I0, 0, N, KB32, NAUX>(ring, aux, ah, am, al, q, lane, pre, epi);
}
// tall GEMM (Nout = H = 16 E) of one wave on split operands: acc[nt] += A(:, 32-column block kb) . W(unit (kb, nt)),
// kb < NKB, unit (kb, nt) = unit I0 + kb E + nt of the block; fa(kb) = this lane's four floats at columns 4 quad of block
// kb (the second four are 16 floats on).  The next block's A is read from LDS before the MFMAs of this one are issued.
// EXT: one fp32 k-step on top for a head's extension columns (of which only 0..3 carry data): A element *ext_a, weights
// ext_w[nt * ext_ts] (the s = 0 slots of the fp32 image's extension block, dff_host.hip pack_b), requested first, used last.
template <int I0, int KB, int NKB, int E, int DR, class Q, class FA>
DEVI void stall_from(SRing<DR>& ring, f32x4 (&acc)[E], f32x4 (&acc2)[E], f32x4& x0, f32x4& x1, const FA& fa, const Q& q, int lane, float sa) {
    if constexpr (KB < NKB) {
        constexpr int KD = Q::kind(I0 + KB * E);
        u32x4 ah, am, al;
        if constexpr (KD < 0) { split8h(x0, x1, ah, am); al = am; }   // fp16: (h, l')
        else split8(x0, x1, ah, am, al);
        if constexpr (KB + 1 < NKB) {
            const lfloat* pn = fa(KB + 1);
            x0 = *(const lf32x4*)pn; x1 = *(const lf32x4*)(pn + 16);
            if constexpr (KD < 0) { x0 *= sa; x1 *= sa; }
        }
        if constexpr (KD < 0) {
            // acc: h.h ; acc2: the two cross terms, 2^11 too large (stall_run folds them in at the end)
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc2[nt] = mfma_f16(am, ring.b[(I0 + KB * E + nt) % DR][0], acc2[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc2[nt] = mfma_f16(ah, ring.b[(I0 + KB * E + nt) % DR][1], acc2[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_f16(ah, ring.b[(I0 + KB * E + nt) % DR][0], acc[nt]);
            static_assert(E == 4, "refill list");
            seq_refill<DR, I0 + KB * E + 0>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 1>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 2>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 3>(ring, q, lane);
        } else if constexpr (KD == 0) {
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(al, ring.b[(I0 + KB * E + nt) % DR][0], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(ah, ring.b[(I0 + KB * E + nt) % DR][2], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(am, ring.b[(I0 + KB * E + nt) % DR][1], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(am, ring.b[(I0 + KB * E + nt) % DR][0], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(ah, ring.b[(I0 + KB * E + nt) % DR][1], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_bf16(ah, ring.b[(I0 + KB * E + nt) % DR][0], acc[nt]);
            static_assert(E == 4, "refill list");
            seq_refill<DR, I0 + KB * E + 0>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 1>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 2>(ring, q, lane);
            seq_refill<DR, I0 + KB * E + 3>(ring, q, lane);
        } else {
            // fp32 units: split, refill the slot at once, then the unit's six products on two chains
            auto one = [&](auto nti) {
                constexpr int nt = decltype(nti)::value;
                u32x4 bh, bm, bl;
                unit_pieces<Q::kind(I0 + KB * E)>(ring.b[(I0 + KB * E + nt) % DR], bh, bm, bl);
                seq_refill<DR, I0 + KB * E + nt>(ring, q, lane);
                f32x4 cs = {0.f, 0.f, 0.f, 0.f};
                cs = mfma_bf16(al, bh, cs);
                acc[nt] = mfma_bf16(am, bh, acc[nt]);
                cs = mfma_bf16(ah, bl, cs);
                acc[nt] = mfma_bf16(ah, bm, acc[nt]);
                cs = mfma_bf16(am, bm, cs);
                acc[nt] = mfma_bf16(ah, bh, acc[nt]);
                acc[nt] += cs;
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        }
        stall_from<I0, KB + 1, NKB, E>(ring, acc, acc2, x0, x1, fa, q, lane, sa);
    }
}
// rsc (fp16 engine, backward): per-row power-of-two scales of this GEMM's A operand, rsc[row] = s, rsc[16 + row] = 1 / s (LDS,
// written by the row stage that scaled the chain's input): the A fragments (and the extension element) are multiplied by
// s[row] before they are split, `acc` -- which may already hold true-unit terms -- enters and leaves in true units.
template <int I0, int NKB, int E, bool EXT, int DR, class Q, class FA>
DEVI void stall_run(SRing<DR>& ring, f32x4 (&acc)[E], const FA fa, const Q& q, int lane,
                    const lfloat* ext_a = nullptr, const gfloat* ext_w = nullptr, int ext_ts = 0, const lfloat* rsc = nullptr,
                    bool common = false) {
    constexpr bool F16 = Q::kind(I0) < 0;
    float bx[E];
    if constexpr (EXT) {
#pragma unroll
        for (int nt = 0; nt < E; ++nt) bx[nt] = ext_w[(size_t)nt * ext_ts];
    }
    float sa = 1.0f;
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, si4 = {1.f, 1.f, 1.f, 1.f};
    if (F16 && rsc) {
        if (common) {
            // rows that mix the rows of the chain's input (dK, dV of the unfolded variants: sums over i) cannot carry per-row
            // scales: ONE power of two for the tile -- the smallest row scale (= 1 / the largest inverse), times 2^-8 of headroom
            float mx = row16_max(rsc[16 + (lane & 15)]);
            int ex = (int)((__float_as_uint(mx) >> 23) & 255u);
            ex = ex < 20 ? 20 : (ex > 230 ? 230 : ex);
            sa = __uint_as_float((unsigned)(246 - ex) << 23);
            const float si = __uint_as_float((unsigned)(ex + 8) << 23);
            sc4 = (f32x4){sa, sa, sa, sa}; si4 = (f32x4){si, si, si, si};
        } else {
            sa = rsc[lane & 15];
            const int q4 = (lane >> 4) * 4;
            sc4 = *(const lf32x4*)(rsc + q4); si4 = *(const lf32x4*)(rsc + 16 + q4);
        }
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] *= sc4;
    }
    f32x4 acc2[E];
#pragma unroll
    for (int nt = 0; nt < E; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const lfloat* p0 = fa(0);
    f32x4 x0 = *(const lf32x4*)p0, x1 = *(const lf32x4*)(p0 + 16);
    if constexpr (F16) { x0 *= sa; x1 *= sa; }
    stall_from<I0, 0, NKB, E>(ring, acc, acc2, x0, x1, fa, q, lane, sa);
    if constexpr (F16) {
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] += acc2[nt] * DFF_F16_LINV;
    }
    if constexpr (EXT) {
        const float ax = *ext_a * sa;
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx[nt], acc[nt], 0, 0, 0);
    }
    if (F16 && rsc) {
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] *= si4;
    }
}

// ---- register-chained forms (fp16 engine, FOLD kernel: round 6).  Swapping the two operands of an MFMA transposes its output tile:
// mfma(W, a) leaves lane (row = lane & 15, quad) with output columns 16 t + 4 quad .. + 3 of THAT row -- and two such tiles (2 kb,
// 2 kb + 1) are exactly the eight elements of the row's A fragment for 32-column block kb in the k-permutation every image of this
// engine uses (element j <-> column 32 kb + 16 (j >> 2) + 4 quad + (j & 3)).  So a GEMM whose output is the next product's operand
// hands it over in registers: no LDS store / barrier-free LDS load / wait between the two.
// swideT: the wide GEMM (K = H = 32 KB32) in that orientation; out[t] = tile t + bias (bp[16 t + 4 quad ..], 16-byte aligned).
template <int I0, int T, int N, int KB32, bool BIAS = true, int DR, class Q>
DEVI void swideT_from(SRing<DR>& ring, f32x4 (&out)[N], f32x4 (&bq)[2], const gfloat* bp, const u32x4 (&ah)[KB32], const u32x4 (&al)[KB32],
                      const Q& q, int lane) {
    if constexpr (T < N) {
        static_assert(Q::kind(I0 + T * KB32) < 0 && KB32 == 2, "fp16 units, H = 64");
        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB32; ++kb) {
            const u32x4 (&u)[3] = ring.b[(I0 + T * KB32 + kb) % DR];
            cs = mfma_f16(u[0], al[kb], cs);
            cs = mfma_f16(u[1], ah[kb], cs);
            cb = mfma_f16(u[0], ah[kb], cb);
        }
        seq_refill<DR, I0 + T * KB32 + 0>(ring, q, lane);
        seq_refill<DR, I0 + T * KB32 + 1>(ring, q, lane);
        if constexpr (BIAS) {
            const f32x4 b = bq[T % 2];
            if constexpr (T + 2 < N) bq[T % 2] = *(const gf32x4*)(bp + 16 * (T + 2));
            out[T] = cb + cs * DFF_F16_LINV + b;
        } else out[T] = cb + cs * DFF_F16_LINV;
        swideT_from<I0, T + 1, N, KB32, BIAS>(ring, out, bq, bp, ah, al, q, lane);
    }
}
// stallR: the tall GEMM (Nout = H) whose A operand arrives as such register tiles: xr[2 kb], xr[2 kb + 1] = block kb.  EXT: one fp32
// k-step on top (element *ext_a, weights ext_w[nt ext_ts]), requested first, used last -- as stall_run.
template <int I0, int KB, int NKB, int E, int DR, class Q>
DEVI void stallR_from(SRing<DR>& ring, f32x4 (&acc)[E], f32x4 (&acc2)[E], const f32x4 (&xr)[2 * NKB], const Q& q, int lane, float sa) {
    if constexpr (KB < NKB) {
        static_assert(Q::kind(I0 + KB * E) < 0 && E == 4, "fp16 units, H = 64");
        u32x4 ah, al;
        split8h(xr[2 * KB] * sa, xr[2 * KB + 1] * sa, ah, al);
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc2[nt] = mfma_f16(al, ring.b[(I0 + KB * E + nt) % DR][0], acc2[nt]);
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc2[nt] = mfma_f16(ah, ring.b[(I0 + KB * E + nt) % DR][1], acc2[nt]);
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] = mfma_f16(ah, ring.b[(I0 + KB * E + nt) % DR][0], acc[nt]);
        seq_refill<DR, I0 + KB * E + 0>(ring, q, lane);
        seq_refill<DR, I0 + KB * E + 1>(ring, q, lane);
        seq_refill<DR, I0 + KB * E + 2>(ring, q, lane);
        seq_refill<DR, I0 + KB * E + 3>(ring, q, lane);
        stallR_from<I0, KB + 1, NKB, E>(ring, acc, acc2, xr, q, lane, sa);
    }
}
// rsc (backward): the A rows -- row lane & 15 in the register tiles -- are multiplied by rsc[row] (a power of two) before they are
// split; `acc`, which may hold true-unit terms already, enters and leaves in true units (rows 4 quad + r of the C layout: as stall_run)
template <int I0, int NKB, int E, bool EXT, int DR, class Q>
DEVI void stallR_run(SRing<DR>& ring, f32x4 (&acc)[E], const f32x4 (&xr)[2 * NKB], const Q& q, int lane,
                     const lfloat* ext_a = nullptr, const gfloat* ext_w = nullptr, int ext_ts = 0, const lfloat* rsc = nullptr,
                     bool a_scaled = false /* the register tiles are in scaled units already (the extension element is not) */) {
    float bx[E];
    if constexpr (EXT) {
#pragma unroll
        for (int nt = 0; nt < E; ++nt) bx[nt] = ext_w[(size_t)nt * ext_ts];
    }
    float sa = 1.0f;
    f32x4 si4 = {1.f, 1.f, 1.f, 1.f};
    if (rsc) {
        sa = rsc[lane & 15];
        const int q4 = (lane >> 4) * 4;
        const f32x4 sc4 = *(const lf32x4*)(rsc + q4);
        si4 = *(const lf32x4*)(rsc + 16 + q4);
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] *= sc4;
    }
    f32x4 acc2[E];
#pragma unroll
    for (int nt = 0; nt < E; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stallR_from<I0, 0, NKB, E>(ring, acc, acc2, xr, q, lane, a_scaled ? 1.0f : sa);
#pragma unroll
    for (int nt = 0; nt < E; ++nt) acc[nt] += acc2[nt] * DFF_F16_LINV;
    if constexpr (EXT) {
        const float ax = *ext_a * sa;
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx[nt], acc[nt], 0, 0, 0);
    }
    if (rsc) {
#pragma unroll
        for (int nt = 0; nt < E; ++nt) acc[nt] *= si4;
    }
}

// C layout: acc[r] <-> (row 4*(lane>>4)+r, col lane&15); unconditional (pad rows hold finite junk)
// rmax: last row of dst (pad lanes beyond it store to that dummy row)
DEVI void c_store_all(lfloat* dst, int ld, int col0, const f32x4& acc, int lane, int rmax = 15) {
    const int quad = lane >> 4, col = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[min(quad * 4 + r, rmax) * ld + col0 + col] = acc[r];
}

// the same with the lane's four row offsets (floats, pad rows already collapsed) supplied
DEVI void c_store_offs(lfloat* dst, const int (&ro4)[4], int col0, const f32x4& acc, int lane) {
    const int col = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[ro4[r] + col0 + col] = acc[r];
}

template <int KB>
DEVI void load_afrag(f32x4 (&a)[KB], const lfloat* A, int lda, int lane) {
    const lfloat* ap = A + (lane & 15) * lda + 4 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) a[kb] = *(const lf32x4*)(ap + 16 * kb);
}

// C[i][j] = sum_k A[i][k] B[j][k], K = 80 (5 k-blocks), both operands rows of a head buffer
// The 64 regular columns (two accumulator chains) ...
template <int XLD = DFF_XLD>
DEVI f32x4 wv_dot_base(const lfloat* A, const lfloat* B, int lane) {
    const lfloat* ap = A + (lane & 15) * XLD + 4 * (lane >> 4);
    const lfloat* bp = B + (lane & 15) * XLD + 4 * (lane >> 4);
    f32x4 av[4], bv[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { av[kb] = *(const lf32x4*)(ap + 16 * kb); bv[kb] = *(const lf32x4*)(bp + 16 * kb); }
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
    return acc + acc2;
}
// ... and the extension block on top: only its first 4 columns are ever non-zero (u | s, x | |x|^2, r | g_D): one k-step, exact
template <int XLD = DFF_XLD>
DEVI f32x4 wv_dot_ext(const lfloat* A, const lfloat* B, int lane, const f32x4 base) {
    const float ax = A[(lane & 15) * XLD + 64 + (lane >> 4)], bx = B[(lane & 15) * XLD + 64 + (lane >> 4)];
    return __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, base, 0, 0, 0);
}
template <int XLD = DFF_XLD>
DEVI f32x4 wv_dot_rows(const lfloat* A, const lfloat* B, int lane) {
    return wv_dot_ext<XLD>(A, B, lane, wv_dot_base<XLD>(A, B, lane));
}

// C[m][16nt+n] = sum_{k<16} Aop[m][k] B[k][16nt+n] for tiles nt in [NT0, NT1).
// TRANS = false: Aop[m][k] = T[m][k] (T = 16x16 tile, ld DFF_PLD);  true: Aop[m][k] = T[k][m].
// ks = k-steps that can be non-zero: columns / rows of T at or beyond the real rows are exact zeros (P, dS).
#ifndef DFF_WVMM_VOL
#define DFF_WVMM_VOL 1
#endif
template <int NT0, int NT1, bool TRANS, int XLD = DFF_XLD, class Epi>
DEVI void wv_mm(const lfloat* T, const lfloat* B, int lane, int ks, Epi epi) {
    const int kk = lane >> 4, mm = lane & 15;
    constexpr int XB = XLD, TP = DFF_PLD;
    // k-step s covers k = 4 s .. 4 s + 3 (lane: k = 4 s + kk), so trailing all-zero k-steps can be dropped
    // (volatile: the operands of k-steps 1..3 are only used inside the row-count branches below, and the compiler sinks a
    // plain LDS read into the branch that uses it -- every product then waits out the latency of its own operands)
#if DFF_WVMM_VOL
    typedef const volatile lfloat* vlp;
#else
    typedef const lfloat* vlp;
#endif
    float as[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) as[s] = TRANS ? *(vlp)(T + (4 * s + kk) * TP + mm) : *(vlp)(T + mm * DFF_PLD + 4 * s + kk);
    float bv[NT1 - NT0][4];
#pragma unroll
    for (int nt = NT0; nt < NT1; ++nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[nt - NT0][s] = *(vlp)(B + (4 * s + kk) * XB + 16 * nt + mm);
    f32x4 acc[NT1 - NT0];
#pragma unroll
    for (int nt = 0; nt < NT1 - NT0; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s == 0 || s < ks) {
#pragma unroll
            for (int nt = 0; nt < NT1 - NT0; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s], bv[nt][s], acc[nt], 0, 0, 0);
        }
#pragma unroll
    for (int nt = NT0; nt < NT1; ++nt) epi(nt, acc[nt - NT0]);
}

// all five tiles of a head product; SPLIT: as {0, 1, 2} | {3, 4} -- 28 instead of 44 operand + accumulator registers live at once
// (the hidden-96 kernels sit at the 256-VGPR limit of two waves per SIMD in exactly these phases: DFF_MM_SPLIT)
#ifndef DFF_MM_SPLIT
#define DFF_MM_SPLIT 1
#endif
template <bool SPLIT, bool TRANS, int XLD = DFF_XLD, class Epi>
DEVI void wv_mm5(const lfloat* T, const lfloat* B, int lane, int ks, Epi epi) {
    if constexpr (SPLIT) {
        wv_mm<0, 3, TRANS, XLD>(T, B, lane, ks, epi);
        wv_mm<3, 5, TRANS, XLD>(T, B, lane, ks, epi);
    } else wv_mm<0, 5, TRANS, XLD>(T, B, lane, ks, epi);
}

// q_ext / k / v / P of one (layer, head) travelling stash -> registers -> LDS (16 rows each;
// rows beyond the real ones come from the dummy stash row: finite, and P's are exact zeros)
struct HeadRegs {
    f32x4 q[5], k[4], v[4], p;
    float m;   // GEN: element (row = lane >> 2, c = lane & 3) of the head's [m1 | m2] rows
};
DEVI void head_fetch(HeadRegs& r, const gfloat* sqkv, const gfloat* sp, int RA, bool need_qk, int lane,
                     const gfloat* sm12 = nullptr) {
    if (sm12) r.m = ld_ntg(sm12 + lane);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = lane + 64 * u, row = min(it >> 4, RA), c4 = it & 15;
        r.v[u] = ld_ntg4(sqkv + row * DFF_QKVW + 144 + 4 * c4);
        if (need_qk) r.k[u] = ld_ntg4(sqkv + row * DFF_QKVW + 80 + 4 * c4);
    }
    if (need_qk) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int it = lane + 64 * u, row = min(it / 20, RA), c4 = it % 20;
            r.q[u] = ld_ntg4(sqkv + row * DFF_QKVW + 4 * c4);
        }
    }
    if (sp) r.p = ld_ntg4(sp + 4 * lane);
}
DEVI void head_commit(const HeadRegs& r, lfloat* Qx, lfloat* Kx, lfloat* Vx, lfloat* pb, bool need_qk, bool need_p, int lane,
                      int rla = 16, int xld = DFF_XLD) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = lane + 64 * u, row = it >> 4, c4 = it & 15;
        if (row < rla) {
            *(lf32x4*)(Vx + row * xld + 4 * c4) = r.v[u];
            if (need_qk) *(lf32x4*)(Kx + row * xld + 4 * c4) = r.k[u];
        }
    }
    if (need_qk) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int it = lane + 64 * u, row = it / 20, c4 = it % 20;
            if (row < rla) *(lf32x4*)(Qx + row * xld + 4 * c4) = r.q[u];
        }
    }
    if (need_p) *(lf32x4*)(pb + (lane >> 2) * DFF_PLD + 4 * (lane & 3)) = r.p;
}

// The way out: q_ext | k | v rows of one (layer, head) from the wave's head buffers to the stash as 13 16-byte stores per
// lane, once the QKV_ext GEMM is done -- instead of 52 scalar stores (with 64-bit address arithmetic each) interleaved
// with the GEMM's weight loads in its tile epilogues: loads and stores share the in-order vmcnt queue, and a timing-only
// build without these stores ran 2.2 us / step faster.  Rows beyond the allocated ones go to the dummy stash row.
template <bool QONLY = false>
DEVI void head_store(const lfloat* Qx, const lfloat* Kx, const lfloat* Vx, gfloat* sqkv, int RA, int lane, int rla, int xld = DFF_XLD) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int it = lane + 64 * u, row = it / 20, c4 = it - row * 20;
        if (row < rla) *(gf32x4*)(sqkv + min(row, RA) * DFF_QKVW + 4 * c4) = *(const lf32x4*)(Qx + row * xld + 4 * c4);
    }
    if (QONLY) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = lane + 64 * u, row = it >> 4, c4 = it & 15;
        if (row < rla) {
            *(gf32x4*)(sqkv + min(row, RA) * DFF_QKVW + 80 + 4 * c4) = *(const lf32x4*)(Kx + row * xld + 4 * c4);
            *(gf32x4*)(sqkv + min(row, RA) * DFF_QKVW + 144 + 4 * c4) = *(const lf32x4*)(Vx + row * xld + 4 * c4);
        }
    }
}

// q_ext | k | v (| P) of one (layer, head) from the stash (or the layer-0 table) straight into the wave's head buffers by
// LDS-DMA: global_load_lds_dwordx4 takes a per-lane global address and writes lane i's 16 bytes to LDS at M0 + 16 i, so
// one instruction fills 1 KiB of the contiguous [Q | K | V | P] regions and the row structure (84-float rows of which 80 /
// 64 are data, 20-float P rows) is folded into the source addresses.  No registers are held while the data travels, so
// the request can be issued a whole row stage before the attention block needs it (through registers that prefetch
// costs 56 VGPRs across the row stage, i.e. spills); the consumer waits with head_dma_wait().  Inline asm because the
// compiler would order EVERY later LDS read of the wave behind a builtin LDS-DMA (it has no alias information); its own
// s_waitcnt bookkeeping stays safe: vmcnt retires in order, so the extra loads can only make its waits longer.
DEVI void lds_dma16(unsigned lds_byte, const gfloat* src) {
    // (m0 cannot go on the clobber list: hipcc rejects it as a RESERVED register -- "inline asm clobber list contains reserved
    // registers: m0", ROCm 7.2 -- which is also why the write is safe: the compiler never keeps a value in a reserved register
    // across statements, it sets m0 right before each instruction of its own that reads it)
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(src) : "memory");
}
DEVI void head_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// table entry of 16-byte slot `sl` of the contiguous [Q | K | V | P] regions: float offset of its source relative to the
// head's stash rows (q_ext | k | v) or, with bit 31 set, to the head's P tile; 0xffffffff: no such slot
template <int RLA, bool QP = false, int XLD = DFF_XLD>
DEVI unsigned head_dma_entry(int sl, int RA) {
    constexpr int RQ = RLA * (XLD / 4);                    // 16-byte slots per Q / K / V region
    constexpr int NP = 16 * DFF_PLD / 4;                   // ... of the P tile
    constexpr int NREG = QP ? 1 : 3;                       // QP (FOLD layout): the regions are [Q | P], nothing else is loaded
    if (sl >= NREG * RQ + NP) return 0xffffffffu;
    const int reg = QP ? (sl >= RQ ? 3 : 0) : (sl >= RQ) + (sl >= 2 * RQ) + (sl >= 3 * RQ);
    const int r = sl - (QP ? (reg ? RQ : 0) : reg * RQ);
    if (reg < 3) {
        const int row = r / (XLD / 4), c4 = r - row * (XLD / 4);
        // q_ext: 20 slots of data + 1 pad; k / v: 16 + 4 (extension columns, rewritten by write_xext) + 1 pad
        const int lim = reg == 0 ? 19 : 15, off = reg == 0 ? 0 : reg == 1 ? 80 : 144;
        return (unsigned)(min(row, RA) * DFF_QKVW + off + 4 * min(c4, lim));
    }
    const int row = r / (DFF_PLD / 4), c4 = r - row * (DFF_PLD / 4);
    return 0x80000000u | (unsigned)(row * 16 + 4 * min(c4, 3));
}
// QROWS > 0: only the q_ext rows are wanted (no P tile: the caller keeps it in LDS) and they are the first QROWS 16-byte slots of the
// table -- ceil(QROWS / 64) instructions, no source select per lane (the general form below issued ~90 instructions for the same four
// loads at the top of a row stage every wave waits for)
template <int QROWS>
DEVI void head_dma_q(const lu32* tab, const lfloat* Qx, const gfloat* sqkv, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)Qx);
    constexpr int NI = (QROWS + 63) / 64;
    unsigned e[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) e[k] = tab[64 * k + lane];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        if (64 * k + 63 < QROWS || lane < QROWS - 64 * k) lds_dma16(base + 1024u * k, sqkv + e[k]);
    }
}
template <int NI>
DEVI void head_dma(const lu32* tab, const lfloat* Qx /* wave-uniform; the regions the table describes start here */, const gfloat* sqkv, const gfloat* sp, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)Qx);
    unsigned e[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) e[k] = tab[64 * k + lane];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const bool isp = (e[k] >> 31) != 0;
        const gfloat* src = (isp ? sp : sqkv) + (e[k] & 0x7fffffffu);
        if (e[k] != 0xffffffffu && (!isp || sp)) lds_dma16(base + 1024u * k, src);
    }
}

// ---------------------------------------------------------------- the kernel
// FOLD (hidden == head dimension, dff_host.hip folds W_k into W_q and W_v into W_o): keys and values ARE the LayerNorm rows,
// so the QKV_ext GEMM has 5 tiles per head instead of 13 (q' | u), K_ext = V_ext is ONE shared fp32 copy of the LayerNorm
// output (+ x), written by the row stages next to its bf16 pieces, dK and dV go straight into the wave's partial of
// d(LayerNorm output) (identity back-projection) and the QKV_ext^T GEMM keeps only its dQ blocks; only q' is stashed.
// MODE (DFF_MODE_SCORE / LANGEVIN / DDPM) is a template argument: one kernel per sampler mode.  With the three update stages
// and their mode tests inside one step loop the register allocator and the scheduler paid for all of them everywhere
// (the Langevin step ran 1.5 us slower next to the reverse-DDPM update's code than without it); each mode is its own
// translation unit (build.sh compiles this file three times, -DDFF_SMALL_MODE=0|1|2).
// PAIR (round 6; FOLD kernel on the fp16 engine): TWO workgroups per protein for per-GPU batches that leave half the CUs idle
// (the reference's published protocol is --parallel_sim 100, evaluate/sampling_commands.md:13; BASELINE config 2 over 8 GPUs is 32 per
// GPU).  Blocks b and b + 8 (one XCD under the observed placement) own heads / FFN slices 0-3 and 4-7, on four waves -- one per SIMD --
// each; the row stages run redundantly in both; the H-wide partial sums a row stage collects (four per layer) and the final dE/dx are
// exchanged through global memory with the protocol of dff_fused_kernel<..., PAIR> (csrc/dff_kernels.hip): 16-byte stores, flag,
// poll, 16-byte loads, a + b = b + a so both blocks stay bit-identical.  tools_ubench/pair_exchange.hip: 0.25 - 0.45 us per exchange
// on one XCD, 1.0 - 1.2 us across XCDs.
template <int H, int NW, bool GEN, bool SPW, bool FOLD, int MODE, bool PAIR = false>
__global__ __launch_bounds__((PAIR ? 4 : NW) * 64) void dff_small_kernel(const DffModelDev m, const DffRunArgs a) {
    static_assert(!FOLD || (SPW && !GEN && H == DFF_DH), "FOLD: the split, shipped-branch, hidden == 64 variant");
    static_assert(!PAIR || (FOLD && NW == 8 && DFF_F16_ON(FOLD)), "PAIR: the FOLD kernel on the fp16 engine");
    constexpr int NWR = PAIR ? 4 : NW;       // waves of this workgroup
    using LL = SmallLds<H, NW, FOLD, NWR>;
    constexpr int LH = LL::LH, F = 4 * H, E = H / 16;
    constexpr int NTHR = NWR * 64;           // threads
    constexpr int HPW = DFF_HEADS / NW;      // heads per wave (2 or 1)
    constexpr int DR = NW == 4 ? 4 : 2;      // weight-ring depth (2 waves/SIMD need less run-ahead)
    constexpr int RLA = LL::RLA;             // rows allocated per head buffer
    constexpr int XLD = LL::XLD;             // row stride of the head buffers
    constexpr int RS = RLA * XLD;            // floats between the Q / K / V / G buffers of a wave
    constexpr int FS = F / NW, NTS = FS / 16, LF = FS + 4;   // FFN hidden slice of a wave
    static_assert(HPW == 1 || HPW == 2, "4 or 8 waves");
    constexpr int KB32 = H / 32, SDR = DFF_SDR;   // SPW: 32-row k-blocks of a K = H GEMM, split-ring depth in units
    static_assert(!SPW || (NW == 8 && H == 64 && FS == 32), "split-bf16 variant: one head per wave; the unit counts below are H = 64's");
    // units per GEMM of a wave (H = 64): QKV_ext 13 tiles x 2, [W_o;W_oc] 2 k-blocks x 4, W1 / W2^T 2 x 2, W2 / W1^T 1 x 4,
    // [W_o;W_oc]^T 5 x 2, QKV_ext^T 6 x 4
    constexpr int NQT = FOLD ? 5 : 13;        // output tiles per head of the QKV_ext GEMM ([q' | u] or [q | u | k | v])
    constexpr int NKT = FOLD ? 2 : 6;         // 32-row k-blocks per head of its transpose (dQ or dQ | dK | dV)
    constexpr int U_QKV = NQT * KB32, U_WOX = 2 * E, U_W1 = NTS * KB32, U_W2 = (FS / 32) * E, U_GX = 5 * KB32, U_QKVT = NKT * E;
    constexpr int MW = U_W1 < SDR ? U_W1 : SDR;   // how many of a block's first SDR units come from its first GEMM when that is W1 / W2^T
    static_assert(!SPW || (U_W1 + U_W2 >= SDR && U_WOX >= SDR && U_GX >= SDR && U_QKV >= SDR), "every block fills the ring");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    // the wave index is wave-uniform: say so (readfirstlane), so that every per-wave pointer and
    // weight-stream base lives in SGPRs and its arithmetic runs on the scalar unit
    const int rwave = __builtin_amdgcn_readfirstlane(tid >> 6);                 // this wave's LDS region / dx partial
    // PAIR: blocks b and b + 8 form pair (b & 7) + 8 (b >> 4); hf = which half of the heads / FFN slices (tests, a.xslow == 2:
    // partners are blocks b and b + 1 -- different XCDs under round-robin placement -- to drive the agent-scope protocol)
    const bool adj = PAIR && a.xslow == 2;
    const int hf = PAIR ? (int)(adj ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1)) : 0;
    const int unit = PAIR ? (int)(adj ? (blockIdx.x >> 1) : ((blockIdx.x & 7) + 8 * (blockIdx.x >> 4))) : (int)blockIdx.x;
    const int wave = PAIR ? rwave + 4 * hf : rwave;                             // the head / FFN slice this wave owns
    const int N = m.N, G = a.G;
    const int b0 = a.b_base + unit * G;
    const int gcnt = min(G, a.B - b0);
    if (gcnt <= 0) return;
    const int rows = gcnt * N, RA = G * N;   // real rows of this workgroup / dummy stash row index
    const int ks4 = __builtin_amdgcn_readfirstlane((rows + 3) >> 2);   // k-steps of a P / dS tile that can be non-zero
    lfloat* const sm = (lfloat*)smem;
    lfloat* const xst = sm + LL::xst; lfloat* const xs = sm + LL::xs; lfloat* const dxs = sm + LL::dxs;
    lfloat* const vst = sm + LL::vst; lfloat* const cm = sm + LL::cm; lfloat* const tn = sm + LL::tn;
    lfloat* const xcb = cm;        // Langevin: the integrator's centred x_old of this step (written by the centring, read by the update)
    lfloat* const xib = cm + 64;   // this step's standard normals, one per (bead, component) thread
    lfloat* const abuf = sm + LL::abuf; lfloat* const resbuf = sm + LL::resbuf;
    lu16* const asp16 = (lu16*)(sm + LL::asp);
    lu32* const dmatab = (lu32*)(sm + LL::dmatab);
    // (the row stages' stores of the K = H GEMM input -- fp32 `abuf`, or its pieces at the position the consuming waves' A fragments
    // expect: element j of lane (row, kg) of k-block kb is column 32 kb + 16 (j >> 2) + 4 kg + (j & 3) -- are a_put / n_put below)
    // ... and a wave's A fragments of it (SPW): pieces of k-block kb for row lane & 15, k-group lane >> 4
    auto a_load = [=](u32x4 (&ah)[H / 32], u32x4 (&am)[H / 32], u32x4 (&al)[H / 32], int lane) {
        const lu32* const q = (const lu32*)asp16 + ((lane >> 4) * 16 + (lane & 15)) * 4;
        constexpr int PS = (H / 32) * 256;
#pragma unroll
        for (int kb = 0; kb < H / 32; ++kb) {
            ah[kb] = *(const lu32x4*)(q + kb * 256);
            am[kb] = *(const lu32x4*)(q + PS + kb * 256);
            if constexpr (SPW && DFF_F16_ON(FOLD)) al[kb] = am[kb];   // (two pieces: h, l')
            else al[kb] = *(const lu32x4*)(q + 2 * PS + kb * 256);
        }
    };
    lfloat* const dxw = sm + LL::dxw + rwave * 128;
    lfloat* const wr = sm + LL::wreg + rwave * LL::WREG;
    // FOLD: K_ext = V_ext = ONE shared fp32 copy of the LayerNorm rows (+ x in the extension columns), region `nx`
    lfloat* const Nx = sm + LL::nx;
    lfloat* const Qx = wr; lfloat* const Kx = FOLD ? Nx : wr + RS; lfloat* const Vx = FOLD ? Nx : wr + 2 * RS;
    constexpr bool NSP = FOLD && SPW && DFF_F16_ON(FOLD);   // the attention input's fp16 pieces live in their own region (SmallLds::nsp)
    lu16* const nsp16 = (lu16*)(sm + LL::nsp);
    lu16* const nst16 = (lu16*)(sm + LL::nst);
    // lane (column m = lane & 15 of tile nt, kg = lane >> 4): rows 4 kg .. + 3 of that column as the A operand of a 16x16x16 fp16 MFMA
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto nt_load = [=](int nt, int lane, f16x4& vh, f16x4& vl) {
        const lu32x2* const q = (const lu32x2*)(nst16 + ((nt * 4 + (lane >> 4)) * 16 + (lane & 15)) * 4);
        vh = __builtin_bit_cast(f16x4, *q);
        vl = __builtin_bit_cast(f16x4, *(q + (H / 16) * 4 * 16 * 4 / 4));
    };
    // ---- this thread's place in the row stages, once per kernel (round 6).  The row stages are issue-bound on the SIMD that carries
    // two of the five row waves: every instruction they do not issue is ~0.1 % of a step, and a third of them was address
    // arithmetic re-derived per stage from the thread index (row * LH + column, the piece positions of a_store / n_store).  Four
    // registers hold them for the kernel's lifetime (opaque, so that they are neither re-derived nor folded into per-stage copies):
    //   rs_o  resbuf / partial-sum tiles: row * LH + sub          rs_n  `nx`: row * XLD + sub
    //   rs_a  halfword index of (row, column sub) in asp / nsp    rs_t  ... in the transposed pieces nst
    // column sub + LPR i is a compile-time offset away from each (RS_AOFF, RS_TOFF below).
    constexpr int RSL = NWR == 8 ? 32 : 16;   // = LPR
    int rs_o, rs_n, rs_a, rs_t;
    {
        const int rr = (int)((unsigned)tid / (unsigned)RSL), ss = (int)((unsigned)tid % (unsigned)RSL);
        rs_o = rr * LH + ss;
        rs_n = rr * XLD + ss;
        const int kg = (ss & 15) >> 2, j = ((ss >> 4) << 2) | (ss & 3);   // (ss < 32: k-block 0)
        rs_a = ((((0 * 4 + kg) * 16 + rr) * 4 + (j >> 1)) * 2 + (j & 1));
        rs_t = ((((ss >> 4) * 4 + (rr >> 2)) * 16 + (ss & 15)) * 4 + (rr & 3));
        asm volatile("" : "+v"(rs_o), "+v"(rs_n), "+v"(rs_a), "+v"(rs_t));
    }
#define RS_AOFF(i) (RSL == 32 ? (i) * 512 : ((i) >> 1) * 512 + ((i) & 1) * 4)   /* halfwords: k-block stride 512; 16-lane rows: column + 16 = element j + 4 */
#define RS_TOFF(i) (RSL == 32 ? (i) * 512 : (i) * 256)                          /* halfwords: 16-column tile stride 256 */
    // row-stage stores of this thread's element i (column sub + LPR i): the K = H GEMM input (a_put) / the attention input (n_put)
    auto a_put = [&](int i /* unrolled loops: a constant */, float v) {
        if constexpr (SPW) {
            lu16* const q = asp16 + rs_a + RS_AOFF(i);
            constexpr int PS = (H / 32) * 256 * 2;
            if constexpr (DFF_F16_ON(FOLD)) {
                unsigned short hh, ll;
                split1h(v, hh, ll);
                q[0] = hh; q[PS] = ll;
            } else {
                const unsigned b = __float_as_uint(v);
                const float r = v - __uint_as_float(b & 0xffff0000u);
                const unsigned c = __float_as_uint(r);
                const float s2 = r - __uint_as_float(c & 0xffff0000u);
                q[0] = (unsigned short)(b >> 16); q[PS] = (unsigned short)(c >> 16); q[2 * PS] = (unsigned short)(__float_as_uint(s2) >> 16);
            }
        } else abuf[rs_o + RSL * i] = v;
    };
    auto n_put = [&](int i, float v) {
        if constexpr (FOLD) Nx[rs_n + RSL * i] = v;
        if constexpr (NSP) {
            unsigned short hh, ll;
            split1h(v, hh, ll);
            lu16* const q = nsp16 + rs_a + RS_AOFF(i);
            q[0] = hh; q[(H / 32) * 256 * 2] = ll;
            lu16* const qt = nst16 + rs_t + RS_TOFF(i);
            qt[0] = hh; qt[(H / 16) * 4 * 16 * 4] = ll;
        }
    };
    // (the A fragments of them: as a_load)
    auto an_load = [=](u32x4 (&ah)[H / 32], u32x4 (&al)[H / 32], int lane) {
        const lu32* const q = (const lu32*)nsp16 + ((lane >> 4) * 16 + (lane & 15)) * 4;
#pragma unroll
        for (int kb = 0; kb < H / 32; ++kb) {
            ah[kb] = *(const lu32x4*)(q + kb * 256);
            al[kb] = *(const lu32x4*)(q + (H / 32) * 256 + kb * 256);
        }
    };
    // RELAY (8 waves, H = 64): region order [Q | K | V | P | G | dS] ([Q | P | G | dS | ...] in the FOLD layout, SmallLds),
    // and everything that is not an attention operand -- o_ext = P V_ext, the FFN hidden slice, the wave's partial H-wide
    // outputs -- lives in G | dS.  q_ext, k, v and P of the LAST layer then survive in LDS from its forward to its backward
    // attention block: no stash reload (and no exposed HBM round trip) for one layer in three.  Otherwise
    // [Q | K | V | G | P | dS] with those tiles aliasing Q | K.
    constexpr bool RELAY = NW == 8 && H == 64;
    constexpr bool KEEP_LAST = RELAY && !GEN;   // (the GEN variants re-derive their x-dependent extension columns on reload)
    constexpr bool HDMA = RELAY && !GEN;        // stash -> head buffers by LDS-DMA, requested a row stage ahead (head_dma)
    // ... of a head's rows WITHOUT its P tile (layer 0 from the table / the stash): the FOLD layout has only q' rows to fetch
    auto dma_nop = [=](const gfloat* sqkv, int lane) {
        if constexpr (FOLD) head_dma_q<LL::RLA * (LL::XLD / 4)>(dmatab, Qx, sqkv, lane);
        else head_dma<LL::DMA_N>(dmatab, Qx, sqkv, nullptr, lane);
    };
    // FOLD has no K / V regions, and the layer BEFORE the last one keeps its q' and P in LDS too: copied to Qsave | Psave
    // after its forward attention, copied back before its backward one -- with the layer-0 table and KEEP_LAST no q' / P of
    // a 3-layer model ever goes through the stash in the sampling loops.
    constexpr bool KEEP2 = FOLD && KEEP_LAST;
    lfloat* const pb = FOLD ? wr + RS : RELAY ? wr + 3 * RS : wr + 4 * RS;
    lfloat* const Gx = RELAY ? pb + 16 * DFF_PLD : wr + 3 * RS;
    lfloat* const dsb = RELAY ? Gx + RS : pb + 16 * DFF_PLD;
    lfloat* const Qsave = dsb + 16 * DFF_PLD;          // (FOLD layout only)
    // Saved softmax tiles: the 10 x 10 real entries of a head's P live in the spare columns of the Qsave rows -- layer 0's in
    // columns 68..77, layer (L - 2)'s in 78..87 -- between the layer's forward and its backward attention block.  With the last
    // layer's tile left in place, no q' / P of a 3-layer model goes through the stash in the sampling loops.
    constexpr int P0C = 68, P1C = 78;
    static_assert(!FOLD || (P1C + 10 <= XLD && RLA >= 10), "the saved P rows fit behind the q' columns of Qsave");
    // (ij = the lane's two entries as (row, column) nibbles: `pcij`, set up once per kernel)
    auto p_copy = [=](int c0, bool save, unsigned ij) {   // pb (zero outside the real entries: left alone) <-> Qsave columns c0..c0+9
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = (ij >> (8 * u)) & 15, j = (ij >> (8 * u + 4)) & 15;
            if (u == 0 || (ij >> 16)) {
                if (save) Qsave[i * XLD + c0 + j] = pb[i * DFF_PLD + j];
                else pb[i * DFF_PLD + j] = Qsave[i * XLD + c0 + j];
            }
        }
    };
    auto p0_copy = [=](bool save, unsigned ij) { p_copy(P0C, save, ij); };
    // (q': the 68 data columns of every row)
    auto keep2_copy = [=](const lfloat* qs, lfloat* qd, bool save, int lane, unsigned ij) {
#pragma unroll
        for (int u = 0; u < (RLA * 17 + 63) / 64; ++u) {
            const int it = lane + 64 * u, row = it / 17, o = row * XLD + 4 * (it - row * 17);
            if (it < RLA * 17) *(lf32x4*)(qd + o) = *(const lf32x4*)(qs + o);
        }
        p_copy(P1C, save, ij);
    };
    static_assert(!HDMA || XLD % 4 == 0, "16-byte slots");
    constexpr unsigned MPO = FOLD ? RS + 16 * DFF_PLD : RELAY ? 3 * RS + 16 * DFF_PLD : 0;   // offset of G | dS (the aliased tiles) inside a wave region
    static_assert(!RELAY || RS + 16 * DFF_PLD >= 16 * (H + 4), "G | dS must hold a 16 x (H + 4) partial-sum tile");
    lfloat* const Ox = RELAY ? Gx : Qx;   // o_ext
    lfloat* const hbuf = wr + MPO;        // FFN hidden slice of this wave (aliases head buffers)
    lfloat* const mypart = wr + MPO;      // this wave's partial H-wide output (ditto; summed by the row stages)
    // rows of the partial-sum tile: 16, or (FOLD) the RLA = 11 allocated ones -- pad rows collapse onto the last -- which
    // leaves room behind it for the last layer's GELU' tile
    constexpr int PRMAX = FOLD ? RLA - 1 : 15;
    // GELU'(h_pre) of this wave's FFN hidden slice stays in LDS in the sampling loops (FOLD: the host uses this variant
    // for models of <= 3 layers only): the last layer's tile behind the partial sums in G | dS (its backward follows at
    // once), layers 0 and 1 of a deeper model in GP0 / GP1
    auto gp_tile = [=](int l, int L) -> lfloat* {
        return l == L - 1 ? wr + MPO + LL::GP_LAST : wr + (3 * RS + 2 * 16 * DFF_PLD) + (unsigned)l * LL::GPT;
    };
    // sum of the NW waves' partial outputs for this lane's HC columns of a row: ALL NW x HC LDS reads are
    // issued first (one latency), then added in wave order (left to itself the compiler issues one read,
    // waits, adds, issues the next: NW/2 serial LDS latencies inside a stage every other wave waits for)
    auto psum_all = [=](float (&out)[(H / (NWR == 8 ? 32 : 16))], int o0) {
        constexpr int HC_ = H / (NWR == 8 ? 32 : 16), LPR_ = NWR == 8 ? 32 : 16;
        float pv[NWR][HC_];
#pragma unroll
        for (int w = 0; w < NWR; ++w)
#pragma unroll
            for (int i = 0; i < HC_; ++i) pv[w][i] = sm[LL::wreg + w * LL::WREG + MPO + o0 + LPR_ * i];
        __builtin_amdgcn_sched_barrier(0);
        // (one v_add_f32 per term, spelled out: left to itself the compiler packs the two columns into v_pk_add_f32 and pays a
        // v_mov per operand to line the pairs up -- 24 instructions for 14 additions, in the stages every wave waits for)
#pragma unroll
        for (int i = 0; i < HC_; ++i) {
            float t = pv[0][i];
#pragma unroll
            for (int w = 1; w < NWR; ++w) asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(pv[w][i]));
            out[i] = t;
        }
    };
    // ---- PAIR: the exchanges.  Slot layout [pair][half][parity][thread][4 floats]: a row-stage thread's four column sums travel as
    // one 16-byte store / load (the partner's threads hold the same (row, columns)); a + b = b + a keeps the two blocks bit-identical.
    // Same-XCD pairs (decided once per launch, below): plain stores acknowledged by the L2 both blocks share + L1-bypassing loads;
    // otherwise agent-scope (sc1) stores.  A bounded spin replaces a hang and sets the model's sticky word (dff_model_status).
    unsigned xseq = 0;
    bool xfast = false;
    constexpr unsigned XSLOT = 4 * 64 * 4;   // floats per slot (four waves x 64 lanes x 4)
    auto pair_flag = [&](unsigned* const flags) {   // one lane: publish exchange xseq, wait for the partner's
        if (xfast) asm volatile("global_store_dword %0, %1, off" ::"v"(flags + hf), "v"(xseq + 1) : "memory");
        else __hip_atomic_store(flags + hf, xseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned* const err = a.xflag;
        unsigned spins = 0;
        while (__hip_atomic_load(flags + (1 - hf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < xseq + 1) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                atomicOr(err, 1u);
                break;
            }
        }
    };
    auto pair_rows = [&](float (&ps)[(H / (NWR == 8 ? 32 : 16))], int rrow_, int sub_, bool ract_, int tq_) {
        if constexpr (PAIR) {
            static_assert(!PAIR || H / 16 == 4, "one 16-byte slot entry per thread");
            float* const mine = a.xchg + ((size_t)(2 * unit + hf) * 2 + (xseq & 1)) * XSLOT + 4 * tq_;
            const float* const theirs = a.xchg + ((size_t)(2 * unit + (1 - hf)) * 2 + (xseq & 1)) * XSLOT + 4 * tq_;
            if (ract_) {
                psum_all(ps, rrow_ * LH + sub_);
                const f32x4 v = {ps[0], ps[1], ps[2], ps[3]};
                if (xfast) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine), "v"(v) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (tq_ == 0) pair_flag(a.xflag + 1 + 2 * unit);
            __syncthreads();
            if (ract_) {
                f32x4 pv;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pv) : "v"(theirs) : "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) ps[i] += pv[i];
            }
            ++xseq;
        }
    };
    auto pair_dx = [&](float t, int tq_) -> float {   // wave 0's lanes (tq < 64) only: the final dE/dx partial sums
        float r = t;
        if constexpr (PAIR) {
            float* const mine = a.xchg + ((size_t)(2 * unit + hf) * 2 + (xseq & 1)) * XSLOT + tq_;
            const float* const theirs = a.xchg + ((size_t)(2 * unit + (1 - hf)) * 2 + (xseq & 1)) * XSLOT + tq_;
            if (xfast) asm volatile("global_store_dword %0, %1, off" ::"v"(mine), "v"(t) : "memory");
            else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(mine), "v"(t) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tq_ == 0) pair_flag(a.xflag + 1 + 2 * unit);
            __builtin_amdgcn_wave_barrier();
            float pv;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(pv) : "v"(theirs) : "memory");
            r = t + pv;   // (the caller counts the exchange: every thread of the block must)
        }
        return r;
    };
    const SmallStash sl = dff_small_stash(N, G, H, m.L);
    gfloat* const stash = (gfloat*)a.stash + (size_t)blockIdx.x * a.stash_stride;
    Ctx c;  // only what bead_mean() needs
    c.N = N; c.G = G; c.gcnt = gcnt; c.rows = rows;

    for (int i = tid; i < (int)LL::total; i += NTHR) smem[i] = 0.f;
    __syncthreads();
    if constexpr (LL::dmatab_size > 0) {
        for (int i = tid; i < (int)LL::dmatab_size; i += NTHR) dmatab[i] = head_dma_entry<LL::RLA, FOLD, LL::XLD>(i, G * m.N);
        __syncthreads();
    }
    if constexpr (PAIR) {
        // A launch on top of a failed one (sticky word set: an earlier PAIR launch lost a partner) leaves at once with its OUTPUTS
        // set to NaN; otherwise the two blocks tell each other which XCD they run on (as dff_fused_kernel<..., PAIR>).
        lu32* const scr = (lu32*)(sm + LL::prof);
        if (tid == 0) scr[0] = __hip_atomic_load(a.xflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned failed = scr[0];
        __syncthreads();
        if (failed) {
            const float qnan = __builtin_nanf("");
            const int nr3 = rows * 3;
            const size_t o3 = (size_t)b0 * N * 3;
            if (hf == 0) {
                if (MODE == DFF_MODE_SCORE) {
                    for (int i = tid; i < nr3; i += NTHR) a.force_out[o3 + i] = qnan;
                    if (a.energy_out) for (int i = tid; i < rows; i += NTHR) a.energy_out[(size_t)b0 * N + i] = qnan;
                } else if (MODE == DFF_MODE_LANGEVIN) {
                    const int nf = a.n_steps / (a.save_interval > 0 ? a.save_interval : 1);
                    for (int f = 0; f < nf; ++f) {
                        if (a.frames) for (int i = tid; i < nr3; i += NTHR) a.frames[(size_t)f * a.B * N * 3 + o3 + i] = qnan;
                        if (a.ke && tid < gcnt) a.ke[(size_t)f * a.B + b0 + tid] = qnan;
                    }
                } else {
                    for (int i = tid; i < nr3; i += NTHR) a.x_io[o3 + i] = qnan;
                }
            }
            return;
        }
        unsigned* const xw = a.xflag + 1 + 2 * a.xpairs + 2 * unit;
        if (tid == 0) {
            const unsigned my_xcc = (__builtin_amdgcn_s_getreg(20 | ((4 - 1) << 11)) & 15u) + 1u;
            __hip_atomic_store(xw + hf, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* const err = a.xflag;
            unsigned theirs = 0, spins = 0;
            while ((theirs = __hip_atomic_load(xw + (1 - hf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    atomicOr(err, 1u);
                    break;
                }
            }
            scr[0] = (theirs == my_xcc && a.xslow != 1) ? 1u : 0u;
        }
        __syncthreads();
        xfast = scr[0] != 0u;
        __syncthreads();
        if (tid == 0) scr[0] = 0u;   // (the stage-tick accumulators live here)
    }
    Prof pf;
    pf.on = (a.prof != nullptr) && blockIdx.x == 0 && tid == 64 * (a.prof_wave < NWR ? a.prof_wave : 0);   // one wave's view (wave 0 unless dff_debug_profile asked for another)
    pf.acc = (unsigned long long*)(smem + LL::prof);
    pf.last = __builtin_readcyclecounter();

    // per-lane constants of the MFMA C layout, re-derived inside each block (see lane_id()):
    // stash row (pad rows -> dummy row RA), dx slot, protein of column j (block-diagonal attention)
    // (packed once per kernel into four VGPRs -- lcP0/1: lro[0..3] as 16-bit fields, lcD: dxi[0..3] as bytes, lcS: srow[0..3]
    // as bytes -- and unpacked with one bit-field extract each where a block uses them: re-deriving them cost ~45 integer
    // VALU instructions at the head of every wave-private block, hoisting them raw would pin 12 registers)
    unsigned lcP0 = 0, lcP1 = 0, lcD = 0, lcS = 0, lcM0 = 0, lcM1 = 0;   // (lcM0/1: row offsets of the partial-sum tile, 16-bit fields)
    {
        const int ln_ = tid & 63, q_ = ln_ >> 4, c_ = ln_ & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row_ = q_ * 4 + r;
            const unsigned lro_ = (unsigned)(min(row_, RLA - 1) * XLD);
            const unsigned dxi_ = (row_ < rows && c_ < 3) ? (unsigned)(row_ * 4 + c_) : 64u + (unsigned)ln_;
            const unsigned srow_ = (unsigned)(row_ < rows ? row_ : RA);
            if (r < 2) lcP0 |= lro_ << (16 * r); else lcP1 |= lro_ << (16 * (r - 2));
            lcD |= dxi_ << (8 * r);
            lcS |= srow_ << (8 * r);
            const unsigned mro_ = (unsigned)(min(row_, FOLD ? RLA - 1 : 15) * (H + 4));
            if (r < 2) lcM0 |= mro_ << (16 * r); else lcM1 |= mro_ << (16 * (r - 2));
        }
    }
    static_assert(16 * XLD < 65536, "lro fits 16 bits");
#define DFF_LANE_CONSTS                                                                  \
    const int lane = lane_id(), quad = lane >> 4, col = lane & 15;                       \
    int srow[4], dxi[4], lro[4], mro[4];                                                 \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                      \
        srow[r] = (int)((lcS >> (8 * r)) & 255u);                                        \
        lro[r] = (int)(((r < 2 ? lcP0 : lcP1) >> (16 * (r & 1))) & 65535u);              \
        dxi[r] = (int)((lcD >> (8 * r)) & 255u);                                         \
        mro[r] = (int)(((r < 2 ? lcM0 : lcM1) >> (16 * (r & 1))) & 65535u);              \
    }                                                                                    \
    (void)srow; (void)dxi; (void)lro; (void)quad; (void)col; (void)mro;
#define DFF_ROW_CONSTS                                  \
    const int tq_ = tid_id();                           \
    const int rrow = (int)((unsigned)tq_ / (unsigned)LPR), sub = (int)((unsigned)tq_ % (unsigned)LPR);   /* (unsigned: one shift / mask each) */ \
    const bool ract = rrow < rows;

    // ---- load state (as dff_fused_kernel) ----
    {
        const float* xin = (MODE == DFF_MODE_SCORE) ? a.x_in : a.x_io;
        if (tid < rows * 4) {
            const int row = tid >> 2, cc = tid & 3;
            float xv = 0.f, vv = 0.f;
            if (cc < 3) {
                const size_t gi = ((size_t)b0 * N + row) * 3 + cc;
                if (MODE == DFF_MODE_DDPM && a.init_prior)
                    xv = philox_normal(a.seed, a.item_offset + b0 + row / N, 0xFFFFFFFFull, row % N, cc);
                else
                    xv = xin[gi];
                if (MODE == DFF_MODE_LANGEVIN && !a.overdamped) vv = a.v_io[gi];
            }
            xst[tid] = xv;
            vst[tid] = vv;
        }
        if (tid < gcnt) tn[tid] = (MODE == DFF_MODE_SCORE) ? a.tnorm[b0 + tid] : a.t_norm;
        __syncthreads();
        if (MODE == DFF_MODE_DDPM && a.init_prior) {
            bead_mean(c, (float*)xst, (float*)cm);
            __syncthreads();
            if (tid < rows * 4) xst[tid] -= cm[(tid >> 2) / N * 4 + (tid & 3)];
            __syncthreads();
        }
    }
    // Row stages: LPR lanes per row, HC columns per lane.  They run between barriers while every other
    // wave waits, so their critical path is pure loss: the 8-wave variant spreads a row over 32 lanes
    // (all 16 possible rows then use the 512 threads) and halves the per-lane work of the 4-wave layout.
    constexpr int LPR = NWR == 8 ? 32 : 16;
    constexpr int HC = H / LPR;
    // In-kernel noise (Philox + Box-Muller: a ~300-instruction dependent chain per lane) does not depend on the forces: when
    // the last wave has no rows in the row stages (chignolin: rows 0..9 are waves 0..4) it draws the step's normals during
    // row stage A, off everybody's critical path, and the update just reads them.
#ifndef DFF_XI_PRE
#define DFF_XI_PRE 1
#endif
#ifndef DFF_XI_STAGE
#define DFF_XI_STAGE 0
#endif
    const bool xi_pre = DFF_XI_PRE && MODE != DFF_MODE_SCORE && !a.noise && rows * LPR <= (NWR - 1) * 64;
    static_assert(H % LPR == 0, "row layout");
    // (DFF_XI_STAGE: the row stage of layer 0 in whose shadow the idle wave draws -- 0: A, 1: B, 2: C)
    // (the seed through an opaque copy: the ten round keys derived from it are otherwise computed once per kernel, spilled to VGPR
    // lanes -- twenty SGPRs the register file does not have -- and read back with a v_readlane + wait state each, per draw)
    auto seed_now = [&]() {
        unsigned lo = (unsigned)a.seed, hi = (unsigned)(a.seed >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));
        return ((uint64_t)hi << 32) | lo;
    };
    auto draw_xi = [&](int t_int_, int step_) {
        if (xi_pre && rwave == NWR - 1) {
            const int ln = lane_id();
            if (ln < rows * 4 && (ln & 3) < 3) {
                const int row = ln >> 2, g = row / N;
                xib[ln] = philox_normal(seed_now(), a.item_offset + (size_t)b0 + g,
                                        MODE == DFF_MODE_DDPM ? (uint64_t)t_int_ : a.step_offset + step_, row - g * N, ln & 3);
            }
        }
    };
    auto rsum = [](float v) {   // all-reduce over the LPR lanes of a row
        if constexpr (LPR == 32) {
            // lanes l and l^16 first (gfx950 v_permlane16_swap: [0] + [1] = v[l] + v[l^16]), then the 16-lane rows
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        return row16_sum(v);
    };
    auto rmaxf = [](float v) {   // all-reduce (max) over the LPR lanes of a row
        if constexpr (LPR == 32) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        return row16_max(v);
    };
    // fp16 engine, backward: a row stage's K = H GEMM input is a gradient of any magnitude -> each row is scaled by a power of two
    // that brings its maximum to [16, 32) before it is split (exact), the wave-private chain behind it runs in scaled units
    // (everything between two row stages is linear per row) and whoever leaves the chain multiplies by the inverse: `inv` for
    // the thread's own row (the FFN backward's partial sums come back to the same thread), rs[row] / rs[16 + row] in LDS for
    // the waves of the attention backward block.  Without the fp16 engine: plain stores, inv = 1.
    lfloat* const rsc = sm + LL::asp + 2 * (H / 32) * 256;   // (the third piece's area of `asp`: the fp16 engine has two)
    auto a_store_row = [&](int rrow, int sub, const float (&v)[HC], float& inv, bool publish) {
        float sc = 1.0f;
        inv = 1.0f;
        if constexpr (SPW && DFF_F16_ON(FOLD)) {
            // the row maximum of |v| as an UNSIGNED-INTEGER maximum of the bit patterns (non-negative floats order like their bits): one
            // v_max_u32 per reduction step where fmaxf costs three (it canonicalises both inputs); only the exponent is used
            unsigned um = 0u;
#pragma unroll
            for (int i = 0; i < HC; ++i) um = max(um, __float_as_uint(v[i]) & 0x7fffffffu);
            if constexpr (LPR == 32) {
                const auto sw = __builtin_amdgcn_permlane16_swap(um, um, false, false);
                um = max(sw[0], sw[1]);
            }
            um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0xB1, 0xF, 0xF, true));
            um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x4E, 0xF, 0xF, true));
            um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x141, 0xF, 0xF, true));
            um = max(um, (unsigned)__builtin_amdgcn_update_dpp(0, (int)um, 0x140, 0xF, 0xF, true));
            pow2_scale(__uint_as_float(um), sc, inv);
            if (publish && sub == 0) { rsc[rrow] = sc; rsc[16 + rrow] = inv; }
        }
#pragma unroll
        for (int i = 0; i < HC; ++i) a_put(i, v[i] * sc);
    };
    float invD = 1.0f;   // inverse row scale of the FFN backward chain in flight (stage D -> stage E of the same layer)
    auto ln_stats_row = [&](const float (&x)[HC], float& mean, float& rstd) {   // LayerNorm statistics of one row
        // (16 lanes x 4 columns: summed as (x0 + x2) + (x1 + x3) -- columns s, s + 32 | s + 16, s + 48 -- which is the order the
        // 32-lane layout reaches through its first lane swap: the two layouts give the same bits, so a layer-0 table built by the
        // 8-wave score kernel serves the two-workgroups variant bit for bit)
        float t = 0.f;
        if constexpr (HC == 4) t = (x[0] + x[2]) + (x[1] + x[3]);
        else {
#pragma unroll
            for (int i = 0; i < HC; ++i) t += x[i];
        }
        mean = rsum(t) * (1.0f / H);
        float q = 0.f;
        if constexpr (HC == 4) {
            const float d0 = x[0] - mean, d1 = x[1] - mean, d2 = x[2] - mean, d3 = x[3] - mean;
            q = (d0 * d0 + d2 * d2) + (d1 * d1 + d3 * d3);
        } else {
#pragma unroll
            for (int i = 0; i < HC; ++i) { const float d = x[i] - mean; q += d * d; }
        }
        const float var = rsum(q) * (1.0f / H);
        rstd = fast_rsqrt(var + 1e-5f);
    };

    Ring<E, DR> ring;
    SRing<SDR> sring;   // SPW variants use this one instead (the unused ring is never materialised)
    HeadRegs hr;
    // GELU'(h_pre) slice of the NEXT backward layer, fetched a whole layer ahead (the stash is HBM-resident:
    // its latency is several times the weight ring's run-ahead).  Only when the slice is one ring revolution.
    constexpr bool HP_EARLY = NTS == DR;
    float hpn[DR][4];
    // Row-stage operand registers: every row stage ends by issuing the (global) loads of the NEXT
    // row stage's per-column parameters and stashed activations -- the thread<->(row, column)
    // mapping is the same in all row stages -- so they are in flight during the wave-private block
    // in between and the row stages themselves never wait on global memory.
    float ro[9][HC];
    auto ro_load = [&](int k, const float* src, int sub) {
        const gfloat* g = (const gfloat*)src;
#pragma unroll
        for (int i = 0; i < HC; ++i) ro[k][i] = g[sub + LPR * i];
    };
    auto ro_load3 = [&](int k, const float* w, int sub) {   // gate weights [x | res | x-res]
        ro_load(k, w, sub); ro_load(k + 1, w + H, sub); ro_load(k + 2, w + 2 * H, sub);
    };
    auto ro_gate = [&](const float (&x)[HC], const float (&res)[HC], int k) {
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < HC; ++i) z += x[i] * ro[k][i] + res[i] * ro[k + 1][i] + (x[i] - res[i]) * ro[k + 2][i];
        return sigmoid_f(rsum(z));
    };
    // KEEPROWS (FOLD, sampling loops; <= 3 layers): what the backward row stages need of the forward row stages --
    // attn_out, ff and nodes_in, two values per thread and array -- stays in the registers of the thread that made it (the
    // thread <-> (row, column) mapping is the same in every row stage) instead of going through the stash: 18 VGPRs.  With
    // GELU' in LDS and q' / P in LDS, the only stash traffic left in the sampling loops is layer 0's P: the HBM / MALL
    // traffic of a launch drops to (almost) the trajectories themselves, and the per-XCD L2 working set (4 MB) to the
    // 3.6 MB of weights a step streams (measured bound with an L2-resident stash: -2.9 us / step).  Slot = layer.  The
    // backward stages read their slot into fresh locals (keep_get at the point of use): a copy into the ro[] registers the
    // parameter prefetches also target made the compiler guard every copy with an s_waitcnt vmcnt that drained the weight ring.
    constexpr bool KEEPROWS = FOLD && MODE != DFF_MODE_SCORE;
    float kp0[4][HC] = {}, kp1[4][HC] = {}, kp2[4][HC] = {};   // layer 0 / 1 / 2: [attn_out | ff | nodes_in | LayerNorm-1 output][HC]  (separate arrays, constant indices: registers)
    auto keep_put = [&](int k, auto ai, const float (&x)[HC]) {
        constexpr int A = decltype(ai)::value;
#pragma unroll
        for (int i = 0; i < HC; ++i) {   // selects of VALUES (a branch or a select of addresses would put the arrays in scratch)
            kp0[A][i] = k == 0 ? x[i] : kp0[A][i];
            kp1[A][i] = k == 1 ? x[i] : kp1[A][i];
            kp2[A][i] = k == 2 ? x[i] : kp2[A][i];
        }
    };
    // Reads select by LANE masks: written as `k == 0 ? .. : ..` on the (wave-uniform) layer index the compiler turns every read into
    // scalar compare-and-branch chains -- half a dozen taken branches per kept value in the stages every wave waits for; a copy of the
    // index the compiler cannot see through (once per stage: `lsel`) makes them two v_cndmask each.
    struct LSel { bool z, o; };
    auto lsel = [](int k) { int kv = k; asm volatile("" : "+v"(kv)); return LSel{kv == 0, kv == 1}; };
    auto keep_get = [&](const LSel k, auto ai, float (&x)[HC]) {
        constexpr int A = decltype(ai)::value;
#pragma unroll
        for (int i = 0; i < HC; ++i) x[i] = k.z ? kp0[A][i] : (k.o ? kp1[A][i] : kp2[A][i]);
    };
    // attn_out / nodes_in / ff of layer l for a backward row stage: this thread's kept registers, or the prefetched ro[]
    auto rows_of = [&](const LSel l, float (&ao)[HC], float (&ni)[HC], float (*fv)[HC], int ffslot) {
#pragma unroll
        for (int i = 0; i < HC; ++i) { ao[i] = ro[0][i]; ni[i] = ro[1][i]; if (fv) (*fv)[i] = ro[ffslot][i]; }
        if constexpr (KEEPROWS) {
            keep_get(l, std::integral_constant<int, 0>{}, ao);
            if (fv) keep_get(l, std::integral_constant<int, 1>{}, *fv);
            keep_get(l, std::integral_constant<int, 2>{}, ni);
        }
    };
    using KA = std::integral_constant<int, 0>; using KF = std::integral_constant<int, 1>; using KN = std::integral_constant<int, 2>;
    using KL = std::integral_constant<int, 3>;
    // ... and six numbers per row and layer (the same in all lanes of the row): the two gate values [0, 1] and -- round 6 -- mean and
    // 1 / std of the two LayerNorms [2, 3: LN1; 4, 5: LN2], so that the backward stages E and F normalise their kept rows without
    // taking the statistics again (two lane reductions each, on the stage every wave waits for; same inputs, same code: same bits)
    constexpr int NKG = 6;
    float kg0[NKG] = {}, kg1[NKG] = {}, kg2[NKG] = {};
    f32x4 s0keep = {0.f, 0.f, 0.f, 0.f};   // layer 0's q' . n logits of this wave's head (Langevin: constant over a launch)
    auto gate_put = [&](int k, int which, float g) {
#pragma unroll
        for (int q = 0; q < NKG; ++q) {
            kg0[q] = (k == 0 && which == q) ? g : kg0[q];
            kg1[q] = (k == 1 && which == q) ? g : kg1[q];
            kg2[q] = (k == 2 && which == q) ? g : kg2[q];
        }
    };
    auto gate_get = [&](const LSel k, auto wi) {
        constexpr int Q = decltype(wi)::value;
        return k.z ? kg0[Q] : (k.o ? kg1[Q] : kg2[Q]);
    };
    // KEEPROWS: row stage D of layer ld (gate-2 backward: dn -> dff as the FFN-backward GEMM's input, dn1 partial -> resbuf)
    // on the kept rows and gate values.  It needs no partial sums, so it runs INSIDE the row stage that produces dn -- stage C
    // of the last layer, stage F of layer ld + 1 -- instead of behind a barrier of its own: 3 barriers and 3 serial stages
    // less per step.  W = ro[] slot of the gate-2 weights [x | res | x - res].
    auto stage_D = [&](int ld, const float (&dn)[HC], auto wslot, int rrow, int sub) {
        constexpr int W = decltype(wslot)::value;
        float ao[HC], ni[HC], fv[HC], n1[HC];
        const LSel sd = lsel(ld);
        keep_get(sd, KA{}, ao); keep_get(sd, KN{}, ni); keep_get(sd, KF{}, fv);
        const float g1 = gate_get(sd, std::integral_constant<int, 0>{}), g2 = gate_get(sd, std::integral_constant<int, 1>{});
        float dg = 0.f;
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            n1[i] = ao[i] * g1 + ni[i] * (1.0f - g1);
            dg += dn[i] * (fv[i] - n1[i]);
        }
        dg = rsum(dg);
        const float dz = dg * g2 * (1.0f - g2);
        float dffv[HC];
#pragma unroll
        for (int i = 0; i < HC; ++i) {
            const int cl = sub + LPR * i;
            dffv[i] = dn[i] * g2 + dz * (ro[W][i] + ro[W + 2][i]);
            resbuf[rs_o + LPR * i] = dn[i] * (1.0f - g2) + dz * (ro[W + 1][i] - ro[W + 2][i]);
        }
        a_store_row(rrow, sub, dffv, invD, false);
    };
    // The empty asm READS every row-stage operand register: the compiler takes whatever s_waitcnt their loads still need
    // here.  Called at the START of each wave-private block -- the operands were requested by the row stage before it and
    // are older than everything but the block's own first weight units, which the first MFMA needs anyway: free -- it
    // makes the operands "known complete" for the NEXT row stage, which otherwise starts with an s_waitcnt vmcnt(<= 2)
    // that also waits for the next block's prefetched weight units (loads retire in order) to land: measured 0.8 us / step.
    // (The alternative -- let the row stage itself request the next block's units after reading its operands -- was built
    // and measured 1.5 us SLOWER: the request is then late by the whole tail of the block before.)
    auto ro_touch = [&]() {
        asm volatile("" ::"v"(ro[0][0]), "v"(ro[0][HC - 1]), "v"(ro[1][0]), "v"(ro[1][HC - 1]), "v"(ro[2][0]), "v"(ro[2][HC - 1]),
                     "v"(ro[3][0]), "v"(ro[3][HC - 1]), "v"(ro[4][0]), "v"(ro[4][HC - 1]), "v"(ro[5][0]), "v"(ro[5][HC - 1]),
                     "v"(ro[6][0]), "v"(ro[6][HC - 1]), "v"(ro[7][0]), "v"(ro[7][HC - 1]), "v"(ro[8][0]), "v"(ro[8][HC - 1]));
    };
    // stage B operands: bo, g1 (3), ln2 gamma, ln2 beta
    auto pre_B = [&](const DffLayerDev& w, int sub) {
        ro_load(0, w.bo, sub); ro_load3(1, w.g1, sub); ro_load(4, w.ln2_g, sub); ro_load(5, w.ln2_b, sub);
    };
    // weight streams of this wave (per layer lw): helpers
    auto s_qkv = [&](const DffLayerDev& lw, int h) { return wide_stream(lw.Wqkvx_p, E, h * 13); };
    auto s_wox = [&](const DffLayerDev& lw, int h) { return tall_stream(lw.Wox_p, DFF_HEADS * 5, h * 5); };
    auto s_w1 = [&](const DffLayerDev& lw) { return wide_stream(lw.W1_p, E, wave * NTS); };
    auto s_w2 = [&](const DffLayerDev& lw) { return tall_stream(lw.W2_p, F / 16, wave * NTS); };
    auto s_w2t = [&](const DffLayerDev& lw) { return wide_stream(lw.W2T_p, E, wave * NTS); };
    auto s_w1t = [&](const DffLayerDev& lw) { return tall_stream(lw.W1T_p, F / 16, wave * NTS); };
    auto s_woxt = [&](const DffLayerDev& lw, int h) { return wide_stream(lw.WoxT_p, E, h * 5); };
    auto s_qkvt = [&](const DffLayerDev& lw, int h) { return tall_stream(lw.WqkvxT_p, DFF_HEADS * 13, h * 13); };
    // the same streams of the split images (units, see the split engine above)
    // (every stream is a host-split image: fp32 images split in registers by the consuming wave -- 4 B per weight, +44 VALU per unit --
    // measured 90.3 - 94.1 us / step against 90.1 in round 3: the extra VALU costs what the smaller stream saves)
    constexpr bool F16E = SPW && DFF_F16_ON(FOLD);   // fp16 two-piece engine (kind -1 streams)
    constexpr int KS = F16E ? -1 : 0;        // kind of the host-split images
    constexpr int KQ = KS, KO = KS, KT = KS;
    constexpr int UST = F16E ? 128 : 192;    // 16-byte slots per unit of a host-split image
    constexpr bool EARLY = SPW && KQ == KS && KO == KS;   // a block's tail refills can fetch either first stream of a step (one unit format)
    auto ss_qkv = [&](const DffLayerDev& lw, int h) { return sstream(lw.Wqkvx_w, h * U_QKV, UST); };
    auto ss_wox = [&](const DffLayerDev& lw, int h) { return sstream(lw.Wox_t, h * 2 * E, UST); };
    auto ss_w1 = [&](const DffLayerDev& lw) { return sstream(lw.W1_w, wave * NTS * KB32, UST); };
    auto ss_w2 = [&](const DffLayerDev& lw) { return sstream(lw.W2_t, wave * (FS / 32) * E, UST); };
    auto ss_w2t = [&](const DffLayerDev& lw) { return sstream(lw.W2T_w, wave * NTS * KB32, UST); };
    auto ss_w1t = [&](const DffLayerDev& lw) { return sstream(lw.W1T_t, wave * (FS / 32) * E, UST); };
    auto ss_woxt = [&](const DffLayerDev& lw, int h) { return sstream(lw.WoxT_w, h * 5 * KB32, UST); };
    auto ss_qkvt = [&](const DffLayerDev& lw, int h) { return sstream(lw.WqkvxT_t, h * U_QKVT, UST); };
    // extension-block weights of the tall GEMMs for the fp32 k-step (s = 0 slots of the fp32 images)
    auto wox_ext = [&](const DffLayerDev& lw, int h, int lane) { return (const gfloat*)lw.Wox_p + ((size_t)(5 * h + 4) * 64 + lane) * 4; };
    auto qkvt_ext = [&](const DffLayerDev& lw, int h, int lane) { return (const gfloat*)lw.WqkvxT_p + ((size_t)(13 * h + 4) * 64 + lane) * 4; };
    // x extension of K_ext / V_ext: columns 64..79 = [x_j, 0 ...]
    auto write_xext = [&](int lane) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = lane + 64 * e, row = idx >> 4, cc = idx & 15;
            float xv = (cc < 3 && row < rows) ? xs[row * 4 + cc] : 0.f;
            if constexpr (GEN) {
                if (cc == 3 && row < rows) {
                    const float x0 = xs[row * 4], x1 = xs[row * 4 + 1], x2 = xs[row * 4 + 2];
                    xv = x0 * x0 + x1 * x1 + x2 * x2;
                }
            }
            if (row < RLA) {
                Kx[row * XLD + 64 + cc] = xv;
                if constexpr (!FOLD) Vx[row * XLD + 64 + cc] = xv;   // (FOLD: the same buffer)
            }
        }
    };
    // ---- GEN variants (other input branches, see dff_kernels.hip / oracle/kernel_model_gen.py): extension columns
    //   Q_ext [u - 2 s x_i | s]   K_ext, V_ext [x_j | |x_j|^2]   o_ext [xrel | D]   G_ext [r - 2 gD x_i | gD]
    // the GEMMs give the x-independent parts; the fix-ups run on this wave's own 16-row buffers, lane = (row, c).
    auto fix_q = [&](int lane) {
        const int row = lane >> 2, c3 = lane & 3;
        if (row < rows && c3 < 3) {
            lfloat* q = Qx + row * XLD + 64;
            q[c3] = q[c3] - 2.0f * q[3] * xs[row * 4 + c3];
        }
    };
    auto fix_g = [&](int lane, float mv) {   // mv = [m1 | m2] element of this lane; dE/dx_i += gD (2 x_i - 2 m1_i)
        const int row = lane >> 2, c3 = lane & 3;
        if (row < rows && c3 < 3) {
            lfloat* gp = Gx + row * XLD + 64;
            const float gD = gp[3], xc = xs[row * 4 + c3];
            dxw[row * 4 + c3] += gD * (2.0f * xc - 2.0f * mv);
            gp[c3] = gp[c3] - 2.0f * gD * xc;
        }
    };
    // extension tile of dQ_ext = dS K_ext: [A | B] -> du = A, ds = -2 x_i.A + B (column 3), dE/dx_i += -2 s_i A
    auto dq_ext = [&](int row, int col, float e) {
        const int rw = min(row, rows - 1);
        const float xr = col < 3 ? xs[rw * 4 + col] : 0.f;
        const float tq = col < 3 ? -2.0f * xr * e : col == 3 ? e : 0.f;
        const float dsv = quad_sum(tq);
        if (row < rows && col < 3) dxw[row * 4 + col] += -2.0f * Qx[row * XLD + 67] * e;
        return col == 3 ? dsv : e;
    };
    // extension tile of dV_ext / dK_ext: [c | w] -> dE/dx_j += c + 2 x_j w
    auto dx_ext = [&](int row, int col, float e) {
        const float w = quad_bcast3(e);
        const int rw = min(row, rows - 1);
        return e + 2.0f * (col < 3 ? xs[rw * 4 + col] : 0.f) * w;
    };

    // Centring (utils.py:65-70; Langevin centres twice: the integrator's x_old, then the network's input).  Every
    // (bead, component) thread -- all of them in wave 0 -- reads its protein's column once (one batch of LDS reads) and
    // forms the means itself, summing in bead order exactly as bead_mean() does: same bits, no serial one-thread-per-column
    // loops in front of the whole workgroup.  The integrator's centred x_old goes to `xcb` (the update reads it there), the
    // network's input to xs; xst itself is not written.  Step 0 centres at its start; every later step was centred by the
    // update stage of the step before it (same wave: the column is read right after the new x was written, no workgroup
    // barrier in between), so a step begins with its first row stage.
    // Kernel-lifetime per-lane constants of the attention blocks (one VGPR each): which of the lane's four logits of a head's
    // 16 x 16 tile are real, same-protein pairs (block-diagonal softmax), and where the lane's two entries of a saved 10 x 10
    // softmax tile live.  Re-derived per block they were ~60 integer VALU instructions per head and layer (three divisions by
    // N, two by 10) in a phase where the two waves of a SIMD run back to back.
    unsigned okmask = 0, pcij = 0;
    {
        const int ln_ = tid & 63, q_ = ln_ >> 4, c_ = ln_ & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q_ * 4 + r;
            if (c_ < rows && i < rows && i / N == c_ / N) okmask |= 1u << r;
        }
        // entries ln_ and ln_ + 64 of a 10 x 10 tile as (row, column) nibbles; bit 16: the second one exists (< 100)
        const int i0 = ln_ / 10, i1 = (ln_ + 64) / 10;
        pcij = (unsigned)i0 | ((unsigned)(ln_ - 10 * i0) << 4) | ((unsigned)(i1 & 15) << 8) | ((unsigned)((ln_ + 64 - 10 * i1) & 15) << 12) |
               (ln_ + 64 < 100 ? 1u << 16 : 0u);
    }
    // sum of the first N of 16 column values in bead order (0 + v0 + v1 + ...), by EARLY EXIT: written as `i < N ? v[i] : 0` the
    // sixteen conditions are sixteen lane masks (32 SGPRs, with the sixteen row clamps min(i, N - 1) of the loads in front of
    // them 48) that the compiler computes once per kernel, spills to VGPR lanes and reads back one v_readlane at a time
    // in the update stage of every step -- where wave 0 works alone in front of the whole workgroup.  The loads are not
    // clamped either: rows past the protein's are read (inside the LDS allocation) and never added.
    auto colsum16 = [&](const float (&v)[16], float off) {
        int n_ = N;
        asm volatile("" : "+s"(n_));
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= n_) break;
            asm volatile("" : "+v"(sacc));   // (keeps the exit a scalar branch)
            sacc += v[i] - off;
        }
        return sacc;
    };
    auto centre = [&]() {
        const int tq = tid_id();
        if (tq < rows * 4) {
            const int cc = tq & 3, pb0 = ((tq >> 2) / N) * N;
            float v[16];
            const lfloat* const xcol = xst + pb0 * 4 + cc;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = xcol[4 * i];
            const float own = xst[tq];
            const float s1 = colsum16(v, 0.f);
            const float m1 = s1 / (float)N;
            const float xc = own - m1;
            float xn_in = xc;
            if (MODE == DFF_MODE_LANGEVIN) {
                const float s2 = colsum16(v, m1);
                xn_in = xc - s2 / (float)N;
                xcb[tq] = xc;
            }
            xs[tq] = xn_in;
            // FOLD: K_ext = V_ext is one shared buffer whose extension columns [x_j | 0 ...] no layer rewrites: once per step
            // here instead of by every wave in every attention block (rows beyond the real ones and column 3 stay zero)
            if constexpr (FOLD) { if (cc < 3) Nx[(tq >> 2) * XLD + 64 + cc] = xn_in; }
        }
    };
    auto step_prefetch = [&](bool cached0_, const gfloat* l0e_) {
        if constexpr (SPW) {
            const int lane = lane_id();
            if (cached0_) {
                sring_prefetch<SDR, KO, E>(sring, ss_wox(m.layer[0], wave), lane);
                if constexpr (HDMA) dma_nop(l0e_ + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, lane);
            } else sring_prefetch<SDR, KQ, E>(sring, ss_qkv(m.layer[0], wave), lane);
        }
    };
    // this (bead, component) thread's mass and noise amplitude, once per launch: indexed per lane they are global loads from the
    // kernel-argument segment, whose latency the update stage of every step used to wait out in front of the whole workgroup
    float mass_i = 1.0f, nsig_i = 0.0f;
    if (MODE == DFF_MODE_LANGEVIN && tid < rows * 4) {
        const int i_ = (tid >> 2) % N;
        mass_i = a.mass[i_]; nsig_i = a.noise_sigma[i_];
    }
    for (int step = 0; step < a.n_steps; ++step) {
        int t_int = 0;
        // (the reverse step's schedule constants: requested here, a whole network evaluation before wave 0's update needs them)
        float sch_sr = 0.f, sch_srm1 = 0.f, sch_c1 = 0.f, sch_c2 = 0.f, sch_lv = 0.f;
        if (MODE == DFF_MODE_DDPM) {
            t_int = a.t_start - step;
            if (tid < gcnt) tn[tid] = (1.0f * (float)t_int) / (float)m.T;
            sch_sr = m.sqrt_recip_ac[t_int]; sch_srm1 = m.sqrt_recipm1_ac[t_int];
            sch_c1 = m.post_c1[t_int]; sch_c2 = m.post_c2[t_int]; sch_lv = m.post_logvar[t_int];
        }
        // Layer-0 inputs are x-independent: with a precomputed table entry for this step's t
        // (one entry per noise level, shared by all workgroups and L2-resident; built by the host with this
        // very kernel, see ensure_l0_table) layer 0 never runs its QKV GEMM.  Without a table, Langevin
        // (fixed t) still re-reads what step 0 left in this workgroup's own stash.
        const bool tab = a.l0_tab != nullptr;
        const gfloat* const l0e = tab ? (const gfloat*)a.l0_tab + (size_t)(MODE == DFF_MODE_DDPM ? t_int : 0) * sl.layer_stride
                                      : (const gfloat*)stash;
        const bool full0 = GEN && m.in_abs;   // absolute coordinates: layer 0 depends on x (no caching, VJP through layer 0)
        const bool cached0 = !full0 && (tab || ((MODE == DFF_MODE_LANGEVIN) && step > 0));
        // the step after this one: is there one, does its layer 0 come from the table / the stash, and from which entry
        const bool nxt = step + 1 < a.n_steps;
        const bool c0n = !full0 && (tab || MODE == DFF_MODE_LANGEVIN);
        const gfloat* const l0n = tab ? (const gfloat*)a.l0_tab + (size_t)(MODE == DFF_MODE_DDPM ? a.t_start - step - 1 : 0) * sl.layer_stride
                                      : (const gfloat*)stash;
        bool early_done = false;   // the backward sweep requested the next step's first weights / head rows (EARLY)
        // reverse DDPM on the table: the NEXT step's row stage A (node inputs of noise level t - 1 -> LayerNorm -> resbuf / nx) has no
        // input of this step's: stage E of layer 0 does it (its loads issued at the top of the stage), the next step starts with its
        // attention block
        const bool hoistA = KEEPROWS && MODE == DFF_MODE_DDPM && tab && !full0 && m.conservative && nxt;
        // First weights of the first block, and layer 0's q_ext | k | v rows of this head (shared table entry) by LDS-DMA
        // (measured: 86.8 vs 87.3 us / step with a register fetch inside the attention block).  SPW: step 0 only -- every
        // later step's were requested in the shadow of the update stage of the step before it (step_prefetch).
        { const int lane = lane_id();
        if constexpr (SPW) {
            if (step == 0) step_prefetch(cached0, l0e);
        } else if (cached0) {
            const gfloat* sb0 = l0e;
            if constexpr (HPW == 2) head_fetch(hr, sb0 + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, nullptr, RA, true, lane);
            ring_prefetch<E>(ring, s_wox(m.layer[0], wave), lane);
        } else {
            ring_prefetch<E>(ring, s_qkv(m.layer[0], wave), lane);
        } }
        if (step == 0) centre();
        if (step == 0 || !cached0) __syncthreads();   // (later steps: the barrier that ended the previous update stage)
        pf.tick(0); DFF_MARK(0);

        // =============================== forward ===============================
        if (!cached0) {
            for (int idx = tid; idx < rows * H; idx += NTHR) {
                const int row = idx / H, cl = idx - row * H;
                const int g = row / N, i = row - g * N;
                float nv = m.WnT[i * H + cl] + tn[g] * m.WnT[(GEN ? m.wn_t : N) * H + cl] + m.bn[cl];
                if constexpr (GEN) {
                    if (m.in_abs) nv += xs[row * 4] * m.WnT[N * H + cl] + xs[row * 4 + 1] * m.WnT[(N + 1) * H + cl] +
                                        xs[row * 4 + 2] * m.WnT[(N + 2) * H + cl];
                }
                resbuf[row * LH + cl] = nv;
            }
            __syncthreads();
        }
        for (int l = 0; l < m.L; ++l) {
            const DffLayerDev& lw = m.layer[l];
            gfloat* const sb = stash + (size_t)l * sl.layer_stride;
            const gfloat* const sbq = l == 0 ? l0e : (const gfloat*)sb;   // where this layer's nodes_in / q|k|v are read from
            const bool cached = cached0 && l == 0;
            // ---- row stage A (layer 0 only; later layers get LN1 fused into stage C) ----
            // KEEPROWS at a fixed noise level (Langevin), steps after the first: the stage would only copy layer 0's kept node
            // inputs to resbuf (its LayerNorm rows are still in `nx`: stage E of layer 0 put them back for its backward attention
            // block, and nothing wrote there since) -- stage E of the step before did that too (`skipA` below): no stage, no barrier
            // (reverse DDPM with the per-noise-level table, round 6: stage E of layer 0 of the step before loaded THIS step's node inputs,
            // took their LayerNorm and left the rows in the kept registers; the update stage put them into `nx`: `hoistA` below)
            const bool skipA = KEEPROWS && step > 0 && cached0 && m.conservative && (MODE == DFF_MODE_LANGEVIN || (MODE == DFF_MODE_DDPM && tab));
            if (l == 0 && !skipA) {
                DFF_ROW_CONSTS
                // KEEPROWS: layer 0's node inputs and LayerNorm rows are kept in this thread's registers for the backward stages;
                // with a fixed noise level (Langevin) they do not change from step to step: read / computed on step 0 only
                const bool reuse0 = KEEPROWS && MODE == DFF_MODE_LANGEVIN && step > 0;
                if (cached) {
                    if (ract) {
                        float x[HC], nva[HC];
                        if (reuse0) {
                            keep_get(LSel{true, false}, KN{}, x);
                            keep_get(LSel{true, false}, KL{}, nva);
                        } else {
#pragma unroll
                            for (int i = 0; i < HC; ++i) x[i] = ld_ntg(sbq + sl.nodes_in + rrow * H + sub + LPR * i);
                            if constexpr (FOLD) {   // the attention block needs layer 0's LayerNorm rows even when its q' comes from the table
                                float mean, rstd;
                                ln_stats_row(x, mean, rstd);
                                if constexpr (KEEPROWS) { gate_put(0, 2, mean); gate_put(0, 3, rstd); }
#pragma unroll
                                for (int i = 0; i < HC; ++i) {
                                    const int cl = sub + LPR * i;
                                    nva[i] = (x[i] - mean) * rstd * lw.ln1_g[cl] + lw.ln1_b[cl];
                                }
                            }
                            if constexpr (KEEPROWS) { keep_put(0, KN{}, x); keep_put(0, KL{}, nva); }
                        }
#pragma unroll
                        for (int i = 0; i < HC; ++i) {
                            resbuf[rs_o + LPR * i] = x[i];
                            if constexpr (FOLD) {
                                n_put(i, nva[i]);   // (fp32 row + its fp16 pieces: the logits' 64-column part runs on the
                                                                        // pieces whether q' was just computed or comes from the table)
                            }
                        }
                    }
                } else if (ract) {
                    float x[HC], nva[HC];
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        x[i] = resbuf[rs_o + LPR * i];
                        st_ntg(sb + sl.nodes_in + rrow * H + sub + LPR * i, x[i]);
                    }
                    float mean, rstd;
                    ln_stats_row(x, mean, rstd);
                    if constexpr (KEEPROWS) { gate_put(0, 2, mean); gate_put(0, 3, rstd); }
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        nva[i] = (x[i] - mean) * rstd * lw.ln1_g[cl] + lw.ln1_b[cl];
                        if constexpr (!NSP) a_put(i, nva[i]);
                        n_put(i, nva[i]);
                    }
                    if constexpr (KEEPROWS) { keep_put(0, KN{}, x); keep_put(0, KL{}, nva); }
                }
                if (ract) pre_B(lw, sub);
                if constexpr (DFF_XI_STAGE == 0) draw_xi(t_int, step);
                __syncthreads();
            }
            pf.tick(1); DFF_MARK(1);
            // ---- attention block: wave w owns heads w and w+4 (ring holds the first entries) ----
            {
                DFF_LANE_CONSTS
                ro_touch();
                f32x4 acc_o[E];
#pragma unroll
                for (int nt = 0; nt < E; ++nt) acc_o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const WStream after = s_w1(lw);   // the FFN block follows
                // the sampling loops never read the last layer's q_ext | k | v | P back from the stash (KEEP_LAST)
                const bool keep2 = KEEP2 && l == m.L - 2 && l > 0 && MODE != DFF_MODE_SCORE;
                const bool p0keep = KEEP2 && l == 0 && m.L > 2 && MODE != DFF_MODE_SCORE && rows <= 10;   // layer 0's P: p0_copy
                const bool st_qkv = !(KEEP_LAST && l == m.L - 1 && l > 0 && MODE != DFF_MODE_SCORE) && !keep2;
                const SStream sn0 = ss_w1(lw), sn1 = ss_w2(lw);   // the FFN block's units follow: W1 (U_W1), then W2
                auto head_math = [&](int h) {
                    if constexpr (!FOLD) write_xext(lane);   // (FOLD: the shared buffer's x columns are written once per step, by the centring)
                    if constexpr (GEN) fix_q(lane);
                    // Layer 0 at a fixed noise level (Langevin): q' and the LayerNorm rows are the same every step, so the 64
                    // regular columns of the logits are too -- computed on step 0 of the launch, kept in 4 registers; only the
                    // extension k-step (u . x_j) is redone
                    // x_i of this lane's four C-layout rows (for xrel below), requested before the products instead of
                    // after them (volatile: the compiler would sink the reads to their use in the last tile's epilogue)
                    float xsub[4];
                    if constexpr (!GEN) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xsub[r] = *(const volatile lfloat*)(xs + (quad * 4 + r) * 4 + (col & 3));
                    }
                    f32x4 Sb;
                    if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0 && step > 0) Sb = s0keep;
                    else {
                        Sb = wv_dot_base<XLD>(Qx, Kx, lane);
                        if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0) s0keep = Sb;
                    }
                    const f32x4 S = wv_dot_ext<XLD>(Qx, Kx, lane, Sb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = quad * 4 + r;
                        const bool ok = (okmask >> r) & 1u;   // (col < rows) && (i < rows) && (i / N == col / N), once per kernel
                        const float s = ok ? S[r] * 0.125f : -INFINITY;
                        const float mx = row16_max(s);
                        const float e = ok ? fast_exp(s - mx) : 0.f;
                        const float den = row16_sum(e);
                        const float p = den > 0.f ? e * fast_rcp(den) : 0.f;
                        pb[i * DFF_PLD + col] = p;
                    }
                    // the 16 x 16 tile to the stash as ONE 16-byte store per lane (was four scalar stores)
                    if (p0keep) p0_copy(true, pcij);
                    else if (st_qkv) *(gf32x4*)(sb + sl.P + (size_t)h * 256 + 4 * lane) = *(const lf32x4*)(pb + (lane >> 2) * DFF_PLD + 4 * (lane & 3));
                    // O_ext = P V_ext (5 tiles) -> Q region; extension columns become xrel = xbar - x_i
                    wv_mm5<(DFF_MM_SPLIT && H > 64), false, XLD>(pb, Vx, lane, ks4, [&](int nt, const f32x4& acc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[r];
                            if constexpr (GEN) {
                                if (nt == 4) {   // [m1 | m2] -> xrel = m1 - x_i ; D = |x_i|^2 - 2 x_i.m1 + m2 ; stash [m1 | m2]
                                    const int row = quad * 4 + r;
                                    const float x0 = xs[row * 4], x1 = xs[row * 4 + 1], x2 = xs[row * 4 + 2];
                                    const float xr = col == 0 ? x0 : col == 1 ? x1 : col == 2 ? x2 : 0.f;
                                    const float tq = col < 3 ? -2.0f * xr * v : col == 3 ? v + (x0 * x0 + x1 * x1 + x2 * x2) : 0.f;
                                    const float D = quad_sum(tq);
                                    if (col < 4) st_ntg(sb + sl.m12 + (h * 16 + row) * 4 + col, v);
                                    v = col < 3 ? v - xr : col == 3 ? D : 0.f;
                                }
                            } else {
                                if (nt == 4) v -= (col < 3) ? xsub[r] : 0.f;
                            }
                            Ox[lro[r] + 16 * nt + col] = v;
                        }
                    });
                };
                const lfloat* const wox_a = Ox + col * XLD + 4 * quad;
                auto wox_fa = [=](int kb) { return wox_a + 16 * kb; };
                auto wox_fa32 = [=](int kb) { return wox_a + 32 * kb; };
                const lfloat* const wox_xa = Ox + col * XLD + 64 + quad;   // extension column `quad` of row `col`
                // REGCHAIN (FOLD kernel on the fp16 engine, round 6): QKV' -> S -> softmax -> P.V -> [W_o;W_oc] with every hand-over in
                // registers.  The logits are computed TRANSPOSED, S^T[j][i] = n_j . q'_i: lane (i = col, quad) then holds row i's
                // logits for the keys j = 4 quad .. + 3, the softmax of a row is an in-lane reduction + two lane swaps, and the
                // probabilities are, as they stand, the B operand of O^T = V_ext^T P^T (k-step r contracts j = 4 kk + r: any
                // permutation of the contraction index does, applied to both operands) -- whose output tiles are the A fragments of
                // the output projection.  q' and P still go to LDS (the backward's operands), off the chain.
                constexpr bool REGCHAIN = FOLD && F16E;
                const int lroT = min(col, RLA - 1) * XLD;   // this lane's row in the transposed-output tiles (pad rows -> the dummy row)
                auto head_rest = [&](int h, const f32x4 Sb, f32x4 (&ot)[5]) {
                    // extension k-step: S^T[j][i] += x_j . u_i
                    const f32x4 S = wv_dot_ext<XLD>(Kx, Qx, lane, Sb);
                    float sv[4], ev[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[r] = ((okmask >> r) & 1u) ? S[r] * 0.125f : -INFINITY;   // (the mask is symmetric in i, j)
                    auto quads = [](float v, auto op) {   // all-reduce over lanes l, l ^ 16, l ^ 32, l ^ 48
                        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                        v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
                        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                        return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
                    };
                    const float mx = quads(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), [](float a_, float b_) { return fmaxf(a_, b_); });
#pragma unroll
                    for (int r = 0; r < 4; ++r) ev[r] = ((okmask >> r) & 1u) ? fast_exp(sv[r] - mx) : 0.f;
                    const float den = quads((ev[0] + ev[1]) + (ev[2] + ev[3]), [](float a_, float b_) { return a_ + b_; });
                    const float inv = den > 0.f ? fast_rcp(den) : 0.f;
                    f32x4 p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = ev[r] * inv;
                    *(lf32x4*)(pb + col * DFF_PLD + 4 * quad) = p;   // P[i = col][j = 4 quad ..]: the backward's layout
                    if (p0keep) p0_copy(true, pcij);
                    else if (st_qkv) *(gf32x4*)(sb + sl.P + (size_t)h * 256 + 4 * lane) = *(const lf32x4*)(pb + (lane >> 2) * DFF_PLD + 4 * (lane & 3));
                    // O^T = V_ext^T P^T.  The 64 regular columns on the fp16 pipe: the probabilities of lane (i, kg) ARE the B operand of a
                    // v_mfma_f32_16x16x16_f16 (k = j = 4 kg ..), split here into two pieces; A = the transposed pieces of the LayerNorm rows
                    // (region nst: 8 bytes per lane, tile and piece).  The extension tile (x-bar) stays on the fp32 pipe:
                    // A = V_ext[j = 4 kk + r][64 + m], B = p[r].
                    const f32x4 xi4 = *(const lf32x4*)(xs + col * 4);   // x_i of this lane's row (component 3 is zero)
                    typedef const volatile lfloat* vlp;
                    float bvx[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bvx[r] = *(vlp)(Vx + (4 * quad + r) * XLD + 64 + col);
                    f16x4 vh[4], vl[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) nt_load(nt, lane, vh[nt], vl[nt]);
                    f16x4 ph, pl;
                    {
                        unsigned h0, l0, h1, l1;
                        split2h(p[0], p[1], h0, l0); split2h(p[2], p[3], h1, l1);
                        ph = __builtin_bit_cast(f16x4, (u32x2){h0, h1}); pl = __builtin_bit_cast(f16x4, (u32x2){l0, l1});
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
                        cs = __builtin_amdgcn_mfma_f32_16x16x16f16(vl[nt], ph, cs, 0, 0, 0);
                        cs = __builtin_amdgcn_mfma_f32_16x16x16f16(vh[nt], pl, cs, 0, 0, 0);
                        cb = __builtin_amdgcn_mfma_f32_16x16x16f16(vh[nt], ph, cb, 0, 0, 0);
                        ot[nt] = cb + cs * DFF_F16_LINV;
                    }
                    ot[4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(bvx[r], p[r], ot[4], 0, 0, 0);
                    // extension tile: [xbar | 0 ...] -> xrel = xbar - x_i (quad 0: columns 64 .. 67), to LDS for the projection's fp32 k-step
                    if (quad == 0) ot[4] -= xi4;
                    *(lf32x4*)(Ox + lroT + 64 + 4 * quad) = ot[4];
                };
                if constexpr (SPW && REGCHAIN) {
                    f32x4 ot[5];
                    if (cached) {
                        if constexpr (HDMA) head_dma_wait();
                        pf.tick(12); DFF_MARK(12);
                        // layer 0 at a fixed noise level (Langevin): q' and the LayerNorm rows are the same every step, so is the
                        // 64-column part of the logits -- computed on step 0 of the launch (fp32 products on the table's q' rows)
                        f32x4 Sb;
                        if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0 && step > 0) Sb = s0keep;
                        else {
                            // the same fp16 products as below, on the table's q' rows (fp32 in the Q region) and the pieces row stage A
                            // made of the LayerNorm rows: bit-identical to the path that computes q' itself
                            u32x4 nh[KB32], nl[KB32];
                            an_load(nh, nl, lane);
                            f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) {
                                u32x4 qh, ql;
                                const f32x4 q0 = *(const lf32x4*)(Qx + lroT + 32 * kb + 4 * quad), q1 = *(const lf32x4*)(Qx + lroT + 32 * kb + 16 + 4 * quad);
                                split8h(q0, q1, qh, ql);
                                cs = mfma_f16(nl[kb], qh, cs);
                                cs = mfma_f16(nh[kb], ql, cs);
                                cb = mfma_f16(nh[kb], qh, cb);
                            }
                            Sb = cb + cs * DFF_F16_LINV;
                            if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0) s0keep = Sb;
                        }
                        head_rest(wave, Sb, ot);
                        pf.tick(13); DFF_MARK(13);
                        const SSeq<U_WOX, 0, MW, KO, KO, KS, KS, E> sq{ss_wox(lw, wave), ss_wox(lw, wave), sn0, sn1};
                        const f32x4 (&o4)[4] = *reinterpret_cast<const f32x4 (*)[4]>(&ot[0]);
                        stallR_run<0, 2, E, true>(sring, acc_o, o4, sq, lane, wox_xa, wox_ext(lw, wave, lane), DFF_HEADS * 5 * 256);
                        pf.tick(14); DFF_MARK(14);
                    } else {
                        u32x4 ah[KB32], am[KB32];
                        an_load(ah, am, lane);
                        const gfloat* const bp = (const gfloat*)lw.bqkvx + wave * 13 * 16 + 4 * quad;
                        f32x4 bq[2] = {*(const gf32x4*)bp, *(const gf32x4*)(bp + 16)};
                        pf.tick(1); DFF_MARK(1);
                        const SSeq<U_QKV, U_WOX, MW, KQ, KO, KS, KS, E, 4 * KB32> sq{ss_qkv(lw, wave), ss_wox(lw, wave), sn0, sn1};   // tile 4 of 5: [u | s]
                        f32x4 qt[5];
                        swideT_from<0, 0, NQT, KB32>(sring, qt, bq, bp, ah, am, sq, lane);
                        // q'_ext rows -> Q region: the backward's dK operand, the source of the stash / Qsave copies and of the
                        // extension-column reads; one 16-byte store per tile
#pragma unroll
                        for (int t = 0; t < 5; ++t) *(lf32x4*)(Qx + lroT + 16 * t + 4 * quad) = qt[t];
                        if (st_qkv) head_store<FOLD>(Qx, Kx, Vx, sb + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, RA, lane, RLA, XLD);
                        pf.tick(12); DFF_MARK(12);
                        // S^T = n q'^T on the fp16 pipe: both operands are in registers already (the LayerNorm rows' pieces were this
                        // GEMM's A operand, q' is its output)
                        f32x4 Sb;
                        {
                            u32x4 qh[2], ql[2];
                            split8h(qt[0], qt[1], qh[0], ql[0]);
                            split8h(qt[2], qt[3], qh[1], ql[1]);
                            f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) {
                                cs = mfma_f16(am[kb], qh[kb], cs);
                                cs = mfma_f16(ah[kb], ql[kb], cs);
                                cb = mfma_f16(ah[kb], qh[kb], cb);
                            }
                            Sb = cb + cs * DFF_F16_LINV;
                        }
                        if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0) s0keep = Sb;
                        head_rest(wave, Sb, ot);
                        if constexpr (KEEP2) { if (keep2) keep2_copy(Qx, Qsave, true, lane, pcij); }
                        pf.tick(13); DFF_MARK(13);
                        const f32x4 (&o4)[4] = *reinterpret_cast<const f32x4 (*)[4]>(&ot[0]);
                        stallR_run<U_QKV, 2, E, true>(sring, acc_o, o4, sq, lane, wox_xa, wox_ext(lw, wave, lane), DFF_HEADS * 5 * 256);
                        pf.tick(14); DFF_MARK(14);
                    }
                } else if constexpr (SPW) {
                    if (cached) {
                        if constexpr (HDMA) head_dma_wait();
                        else {
                            head_fetch(hr, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, nullptr, RA, true, lane);
                            head_commit(hr, Qx, Kx, Vx, pb, true, false, lane, RLA, XLD);
                        }
                        pf.tick(12); DFF_MARK(12);
                        head_math(wave);
                        pf.tick(13); DFF_MARK(13);
                        const SSeq<U_WOX, 0, MW, KO, KO, KS, KS, E> sq{ss_wox(lw, wave), ss_wox(lw, wave), sn0, sn1};
                        stall_run<0, 2, E, true>(sring, acc_o, wox_fa32, sq, lane, wox_xa, wox_ext(lw, wave, lane), DFF_HEADS * 5 * 256);
                        pf.tick(14); DFF_MARK(14);
                    } else {
                        u32x4 ah[KB32], am[KB32], al[KB32];
                        a_load(ah, am, al, lane);
                        const gfloat* const bqkvx = (const gfloat*)lw.bqkvx;
                        const int l0 = lro[0], l1 = lro[1], l2 = lro[2], l3 = lro[3];
                        const gfloat* const bh = bqkvx + wave * 13 * 16 + col;
                        lfloat* const wq = wr + col;
                        float bq[2][1];
                        bq[0][0] = bh[0]; bq[1][0] = bh[16];
                        pf.tick(1); DFF_MARK(1);
                        const SSeq<U_QKV, U_WOX, MW, KQ, KO, KS, KS, E, 4 * KB32> sq{ss_qkv(lw, wave), ss_wox(lw, wave), sn0, sn1};   // tile 4 of 13: [u | s]
                        swide_run<0, NQT, KB32, 1>(sring, bq, ah, am, al, sq, lane,
                            [=](int t, float (&ax)[1]) { ax[0] = bh[t * 16]; },
                            [=](int t, const f32x4& acc, const float (&ax)[1]) {
                                const int reg = (t >= 5) + (t >= 9);
                                const int cl = 16 * (t - 5 * reg + (reg >> 1));
                                lfloat* const dl = wq + reg * RS + cl;
                                const float v0 = acc[0] + ax[0], v1 = acc[1] + ax[0], v2 = acc[2] + ax[0], v3 = acc[3] + ax[0];
                                dl[l0] = v0; dl[l1] = v1; dl[l2] = v2; dl[l3] = v3;
                            });
                        if (st_qkv) head_store<FOLD>(Qx, Kx, Vx, sb + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, RA, lane, RLA, XLD);
                        pf.tick(12); DFF_MARK(12);
                        head_math(wave);
                        if constexpr (KEEP2) { if (keep2) keep2_copy(Qx, Qsave, true, lane, pcij); }
                        pf.tick(13); DFF_MARK(13);
                        stall_run<U_QKV, 2, E, true>(sring, acc_o, wox_fa32, sq, lane, wox_xa, wox_ext(lw, wave, lane), DFF_HEADS * 5 * 256);
                        pf.tick(14); DFF_MARK(14);
                    }
                } else if (cached) {
                    // layer-0 q_ext / k / v are x-independent and t is fixed: re-read, no GEMM
                    if constexpr (HPW == 2) {
                        head_commit(hr, Qx, Kx, Vx, pb, true, false, lane, RLA, XLD);
                        head_fetch(hr, sbq + sl.qkv + (size_t)(wave + 4) * (RA + 1) * DFF_QKVW, nullptr, RA, true, lane);
                        head_math(wave);
                        tall_run<0, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave), s_wox(lw, wave + 4), lane);
                        head_commit(hr, Qx, Kx, Vx, pb, true, false, lane, RLA, XLD);
                        head_math(wave + 4);
                        tall_run<1, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave + 4), after, lane);
                    } else {
                        head_fetch(hr, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, nullptr, RA, true, lane);
                        head_commit(hr, Qx, Kx, Vx, pb, true, false, lane, RLA, XLD);
                        head_math(wave);
                        tall_run<0, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave), after, lane);
                    }
                    // the ring did not end on phase 0: re-stage the next block's first entries
                    ring_prefetch<E>(ring, after, lane);
                } else {
                    f32x4 afr[E];
                    load_afrag<E>(afr, abuf, LH, lane);
                    const gfloat* const bqkvx = (const gfloat*)lw.bqkvx;
                    const int s0 = srow[0], s1 = srow[1], s2 = srow[2], s3 = srow[3];
                    const int l0 = lro[0], l1 = lro[1], l2 = lro[2], l3 = lro[3];
                    // bias of tile d goes to the aux slot of ring phase ph
                    auto qkv_bias = [&](auto ph, int h, float (&bq)[DR][1]) {
#pragma unroll
                        for (int d = 0; d < DR; ++d) bq[(decltype(ph)::value + d) % DR][0] = bqkvx[(h * 13 + d) * 16 + col];
                    };
                    auto qkv = [&](auto ph, int h, float (&bq)[DR][1], const WStream& wnext) {
                        gfloat* const sqkv = sb + sl.qkv + (size_t)h * (RA + 1) * DFF_QKVW + col;
                        const gfloat* const bh = bqkvx + h * 13 * 16 + col;
                        lfloat* const wq = wr + col;
                        wide_run<decltype(ph)::value, 13, E, 1>(ring, bq, afr, s_qkv(lw, h), wnext, lane,
                            [=](int t, float (&ax)[1]) { ax[0] = bh[t * 16]; },
                            [=](int t, const f32x4& acc, const float (&ax)[1]) {
                                // tiles 0-4 -> Q_ext, 5-8 -> K, 9-12 -> V (t is wave-uniform); pure
                                // arithmetic, no pointer select (a 3-way select becomes a stack table)
                                const int reg = (t >= 5) + (t >= 9);
                                const int cl = 16 * (t - 5 * reg + (reg >> 1));
                                lfloat* const dl = wq + reg * RS + cl;
                                gfloat* const ds = sqkv + 16 * t;
                                const float v0 = acc[0] + ax[0], v1 = acc[1] + ax[0], v2 = acc[2] + ax[0], v3 = acc[3] + ax[0];
                                dl[l0] = v0; dl[l1] = v1; dl[l2] = v2; dl[l3] = v3;
                                st_ntg(ds + s0 * DFF_QKVW, v0); st_ntg(ds + s1 * DFF_QKVW, v1);
                                st_ntg(ds + s2 * DFF_QKVW, v2); st_ntg(ds + s3 * DFF_QKVW, v3);
                            });
                    };
                    float bq[DR][1];
                    qkv_bias(std::integral_constant<int, 0>{}, wave, bq);
                    pf.tick(1); DFF_MARK(1);
                    if constexpr (HPW == 2) {
                        qkv(std::integral_constant<int, 0>{}, wave, bq, s_wox(lw, wave));
                        pf.tick(12); DFF_MARK(12);
                        qkv_bias(std::integral_constant<int, 2>{}, wave + 4, bq);
                        head_math(wave);
                        pf.tick(13); DFF_MARK(13);
                        tall_run<1, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave), s_qkv(lw, wave + 4), lane);
                        pf.tick(14); DFF_MARK(14);
                        qkv(std::integral_constant<int, 2>{}, wave + 4, bq, s_wox(lw, wave + 4));
                        pf.tick(12); DFF_MARK(12);
                        head_math(wave + 4);
                        pf.tick(13); DFF_MARK(13);
                        tall_run<3, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave + 4), after, lane);   // ends at phase 0
                        pf.tick(14); DFF_MARK(14);
                    } else {
                        qkv(std::integral_constant<int, 0>{}, wave, bq, s_wox(lw, wave));
                        pf.tick(12); DFF_MARK(12);
                        head_math(wave);
                        pf.tick(13); DFF_MARK(13);
                        tall_run<13 % DR, 5, E, 4>(ring, acc_o, wox_fa, s_wox(lw, wave), after, lane);       // 18 entries: phase 0
                        pf.tick(14); DFF_MARK(14);
                        static_assert(18 % DR == 0, "ring phase");
                    }
                }
#pragma unroll
                for (int nt = 0; nt < E; ++nt) c_store_offs(mypart, mro, 16 * nt, acc_o[nt], lane);
            }
            __syncthreads();
            pf.tick(2); DFF_MARK(2);
            // ---- row stage B: attn_out = sum_w part + bo ; gate1 ; LN2 -> abuf ----
            { DFF_ROW_CONSTS
            float psx[HC];
            if constexpr (PAIR) pair_rows(psx, rrow, sub, ract, tq_);   // (this block's four partials + the partner's: barriers inside)
            if (ract) {
                float x[HC], res[HC], n1[HC];
                float ps[HC];
                if constexpr (PAIR) { for (int i = 0; i < HC; ++i) ps[i] = psx[i]; } else
                psum_all(ps, rs_o);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + LPR * i, o = rs_o + LPR * i;
                    x[i] = ps[i] + ro[0][i];
                    res[i] = resbuf[o];
                    if constexpr (!KEEPROWS) st_ntg(sb + sl.attn_out + rrow * H + cl, x[i]);
                }
                if constexpr (KEEPROWS) keep_put(l, KA{}, x);
                const float g = ro_gate(x, res, 1);
                if constexpr (KEEPROWS) gate_put(l, 0, g);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    n1[i] = x[i] * g + res[i] * (1.0f - g);
                    resbuf[rs_o + LPR * i] = n1[i];
                }
                float mean, rstd;
                ln_stats_row(n1, mean, rstd);
                if constexpr (KEEPROWS) { gate_put(l, 4, mean); gate_put(l, 5, rstd); }
#pragma unroll
                for (int i = 0; i < HC; ++i) a_put(i, (n1[i] - mean) * rstd * ro[4][i] + ro[5][i]);
                // stage C operands: b2, g2 (3), and the next layer's LN1 gamma / beta
                ro_load(0, lw.b2, sub); ro_load3(1, lw.g2, sub);
                if (l + 1 < m.L) { ro_load(4, m.layer[l + 1].ln1_g, sub); ro_load(5, m.layer[l + 1].ln1_b, sub); }
            } }
            if constexpr (DFF_XI_STAGE == 1) { if (l == 0) draw_xi(t_int, step); }
            else { if (l == 0 && skipA) draw_xi(t_int, step); }   // (no stage A to draw under)
            __syncthreads();
            pf.tick(3); DFF_MARK(3);
            // ---- FFN: wave w owns hidden columns [w H, (w+1) H) ----
            {
                DFF_LANE_CONSTS
                ro_touch();
                const bool lastl = l == m.L - 1;
                const WStream after = lastl ? s_w2t(lw) : s_qkv(m.layer[lastl ? l : l + 1], wave);
                // what follows: the last layer's FFN backward (W2^T, W1^T) or the next layer's QKV_ext (whose units MW.. are "n1")
                const SStream qn = ss_qkv(m.layer[lastl ? l : l + 1], wave);
                const SSeq<U_W1, U_W2, MW, KS, KS, KS, KS, E> sqf_last{ss_w1(lw), ss_w2(lw), ss_w2t(lw), ss_w1t(lw)};
                const SSeq<U_W1, U_W2, SDR, KS, KS, KQ, KQ, E> sqf_next{ss_w1(lw), ss_w2(lw), qn, qn};
                f32x4 afr[E];
                if constexpr (!SPW) load_afrag<E>(afr, abuf, LH, lane);
                float b1r[DR][1];
                const gfloat* const b1p = (const gfloat*)lw.b1 + wave * FS + col;
#pragma unroll
                for (int d = 0; d < DR; ++d) b1r[d][0] = b1p[16 * d];
                // FOLD, sampling loops: GELU'(h_pre) of the last three layers never leaves the LDS (gp_tile); otherwise it goes to
                // the stash slot "h_pre" through a second LDS tile behind the hidden slice
                constexpr bool gp_lds = FOLD && MODE != DFF_MODE_SCORE;
                {
                    lfloat* const hb = hbuf + quad * 4 * LF + col;
                    lfloat* const gq = (gp_lds ? gp_tile(l, m.L) : hbuf + 16 * LF) + col;
                    int go[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) go[r] = gp_lds ? min(quad * 4 + r, LL::GPR - 1) * LL::GPS : (quad * 4 + r) * LF;
                    const int go0 = go[0], go1 = go[1], go2 = go[2], go3 = go[3];
                    auto w1_pre = [=](int t, float (&ax)[1]) { ax[0] = b1p[16 * t]; };
                    auto w1_epi = [=](int t, const f32x4& acc, const float (&ax)[1]) {
                            float g0, g1, g2, g3, p0, p1, p2, p3;
                            gelu_both(acc[0] + ax[0], g0, p0); gelu_both(acc[1] + ax[0], g1, p1);
                            gelu_both(acc[2] + ax[0], g2, p2); gelu_both(acc[3] + ax[0], g3, p3);
                            // On its way to the stash gelu'(h_pre) travels through the second LDS tile and leaves as full
                            // 128-byte rows after the W2 GEMM (below): stored from here, 64 bytes per row and instruction, each
                            // store was a partial-line write whose acknowledgement the W2 weights queued behind in the in-order
                            // vmcnt queue (a timing-only build without the stores: -1.1 us / step)
                            hb[16 * t] = g0; hb[LF + 16 * t] = g1; hb[2 * LF + 16 * t] = g2; hb[3 * LF + 16 * t] = g3;
                            gq[go0 + 16 * t] = p0; gq[go1 + 16 * t] = p1; gq[go2 + 16 * t] = p2; gq[go3 + 16 * t] = p3;
                        };
                    if constexpr (NSP) {
                        // (register chain, below)
                    } else if constexpr (SPW) {
                        u32x4 ah[KB32], am[KB32], al[KB32];
                        a_load(ah, am, al, lane);
                        pf.tick(19); DFF_MARK(19);
                        if (lastl) swide_run<0, NTS, KB32, 1>(sring, b1r, ah, am, al, sqf_last, lane, w1_pre, w1_epi);
                        else swide_run<0, NTS, KB32, 1>(sring, b1r, ah, am, al, sqf_next, lane, w1_pre, w1_epi);
                        pf.tick(20); DFF_MARK(20);
                    } else
                    wide_run<0, NTS, E, 1>(ring, b1r, afr, s_w1(lw), s_w2(lw), lane, w1_pre, w1_epi);
                }
                f32x4 acc_f[E];
#pragma unroll
                for (int nt = 0; nt < E; ++nt) acc_f[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                {
                    const lfloat* const ha = hbuf + col * LF + 4 * quad;
                    if constexpr (NSP) {
                        // W1 -> GELU -> W2 in registers: the W1 slice with its operands swapped leaves lane (row = col, quad) with hidden
                        // columns 16 t + 4 quad .. of that row -- the A fragment of W2's one 32-column block; GELU' goes to its LDS tile
                        // ([row][32 hidden columns], as before) with one 16-byte store per tile
                        static_assert(NTS == 2 && FS == 32, "one 32-column k-block per wave");
                        u32x4 ah[KB32], am[KB32], al[KB32];
                        a_load(ah, am, al, lane);
                        const gfloat* const b1q = (const gfloat*)lw.b1 + wave * FS + 4 * quad;
                        f32x4 bq[2] = {*(const gf32x4*)b1q, *(const gf32x4*)(b1q + 16)};
                        pf.tick(19); DFF_MARK(19);
                        f32x4 ht[2], gd[2];
                        if (lastl) swideT_from<0, 0, NTS, KB32>(sring, ht, bq, b1q, ah, am, sqf_last, lane);
                        else swideT_from<0, 0, NTS, KB32>(sring, ht, bq, b1q, ah, am, sqf_next, lane);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { float g, gp; gelu_both(ht[t][r], g, gp); ht[t][r] = g; gd[t][r] = gp; }
                        lfloat* const gq = gp_lds ? gp_tile(l, m.L) + min(col, LL::GPR - 1) * LL::GPS : hbuf + 16 * LF + col * LF;
                        *(lf32x4*)(gq + 4 * quad) = gd[0];
                        *(lf32x4*)(gq + 16 + 4 * quad) = gd[1];
                        pf.tick(20); DFF_MARK(20);
                        if (lastl) stallR_run<U_W1, 1, E, false>(sring, acc_f, ht, sqf_last, lane);
                        else stallR_run<U_W1, 1, E, false>(sring, acc_f, ht, sqf_next, lane);
                    } else if constexpr (SPW) {
                        if (lastl) stall_run<U_W1, FS / 32, E, false>(sring, acc_f, [=](int kb) { return ha + 32 * kb; }, sqf_last, lane);
                        else stall_run<U_W1, FS / 32, E, false>(sring, acc_f, [=](int kb) { return ha + 32 * kb; }, sqf_next, lane);
                    }
                    else tall_run<NTS % DR, NTS, E, -1>(ring, acc_f, [=](int kb) { return ha + 16 * kb; }, s_w2(lw), after, lane);
                    pf.tick(21); DFF_MARK(21);
                    // gelu'(h_pre) rows of this wave's hidden slice: LDS tile -> stash, 16 bytes per lane (rows beyond the real
                    // ones go to the dummy stash row); before the partial sums below reuse the tile
                    if (!gp_lds) {
                        const lfloat* const gp = hbuf + 16 * LF;
                        gfloat* const dst = sb + sl.h_pre + wave * FS;
#pragma unroll
                        for (int u = 0; u < (16 * (FS / 4) + 63) / 64; ++u) {
                            const int e = lane + 64 * u, row = e / (FS / 4), c4 = e - row * (FS / 4);
                            if (row < 16) *(gf32x4*)(dst + (row < rows ? row : RA) * F + 4 * c4) = *(const lf32x4*)(gp + row * LF + 4 * c4);
                        }
                    }
                }
                static_assert((2 * NTS) % DR == 0, "ring phase");
#pragma unroll
                for (int nt = 0; nt < E; ++nt) c_store_offs(mypart, mro, 16 * nt, acc_f[nt], lane);
            }
            __syncthreads();
            pf.tick(4); DFF_MARK(4);
            // ---- row stage C: ff = sum_w part + b2 ; gate2 ; next layer's LN1 or the energy head ----
            { DFF_ROW_CONSTS
            float psx[HC];
            if constexpr (PAIR) pair_rows(psx, rrow, sub, ract, tq_);
            if (ract) {
                const bool last = l == m.L - 1;
                float x[HC], res[HC], n2[HC];
                float ps[HC];
                if constexpr (PAIR) { for (int i = 0; i < HC; ++i) ps[i] = psx[i]; } else
                psum_all(ps, rs_o);
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + LPR * i, o = rs_o + LPR * i;
                    x[i] = ps[i] + ro[0][i];
                    res[i] = resbuf[o];
                    if constexpr (!KEEPROWS) st_ntg(sb + sl.ff + rrow * H + cl, x[i]);
                }
                if constexpr (KEEPROWS) keep_put(l, KF{}, x);
                const float g = ro_gate(x, res, 1);
                if constexpr (KEEPROWS) gate_put(l, 1, g);
#pragma unroll
                for (int i = 0; i < HC; ++i) n2[i] = x[i] * g + res[i] * (1.0f - g);
                if (last) {
                    float e = 0.f, wdv[HC];
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        const float wd = m.wdec[cl];
                        wdv[i] = wd;
                        e += n2[i] * wd;
                        if (!(KEEPROWS && m.conservative)) resbuf[rs_o + LPR * i] = m.conservative ? wd : n2[i];   // dn = d(sum e)/d nodes_L (or nodes_L for the force head)
                    }
                    if (a.energy_out) {
                        e = rsum(e);
                        if (sub == 0 && hf == 0) a.energy_out[(size_t)b0 * N + rrow] = e + m.bdec;
                    }
                    if constexpr (KEEPROWS) {
                        // stage D of this (last) layer right here: dn = the energy head's weights; gate-2 weights are in ro[1..3]
                        if (m.conservative) stage_D(l, wdv, std::integral_constant<int, 1>{}, rrow, sub);
                    }
                    // stage D operands of this (last) layer: attn_out, nodes_in from the stash, ff (kept),
                    // g1 (3), g2 (3 -- already in ro[1..3], move up)
                    if constexpr (KEEPROWS) {   // stage E operands: LN2 gamma -> ro[2], g1 (3) -> ro[3..5]
                        ro_load(2, lw.ln2_g, sub);
                    } else {
#pragma unroll
                        for (int i = 0; i < HC; ++i) { ro[6][i] = ro[1][i]; ro[7][i] = ro[2][i]; ro[8][i] = ro[3][i]; ro[2][i] = x[i]; }
                        ro_load(0, (const float*)(sb + sl.attn_out + rrow * H), sub);
                        ro_load(1, (const float*)(sbq + sl.nodes_in + rrow * H), sub);
                    }
                    ro_load3(3, lw.g1, sub);
                } else {
                    gfloat* const sbn = stash + (size_t)(l + 1) * sl.layer_stride;
                    float mean, rstd, nva[HC];
                    ln_stats_row(n2, mean, rstd);
                    if constexpr (KEEPROWS) { gate_put(l + 1, 2, mean); gate_put(l + 1, 3, rstd); }
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        resbuf[rs_o + LPR * i] = n2[i];
                        if constexpr (!KEEPROWS) st_ntg(sbn + sl.nodes_in + rrow * H + cl, n2[i]);
                        const float nv = (n2[i] - mean) * rstd * ro[4][i] + ro[5][i];
                        if constexpr (!NSP) a_put(i, nv);
                        n_put(i, nv);
                        nva[i] = nv;
                    }
                    if constexpr (KEEPROWS) { keep_put(l + 1, KN{}, n2); keep_put(l + 1, KL{}, nva); }
                    pre_B(m.layer[l + 1], sub);
                }
            } }
            if constexpr (DFF_XI_STAGE == 2) { if (l == 0) draw_xi(t_int, step); }
            // the stash written in the forward pass is re-read below by other lanes / waves
            if (l == m.L - 1) __threadfence_block();
            __syncthreads();
            pf.tick(5); DFF_MARK(5);
        }

        // =============================== backward ===============================
        auto hp_prefetch = [&](int l) {
            DFF_LANE_CONSTS
            const gfloat* const shp = stash + (size_t)l * sl.layer_stride + sl.h_pre + wave * FS + col;
#pragma unroll
            for (int d = 0; d < DR; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) hpn[d][r] = ld_ntg(shp + srow[r] * F + 16 * d);
        };
        if (!m.conservative) {
            // force head (graph_transformer.py:62-63,112-113): node_decoder is Linear(H, 3) and forces = its output,
            // no VJP.  nodes_L is in resbuf; the update stage takes dE/dx, so the negated forces go to dxs.
            DFF_ROW_CONSTS
            if (ract) {
                float f[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + LPR * i;
                    const float nv = resbuf[rs_o + LPR * i];
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) f[c3] += nv * m.wdec[c3 * H + cl];
                }
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3) {
                    f[c3] = rsum(f[c3]);
                    if (sub == 0) dxs[rrow * 4 + c3] = -(f[c3] + m.bdec3[c3]);
                }
            }
            __syncthreads();
        }
        if constexpr (HP_EARLY && !FOLD) { if (m.conservative) hp_prefetch(m.L - 1); }
        for (int l = m.conservative ? m.L - 1 : -1; l >= 0; --l) {
            const DffLayerDev& lw = m.layer[l];
            const gfloat* const sb = stash + (size_t)l * sl.layer_stride;
            const gfloat* const sbq = l == 0 ? l0e : sb;
            // ---- row stage D: gate2 backward: dn (resbuf) -> dff (abuf), dn1 partial (resbuf) ----
            // operands (prefetched): ro[0] attn_out, ro[1] nodes_in, ro[2] ff, ro[3..5] g1, ro[6..8] g2
            // (KEEPROWS: done inside the stage that made dn -- stage_D)
            if constexpr (!KEEPROWS) { DFF_ROW_CONSTS
            if (ract) {
                float n1[HC], dn[HC], ao[HC], ni[HC], fv[HC];
                rows_of(lsel(l), ao, ni, &fv, 2);
#pragma unroll
                for (int i = 0; i < HC; ++i) dn[i] = resbuf[rs_o + LPR * i];
                const float g1 = ro_gate(ao, ni, 3);
#pragma unroll
                for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + ni[i] * (1.0f - g1);
                const float g2 = ro_gate(fv, n1, 6);
                float dg = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) dg += dn[i] * (fv[i] - n1[i]);
                dg = rsum(dg);
                const float dz = dg * g2 * (1.0f - g2);
                float dffv[HC];
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + LPR * i;
                    dffv[i] = dn[i] * g2 + dz * (ro[6][i] + ro[8][i]);
                    resbuf[rs_o + LPR * i] = dn[i] * (1.0f - g2) + dz * (ro[7][i] - ro[8][i]);
                }
                a_store_row(rrow, sub, dffv, invD, false);
                // stage E operands: attn_out, nodes_in, g1 stay; LN2 gamma -> ro[2]
                ro_load(2, lw.ln2_g, sub);
            }
            __syncthreads();
            }
            pf.tick(6); DFF_MARK(6);
            // ---- FFN backward slice: dh = dff W2[:, slice] ; * gelu'(h_pre) ; partial df = dh_pre W1[slice, :] ----
            {
                DFF_LANE_CONSTS
                ro_touch();
                const WStream after = s_woxt(lw, wave);
                const SStream gn = ss_woxt(lw, wave);   // this layer's attention backward follows (G_ext GEMM: U_GX >= SDR units)
                const SSeq<U_W1, U_W2, SDR, KS, KS, KS, KS, E> sqb{ss_w2t(lw), ss_w1t(lw), gn, gn};
                f32x4 afr[E];
                if constexpr (!SPW) load_afrag<E>(afr, abuf, LH, lane);
                float hp[DR][4];
                const gfloat* const shp = sb + sl.h_pre + wave * FS + col;
                const int s0 = srow[0] * F, s1 = srow[1] * F, s2 = srow[2] * F, s3 = srow[3] * F;
                auto hp_load = [=](int t, float (&ax)[4]) {
                    ax[0] = ld_ntg(shp + s0 + 16 * t); ax[1] = ld_ntg(shp + s1 + 16 * t);
                    ax[2] = ld_ntg(shp + s2 + 16 * t); ax[3] = ld_ntg(shp + s3 + 16 * t);
                };
                if constexpr (FOLD) {
                    // the tile the forward FFN left in LDS (read now, in flight under the first weight units), or -- score mode,
                    // layers before the last three of a deep model -- the stash
                    if constexpr (MODE != DFF_MODE_SCORE) {
                        const lfloat* const gq = gp_tile(l, m.L) + col;
#pragma unroll
                        for (int d = 0; d < DR; ++d)
#pragma unroll
                            for (int r = 0; r < 4; ++r) hp[d][r] = gq[min(quad * 4 + r, LL::GPR - 1) * LL::GPS + 16 * d];
                    } else {
#pragma unroll
                        for (int d = 0; d < DR; ++d) hp_load(d, hp[d]);
                    }
                } else if constexpr (HP_EARLY) {
#pragma unroll
                    for (int d = 0; d < DR; ++d)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hp[d][r] = hpn[d][r];
                    if (l > 0) hp_prefetch(l - 1);      // the next backward layer's slice, a whole layer ahead
                } else {
#pragma unroll
                    for (int d = 0; d < DR; ++d) hp_load(d, hp[d]);
                }
                {
                    lfloat* const hb = hbuf + quad * 4 * LF + col;
                    auto w2t_epi = [=](int t, const f32x4& acc, const float (&ax)[4]) {
                            hb[16 * t] = acc[0] * ax[0]; hb[LF + 16 * t] = acc[1] * ax[1];
                            hb[2 * LF + 16 * t] = acc[2] * ax[2]; hb[3 * LF + 16 * t] = acc[3] * ax[3];
                        };
                    if constexpr (NSP) {
                        // (register chain, below)
                    } else if constexpr (SPW) {
                        u32x4 ah[KB32], am[KB32], al[KB32];
                        a_load(ah, am, al, lane);
                        swide_run<0, NTS, KB32, 4>(sring, hp, ah, am, al, sqb, lane, hp_load, w2t_epi);
                    } else
                    wide_run<0, NTS, E, 4>(ring, hp, afr, s_w2t(lw), s_w1t(lw), lane, hp_load, w2t_epi);
                }
                f32x4 acc_f[E];
#pragma unroll
                for (int nt = 0; nt < E; ++nt) acc_f[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                {
                    const lfloat* const ha = hbuf + col * LF + 4 * quad;
                    if constexpr (NSP) {
                        // W2^T -> x GELU' -> W1^T in registers (as the forward FFN): GELU'(h_pre) of (row = col, hidden columns 16 t + 4 quad ..)
                        // is one 16-byte read per tile of the LDS tile the forward left (sampling loops) or of the stash row (score mode)
                        f32x4 gd[2];
                        if constexpr (MODE != DFF_MODE_SCORE) {
                            const lfloat* const gq = gp_tile(l, m.L) + min(col, LL::GPR - 1) * LL::GPS + 4 * quad;
                            gd[0] = *(const lf32x4*)gq; gd[1] = *(const lf32x4*)(gq + 16);
                        } else {
                            const gfloat* const gq = sb + sl.h_pre + (size_t)(col < rows ? col : RA) * F + wave * FS + 4 * quad;
                            gd[0] = ld_ntg4(gq); gd[1] = ld_ntg4(gq + 16);
                        }
                        u32x4 ah[KB32], am[KB32], al[KB32];
                        a_load(ah, am, al, lane);
                        f32x4 dh[2], nob[2];
                        swideT_from<0, 0, NTS, KB32, false>(sring, dh, nob, nullptr, ah, am, sqb, lane);
                        dh[0] *= gd[0]; dh[1] *= gd[1];
                        stallR_run<U_W1, 1, E, false>(sring, acc_f, dh, sqb, lane);
                    } else
                    if constexpr (SPW) stall_run<U_W1, FS / 32, E, false>(sring, acc_f, [=](int kb) { return ha + 32 * kb; }, sqb, lane);
                    else tall_run<NTS % DR, NTS, E, -1>(ring, acc_f, [=](int kb) { return ha + 16 * kb; }, s_w1t(lw), after, lane);
                }
#pragma unroll
                for (int nt = 0; nt < E; ++nt) c_store_offs(mypart, mro, 16 * nt, acc_f[nt], lane);
            }
            __syncthreads();
            pf.tick(7); DFF_MARK(7);
            // first head of this layer's attention backward: start the stash read (hidden by row stage E)
            if constexpr (HDMA) {
                if (!(KEEP_LAST && l == m.L - 1 && l > 0) && !(KEEP2 && l == m.L - 2 && l > 0 && MODE != DFF_MODE_SCORE)) {
                    if (KEEP2 && l == 0 && m.L > 2 && MODE != DFF_MODE_SCORE && rows <= 10)   // (its P tile is in LDS: p0_copy)
                        dma_nop(sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, lane_id());
                    else head_dma<LL::DMA_N>(dmatab, Qx, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)wave * 256, lane_id());
                }
            }
            if constexpr (HPW == 2) {
                const int lane = lane_id();
                head_fetch(hr, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)wave * 256, RA, l > 0 || full0, lane,
                           GEN ? sb + sl.m12 + wave * 64 : nullptr);
            }
            // ---- row stage E: df = sum_w part ; LN2 backward ; gate1 backward -> dattn (abuf), dn_in partial (resbuf) ----
            // operands: ro[0] attn_out, ro[1] nodes_in, ro[2] ln2 gamma, ro[3..5] g1
            { DFF_ROW_CONSTS
            float psx[HC];
            if constexpr (PAIR) pair_rows(psx, rrow, sub, ract, tq_);
            if (ract) {
                float n1[HC], d1[HC], dyg[HC], xh[HC], ps[HC], ao[HC], ni[HC];
                float nxx[HC], nxg[HC], nxb[HC];   // hoistA: the next step's layer-0 node inputs, LayerNorm-1 gain and bias
                if (hoistA && l == 0) {
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        nxx[i] = ld_ntg(l0n + sl.nodes_in + rrow * H + cl);
                        nxg[i] = lw.ln1_g[cl]; nxb[i] = lw.ln1_b[cl];
                    }
                }
                const LSel sl_ = lsel(l);
                rows_of(sl_, ao, ni, nullptr, 0);
                if constexpr (PAIR) { for (int i = 0; i < HC; ++i) ps[i] = psx[i]; } else
                psum_all(ps, rs_o);
                if constexpr (SPW && DFF_F16_ON(FOLD)) {   // the FFN backward chain ran in this row's scaled units
#pragma unroll
                    for (int i = 0; i < HC; ++i) ps[i] *= invD;
                }
                pf.tick(22); DFF_MARK(22);
                float g1;
                if constexpr (KEEPROWS) g1 = gate_get(sl_, std::integral_constant<int, 0>{});
                else g1 = ro_gate(ao, ni, 3);
                pf.tick(23); DFF_MARK(23);
#pragma unroll
                for (int i = 0; i < HC; ++i) n1[i] = ao[i] * g1 + ni[i] * (1.0f - g1);
                float mean, rstd;
                if constexpr (KEEPROWS) { mean = gate_get(sl_, std::integral_constant<int, 4>{}); rstd = gate_get(sl_, std::integral_constant<int, 5>{}); }
                else ln_stats_row(n1, mean, rstd);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int o = rs_o + LPR * i;
                    xh[i] = (n1[i] - mean) * rstd;
                    dyg[i] = ps[i] * ro[2][i];
                    s1 += dyg[i];
                    s2 += dyg[i] * xh[i];
                }
                s1 = rsum(s1) * (1.0f / H);
                s2 = rsum(s2) * (1.0f / H);
                float dg = 0.f;
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    d1[i] = resbuf[rs_o + LPR * i] + rstd * (dyg[i] - s1 - xh[i] * s2);
                    dg += d1[i] * (ao[i] - ni[i]);
                }
                dg = rsum(dg);
                const float dz = dg * g1 * (1.0f - g1);
                float dav[HC], invE;
#pragma unroll
                for (int i = 0; i < HC; ++i) {
                    const int cl = sub + LPR * i;
                    dav[i] = d1[i] * g1 + dz * (ro[3][i] + ro[5][i]);
                    // (layer 0 of a cached-layer-0 model: nobody reads d(nodes_0); KEEPROWS Langevin leaves the NEXT step's residual
                    // stream there instead -- layer 0's node inputs -- so that step starts without a row stage A)
                    if (KEEPROWS && MODE == DFF_MODE_LANGEVIN && l == 0 && !full0) resbuf[rs_o + LPR * i] = ni[i];
                    else if (hoistA && l == 0) resbuf[rs_o + LPR * i] = nxx[i];
                    else resbuf[rs_o + LPR * i] = d1[i] * (1.0f - g1) + dz * (ro[4][i] - ro[5][i]);
                }
                a_store_row(rrow, sub, dav, invE, true);   // (scale and inverse -> rsc: the attention backward block's waves)
                (void)invE;
                if constexpr (KEEPROWS) {
                    // this layer's LayerNorm rows (the keys / values of its attention block) back into the shared buffer: kept in
                    // registers since the forward stage made them; the last layer's are still there
                    if (l < m.L - 1) {
                        float nva[HC];
                        keep_get(sl_, KL{}, nva);
#pragma unroll
                        for (int i = 0; i < HC; ++i) n_put(i, nva[i]);
                    }
                } else if constexpr (FOLD) {   // ... re-derived from the stashed node inputs, as the forward pass made them
                    float mean1, rstd1;
                    ln_stats_row(ni, mean1, rstd1);
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        n_put(i, (ni[i] - mean1) * rstd1 * lw.ln1_g[cl] + lw.ln1_b[cl]);
                    }
                }
                // stage F operands: nodes_in stays in ro[1]; LN1 gamma -> ro[2]  (KEEPROWS: + gate-2 weights of layer l - 1 for its stage D)
                if (l > 0 || full0) ro_load(2, lw.ln1_g, sub);
                if constexpr (KEEPROWS) {
                    if (l > 0) ro_load3(6, m.layer[l - 1].g2, sub);
                    else if ((MODE == DFF_MODE_LANGEVIN && !full0) || hoistA) pre_B(lw, sub);   // stage B operands of the next step's layer 0 (no stage A there)
                    if (hoistA && l == 0) {
                        // the next step's stage A, on registers (this step's kept layer-0 rows are spent: the LayerNorm rows went back
                        // to `nx` above, the node inputs into d1 / dg)
                        float mean0, rstd0, nva[HC];
                        ln_stats_row(nxx, mean0, rstd0);
#pragma unroll
                        for (int i = 0; i < HC; ++i) nva[i] = (nxx[i] - mean0) * rstd0 * nxg[i] + nxb[i];
                        keep_put(0, KN{}, nxx); keep_put(0, KL{}, nva);
                        gate_put(0, 2, mean0); gate_put(0, 3, rstd0);
                    }
                }
            } }
            __syncthreads();
            pf.tick(8); DFF_MARK(8);
            // ---- attention backward: wave w owns heads w and w+4 ----
            {
                DFF_LANE_CONSTS
                ro_touch();
                f32x4 afr[E];
                if constexpr (!SPW) load_afrag<E>(afr, abuf, LH, lane);   // dattn
                f32x4 acc_a[E];
#pragma unroll
                for (int nt = 0; nt < E; ++nt) acc_a[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // what follows this block: FFN backward of layer l-1, or (after layer 0) next step's first GEMM
                const bool more = (step + 1 < a.n_steps);
                const WStream after = l > 0 ? s_w2t(m.layer[l - 1])
                                            : (MODE == DFF_MODE_LANGEVIN ? s_wox(m.layer[0], wave) : s_qkv(m.layer[0], wave));
                (void)more;
                // what follows: FFN backward of layer l - 1 (W2^T, W1^T); after layer 0 the next step re-stages its own first units
                const DffLayerDev& lwp = m.layer[l > 0 ? l - 1 : 0];
                const SSeq<U_GX, U_QKVT, MW, KS, KT, KS, KS, E, 4 * KB32> sqa{ss_woxt(lw, wave), ss_qkvt(lw, wave), ss_w2t(lwp), ss_w1t(lwp)};   // tile 4 of 5: [r | g_D]
                // ... and after layer 0 (x-independent inputs: no back-projection) the NEXT STEP's first units -- [W_o;W_oc] of layer 0
                // when its q' comes from the table / the stash, QKV_ext otherwise -- requested by the G_ext GEMM's last refills,
                // a whole update stage before the step that needs them begins (they used to be requested at the END of the update
                // stage: with no row stage A in front of it -- Langevin, `skipA` -- the first attention block then opened with
                // an exposed L2 round trip)
                const SStream nx0 = EARLY ? (c0n ? ss_wox(m.layer[0], wave) : ss_qkv(m.layer[0], wave)) : ss_w2t(lwp);
                const SSeq<U_GX, 0, EARLY ? SDR : MW, KS, KS, KS, KS, E, 4 * KB32> sqa0{ss_woxt(lw, wave), ss_woxt(lw, wave), nx0, EARLY ? nx0 : ss_w1t(lwp)};
                u32x4 dah[KB32], dam[KB32], dal[KB32];   // SPW: dattn as bf16 pieces
                if constexpr (SPW) a_load(dah, dam, dal, lane);
                // this lane's share of the head's dE/dx terms (extension tiles of G_ext, dV_ext, dK_ext: rows quad * 4 + r,
                // column col): summed in registers and added to the wave's dxw ONCE per layer -- as three LDS
                // read-modify-writes per layer they were four dependent round trips each (the compiler cannot tell that
                // dxi[0..3] differ)
                f32x4 dxr = {0.f, 0.f, 0.f, 0.f};
                // G_ext = dattn W_o_ext[h]^T  (5 tiles: [G 64 | r 3 | 0]) -> G region ; dx_i -= r_i
                auto gext = [&](auto ph, int h, const WStream& wnext) {
                    float none[DR][1] = {};
                    lfloat* const gb = Gx + col;
                    const int l0 = lro[0], l1 = lro[1], l2 = lro[2], l3 = lro[3];
                    f32x4* const dxrp = &dxr;
                    wide_run<decltype(ph)::value, 5, E, 1>(ring, none, afr, s_woxt(lw, h), wnext, lane,
                        [=](int, float (&)[1]) {},
                        [=](int t, const f32x4& acc, const float (&)[1]) {
                            gb[l0 + 16 * t] = acc[0]; gb[l1 + 16 * t] = acc[1];
                            gb[l2 + 16 * t] = acc[2]; gb[l3 + 16 * t] = acc[3];
                            if (t == 4) *dxrp -= acc;
                        });
                };
                // the same on the split operands; phn = ring phase of the GEMM that follows (2: this head's QKV_ext^T back-projection)
                auto sgext = [&](const auto& sq) {
                    float none[2][1] = {};
                    lfloat* const gb = Gx + col;
                    const int l0 = lro[0], l1 = lro[1], l2 = lro[2], l3 = lro[3];
                    f32x4* const dxrp = &dxr;
                    // fp16 engine: dattn came in row-scaled (stage E): G_ext leaves the GEMM in true units again
                    f32x4 gi4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (F16E) gi4 = *(const lf32x4*)(rsc + 16 + 4 * quad);
                    swide_run<0, 5, KB32, 1>(sring, none, dah, dam, dal, sq, lane,
                        [=](int, float (&)[1]) {},
                        [=](int t, const f32x4& acc0, const float (&)[1]) {
                            const f32x4 acc = F16E ? acc0 * gi4 : acc0;
                            gb[l0 + 16 * t] = acc[0]; gb[l1 + 16 * t] = acc[1];
                            gb[l2 + 16 * t] = acc[2]; gb[l3 + 16 * t] = acc[3];
                            if (t == 4) *dxrp -= acc;
                        });
                };
                // dA = G_ext V_ext^T ; dS = scale * P (dA - sum_j P dA) -> dsb
                auto ds_math = [&]() {
                    if constexpr (!FOLD) write_xext(lane);   // (FOLD: the shared buffer's x columns are written once per step, by the centring)
                    const f32x4 dA = wv_dot_rows<XLD>(Gx, Vx, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = quad * 4 + r;
                        const float p = pb[i * DFF_PLD + col];
                        const float sm = row16_sum(p * dA[r]);
                        dsb[i * DFF_PLD + col] = 0.125f * p * (dA[r] - sm);
                    }
                };
                const int fa_off = col * XLD + 4 * quad;
                lfloat* const gx_ = Gx; lfloat* const kx_ = Kx; lfloat* const vx_ = Vx;
                auto qkvt_fa = [=](int kb) {
                    const lfloat* base = kb < 5 ? gx_ + 16 * kb : kb < 9 ? kx_ + 16 * (kb - 5) : vx_ + 16 * (kb - 9);
                    return base + fa_off;
                };
                auto qkvt_fa32 = [=](int kb) {   // 32-column blocks of [dQ | dK | dV] (the extension columns of dQ_ext go separately)
                    const lfloat* base = kb < 2 ? gx_ + 32 * kb : kb < 4 ? kx_ + 32 * (kb - 2) : vx_ + 32 * (kb - 4);
                    return base + fa_off;
                };
                const lfloat* const qkvt_xa = Gx + col * XLD + 64 + quad;
                auto dqkv = [&]() {
                    // dV_ext = P^T G_ext -> V region (ext columns: dx term sum_i a_ij r_i)
                    constexpr bool MMS = DFF_MM_SPLIT && H > 64;
                    wv_mm5<MMS, true, XLD>(pb, Gx, lane, ks4, [&](int nt, const f32x4& acc) {
                        if constexpr (FOLD) { if (nt < 4) acc_a[nt < 4 ? nt : 0] += acc; }   // v = n: dV IS a term of d(LayerNorm output)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (!FOLD) Vx[lro[r] + 16 * nt + col] = acc[r];
                            if (nt == 4) dxr[r] += GEN ? dx_ext(quad * 4 + r, col, acc[r]) : acc[r];
                        }
                    });
                    // dQ_ext = dS K_ext -> G region (ext columns: du, and ds in the GEN variants)
                    wv_mm5<MMS, false, XLD>(dsb, Kx, lane, ks4, [&](int nt, const f32x4& acc) {
                        f32x4 v = acc;
                        if constexpr (GEN) {
                            if (nt == 4) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = dq_ext(quad * 4 + r, col, acc[r]);
                            }
                        }
                        c_store_offs(Gx, lro, 16 * nt, v, lane);
                    });
                    // dK_ext = dS^T Q_ext -> K region (ext columns: dx term sum_i dS_ij u_i)
                    wv_mm5<MMS, true, XLD>(dsb, Qx, lane, ks4, [&](int nt, const f32x4& acc) {
                        if constexpr (FOLD) { if (nt < 4) acc_a[nt < 4 ? nt : 0] += acc; }   // k = n: so is dK
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (!FOLD) Kx[lro[r] + 16 * nt + col] = acc[r];
                            if (nt == 4) dxr[r] += GEN ? dx_ext(quad * 4 + r, col, acc[r]) : acc[r];
                        }
                    });
                };
                auto dx_only = [&]() {   // layer 0: node inputs do not depend on x
                    wv_mm<4, 5, true, XLD>(pb, Gx, lane, ks4, [&](int, const f32x4& acc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dxr[r] += GEN ? dx_ext(quad * 4 + r, col, acc[r]) : acc[r];
                    });
                    wv_mm<4, 5, true, XLD>(dsb, Qx, lane, ks4, [&](int, const f32x4& acc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dxr[r] += GEN ? dx_ext(quad * 4 + r, col, acc[r]) : acc[r];
                    });
                    if constexpr (GEN) {   // the distance term of the logits reaches x_i through Q_ext: -2 s_i sum_j dS_ij x_j
                        wv_mm<4, 5, false, XLD>(dsb, Kx, lane, ks4, [&](int, const f32x4& acc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) (void)dq_ext(quad * 4 + r, col, acc[r]);
                        });
                    }
                };
                // ---- REGCHAIN (FOLD kernel, fp16 engine; round 6): the backward chain in registers, transposed like the forward one.
                //   G_ext^T (swapped MFMA: lane (i = col, quad) holds row i, columns 16 t + 4 quad ..) -> its pieces are the B operand of
                //   dA^T = n G^T on the fp16 pipe (A = the LayerNorm rows' pieces, region nsp) -> softmax backward of row i in the
                //   lane group (i, quad 0..3) -> dS^T is, as it stands, the B operand of dQ_ext^T = K_ext^T dS^T, whose tiles are the A
                //   fragments of the back-projection.  dV and dK contract over i and stay on the LDS-operand path (P, dS, true-unit G
                //   rows are stored for them, 16 bytes per store).
                f32x4 dxT = {0.f, 0.f, 0.f, 0.f};   // dE/dx terms held as (row i = col, components 0..3) by the quad-0 lanes
                const int lroT = min(col, RLA - 1) * XLD;
                auto quads_sum = [](float v) {   // all-reduce over lanes l, l ^ 16, l ^ 32, l ^ 48
                    const auto a2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    v = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
                    const auto b2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    return __uint_as_float(b2[0]) + __uint_as_float(b2[1]);
                };
                // G_ext^T -> true-unit rows in the G region (all tiles, or the extension tile only), dxT -= r_i, returns dS^T
                auto gds_T = [&](const auto& sq, bool all_tiles) -> f32x4 {
                    f32x4 gt[5], nob[2];
                    swideT_from<0, 0, 5, KB32, false>(sring, gt, nob, nullptr, dah, dam, sq, lane);
                    const float gi = rsc[16 + min(col, RLA - 1)];   // dattn came in row-scaled (stage E): 1 / scale of row i
                    if (all_tiles) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) *(lf32x4*)(Gx + lroT + 16 * t + 4 * quad) = gt[t] * gi;
                    }
                    const f32x4 gx4 = gt[4] * gi;
                    *(lf32x4*)(Gx + lroT + 64 + 4 * quad) = gx4;
                    if (quad == 0) dxT -= gx4;
                    pf.tick(15); DFF_MARK(15);
                    // dA^T[j][i] = n_j . G_i (64 columns, fp16 pieces; in units of row i's scale) + x_j . r_i (fp32 k-step)
                    u32x4 nh[KB32], nl[KB32], gh[2], gl[2];
                    an_load(nh, nl, lane);
                    split8h(gt[0], gt[1], gh[0], gl[0]);
                    split8h(gt[2], gt[3], gh[1], gl[1]);
                    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        cs = mfma_f16(nl[kb], gh[kb], cs);
                        cs = mfma_f16(nh[kb], gl[kb], cs);
                        cb = mfma_f16(nh[kb], gh[kb], cb);
                    }
                    const f32x4 dA = wv_dot_ext<XLD>(Kx, Gx, lane, (cb + cs * DFF_F16_LINV) * gi);
                    const f32x4 p4 = *(const lf32x4*)(pb + col * DFF_PLD + 4 * quad);   // P[i = col][j = 4 quad ..]
                    const float sm = quads_sum((p4[0] * dA[0] + p4[1] * dA[1]) + (p4[2] * dA[2] + p4[3] * dA[3]));
                    f32x4 dS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dS[r] = 0.125f * p4[r] * (dA[r] - sm);
                    *(lf32x4*)(dsb + col * DFF_PLD + 4 * quad) = dS;   // [i][j]: dK's operand
                    pf.tick(16); DFF_MARK(16);
                    return dS;
                };
                if constexpr (NSP) {
                    if constexpr (HDMA) {
                        head_dma_wait();   // (requested before row stage E; nothing for the last layer)
                        if constexpr (KEEP2) {
                            if (l == m.L - 2 && l > 0 && MODE != DFF_MODE_SCORE) keep2_copy(Qsave, Qx, false, lane, pcij);
                            if (l == 0 && m.L > 2 && MODE != DFF_MODE_SCORE && rows <= 10) p0_copy(false, pcij);
                        }
                    }
                    pf.tick(8); DFF_MARK(8);
                    if (l > 0) {
                        const f32x4 dS = gds_T(sqa, true);
                        // dV_ext = P^T G_ext: v = n, so dV is a term of d(LayerNorm output); extension columns -> dx_j
                        wv_mm5<false, true, XLD>(pb, Gx, lane, ks4, [&](int nt, const f32x4& acc) {
                            if (nt < 4) acc_a[nt < 4 ? nt : 0] += acc;
                            else dxr += acc;
                        });
                        // dQ_ext^T = K_ext^T dS^T, as O^T above: the 64 regular columns on the fp16 pipe -- dS^T of lane (i, kg) scaled by row i's
                        // dattn scale (a power of two; the back-projection wants its A rows in exactly those units) and split --, the
                        // extension tile (du) on the fp32 pipe in true units
                        f32x4 dq[5];
                        {
                            typedef const volatile lfloat* vlp;
                            float bvx[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) bvx[r] = *(vlp)(Kx + (4 * quad + r) * XLD + 64 + col);
                            f16x4 kh[4], kl[4];
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) nt_load(nt, lane, kh[nt], kl[nt]);
                            const float si_ = rsc[min(col, RLA - 1)];
                            f16x4 sh, sl2;
                            {
                                unsigned h0, l0, h1, l1;
                                split2h(dS[0] * si_, dS[1] * si_, h0, l0); split2h(dS[2] * si_, dS[3] * si_, h1, l1);
                                sh = __builtin_bit_cast(f16x4, (u32x2){h0, h1}); sl2 = __builtin_bit_cast(f16x4, (u32x2){l0, l1});
                            }
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) {
                                f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
                                cs = __builtin_amdgcn_mfma_f32_16x16x16f16(kl[nt], sh, cs, 0, 0, 0);
                                cs = __builtin_amdgcn_mfma_f32_16x16x16f16(kh[nt], sl2, cs, 0, 0, 0);
                                cb = __builtin_amdgcn_mfma_f32_16x16x16f16(kh[nt], sh, cb, 0, 0, 0);
                                dq[nt] = cb + cs * DFF_F16_LINV;   // (rows in the units of their dattn scale)
                            }
                            dq[4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int r = 0; r < 4; ++r) dq[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(bvx[r], dS[r], dq[4], 0, 0, 0);
                        }
                        *(lf32x4*)(Gx + lroT + 64 + 4 * quad) = dq[4];   // du: the back-projection's fp32 k-step reads it (r_i is spent)
                        // dK_ext = dS^T Q_ext: k = n, likewise; extension columns -> dx_j
                        wv_mm5<false, true, XLD>(dsb, Qx, lane, ks4, [&](int nt, const f32x4& acc) {
                            if (nt < 4) acc_a[nt < 4 ? nt : 0] += acc;
                            else dxr += acc;
                        });
                        pf.tick(17); DFF_MARK(17);
                        const f32x4 (&dq4)[4] = *reinterpret_cast<const f32x4 (*)[4]>(&dq[0]);
                        stallR_run<U_GX, NKT, E, true>(sring, acc_a, dq4, sqa, lane, Gx + col * XLD + 64 + quad, qkvt_ext(lw, wave, lane),
                                                       DFF_HEADS * 13 * 256, rsc, true);
                        pf.tick(18); DFF_MARK(18);
#pragma unroll
                        for (int nt = 0; nt < E; ++nt) c_store_offs(mypart, mro, 16 * nt, acc_a[nt], lane);
                    } else {
                        (void)gds_T(sqa0, false);
                        // layer 0's node inputs do not depend on x: only the extension tiles of dV_ext and dK_ext (the dx_j terms)
                        wv_mm<4, 5, true, XLD>(pb, Gx, lane, ks4, [&](int, const f32x4& acc) { dxr += acc; });
                        wv_mm<4, 5, true, XLD>(dsb, Qx, lane, ks4, [&](int, const f32x4& acc) { dxr += acc; });
                        if constexpr (EARLY) {
                            early_done = true;
                            if constexpr (HDMA) {
                                if (nxt && c0n) dma_nop(l0n + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, lane);
                            }
                        }
                    }
                }
                const int h1 = wave + 4;
                (void)h1;
                // GEN: [m1 | m2] of the head whose buffers are live (hr.m is overwritten by the next head's prefetch),
                // and the x_i fix-ups of Q_ext (after every commit that brings q_ext) and G_ext (after every gext)
                float m_cur = 0.f;
                auto m12p = [&](int h) -> const gfloat* { return GEN ? sb + sl.m12 + h * 64 : nullptr; };
                auto committed = [&]() { if constexpr (GEN) { m_cur = hr.m; fix_q(lane); } };
                auto gfix = [&]() { if constexpr (GEN) fix_g(lane, m_cur); };
                if constexpr (NSP) {
                    // (done above)
                } else
                if (l > 0 || full0) {
                    if constexpr (HPW == 2) {
                        head_commit(hr, Qx, Kx, Vx, pb, true, true, lane, RLA, XLD);
                        committed();
                        head_fetch(hr, sbq + sl.qkv + (size_t)h1 * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)h1 * 256, RA, true, lane, m12p(h1));
                        pf.tick(8); DFF_MARK(8);
                        gext(std::integral_constant<int, 0>{}, wave, s_qkvt(lw, wave));
                        gfix();
                        pf.tick(15); DFF_MARK(15);
                        ds_math();
                        pf.tick(16); DFF_MARK(16);
                        dqkv();
                        pf.tick(17); DFF_MARK(17);
                        tall_run<1, 13, E, 4>(ring, acc_a, qkvt_fa, s_qkvt(lw, wave), s_woxt(lw, h1), lane);
                        pf.tick(18); DFF_MARK(18);
                        head_commit(hr, Qx, Kx, Vx, pb, true, true, lane, RLA, XLD);
                        committed();
                        gext(std::integral_constant<int, 2>{}, h1, s_qkvt(lw, h1));
                        gfix();
                        pf.tick(15); DFF_MARK(15);
                        ds_math();
                        pf.tick(16); DFF_MARK(16);
                        dqkv();
                        pf.tick(17); DFF_MARK(17);
                        tall_run<3, 13, E, 4>(ring, acc_a, qkvt_fa, s_qkvt(lw, h1), after, lane);   // ends at phase 0
                        pf.tick(18); DFF_MARK(18);
                    } else {
                        if constexpr (HDMA) {
                            head_dma_wait();   // (requested before row stage E; nothing for the last layer)
                            if constexpr (KEEP2) { if (l == m.L - 2 && l > 0 && MODE != DFF_MODE_SCORE) keep2_copy(Qsave, Qx, false, lane, pcij); }
                        }
                        else if (!(KEEP_LAST && l == m.L - 1)) {   // the last layer's q_ext | k | v | P are still in the head buffers
                            head_fetch(hr, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)wave * 256, RA, true, lane, m12p(wave));
                            head_commit(hr, Qx, Kx, Vx, pb, true, true, lane, RLA, XLD);
                        }
                        committed();
                        pf.tick(8); DFF_MARK(8);
                        if constexpr (SPW) sgext(sqa);
                        else gext(std::integral_constant<int, 0>{}, wave, s_qkvt(lw, wave));
                        gfix();
                        pf.tick(15); DFF_MARK(15);
                        ds_math();
                        pf.tick(16); DFF_MARK(16);
                        dqkv();
                        pf.tick(17); DFF_MARK(17);
                        if constexpr (SPW) stall_run<U_GX, NKT, E, true>(sring, acc_a, qkvt_fa32, sqa, lane, qkvt_xa, qkvt_ext(lw, wave, lane), DFF_HEADS * 13 * 256,
                                                                         F16E ? rsc : nullptr, !FOLD);   // (dQ' rows re-scaled by their dattn row's scale; unfolded: [dQ | dK | dV] by one common scale)
                        else tall_run<5 % DR, 13, E, 4>(ring, acc_a, qkvt_fa, s_qkvt(lw, wave), after, lane);   // 18 entries: phase 0
                        pf.tick(18); DFF_MARK(18);
                    }
#pragma unroll
                    for (int nt = 0; nt < E; ++nt) c_store_offs(mypart, mro, 16 * nt, acc_a[nt], lane);
                } else if constexpr (HPW == 2) {
                    // layer 0 needs q_ext (for the dS^T u term) but not k
                    head_commit(hr, Qx, Kx, Vx, pb, false, true, lane, RLA, XLD);
                    {   // q_ext of head `wave`
                        const gfloat* sq = sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW;
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const int it = lane + 64 * u, row = it / 20, c4 = it % 20;
                            hr.q[u] = ld_ntg4(sq + min(row, RA) * DFF_QKVW + 4 * c4);
                        }
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const int it = lane + 64 * u, row = it / 20, c4 = it % 20;
                            if (row < RLA) *(lf32x4*)(Qx + row * XLD + 4 * c4) = hr.q[u];
                        }
                    }
                    committed();
                    head_fetch(hr, sbq + sl.qkv + (size_t)h1 * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)h1 * 256, RA, true, lane, m12p(h1));
                    gext(std::integral_constant<int, 0>{}, wave, s_woxt(lw, h1));
                    gfix();
                    ds_math();
                    dx_only();
                    head_commit(hr, Qx, Kx, Vx, pb, true, true, lane, RLA, XLD);
                    committed();
                    gext(std::integral_constant<int, 1>{}, h1, after);
                    gfix();
                    ds_math();
                    dx_only();
                    // the next step (if any) re-stages its own first entries at phase 0
                } else {
                    if constexpr (HDMA) {
                        head_dma_wait();
                        if constexpr (KEEP2) { if (m.L > 2 && MODE != DFF_MODE_SCORE && rows <= 10) p0_copy(false, pcij); }
                    } else {
                        head_fetch(hr, sbq + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, sb + sl.P + (size_t)wave * 256, RA, true, lane, m12p(wave));
                        head_commit(hr, Qx, Kx, Vx, pb, true, true, lane, RLA, XLD);
                    }
                    committed();
                    if constexpr (SPW) sgext(sqa0);
                    else gext(std::integral_constant<int, 0>{}, wave, after);
                    gfix();
                    ds_math();
                    dx_only();
                    if constexpr (EARLY) {
                        // the ring holds the next step's first units; its layer-0 q' rows follow by LDS-DMA into this wave's own
                        // (now dead) Q region
                        early_done = true;
                        if constexpr (HDMA) {
                            if (nxt && c0n) dma_nop(l0n + sl.qkv + (size_t)wave * (RA + 1) * DFF_QKVW, lane);
                        }
                    }
                }
                if (!GEN && l == m.L - 1) {
                    // the first layer of the backward sweep STORES (the GEN variants also add to dxw from inside the block: they
                    // start from the zeros the update stage leaves)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dxw[dxi[r]] = dxr[r];
                } else {
                    float t4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) t4[r] = dxw[dxi[r]];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dxw[dxi[r]] = t4[r] + dxr[r];
                }
                if constexpr (NSP) {
                    // ... and what the quad-0 lanes hold by row: (row col, components 0..3) is one 16-byte slot of the same array (LDS
                    // operations of a wave complete in order: this read-modify-write sees the stores above)
                    if (quad == 0 && col < rows) {
                        lf32x4* const d4 = (lf32x4*)(dxw + col * 4);
                        const f32x4 t = *(volatile lf32x4*)d4;
                        *(volatile lf32x4*)d4 = t + dxT;
                    }
                }
            }
            __syncthreads();
            pf.tick(9); DFF_MARK(9);
            // ---- row stage F: dn = dn_in partial + LN1 backward(sum_w part)  (l > 0) ----
            // operands: ro[1] nodes_in, ro[2] ln1 gamma
            if (l > 0 || full0) {
                DFF_ROW_CONSTS
                float psx[HC];
                if constexpr (PAIR) pair_rows(psx, rrow, sub, ract, tq_);
                if (ract) {
                    float dyg[HC], xh[HC], ps[HC], ao[HC], ni[HC];
                    const LSel sl_ = lsel(l);
                    rows_of(sl_, ao, ni, nullptr, 0);
                    if constexpr (PAIR) { for (int i = 0; i < HC; ++i) ps[i] = psx[i]; } else
                    psum_all(ps, rs_o);
                    float mean, rstd;
                    if constexpr (KEEPROWS) { mean = gate_get(sl_, std::integral_constant<int, 2>{}); rstd = gate_get(sl_, std::integral_constant<int, 3>{}); }
                    else ln_stats_row(ni, mean, rstd);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int o = rs_o + LPR * i;
                        xh[i] = (ni[i] - mean) * rstd;
                        dyg[i] = ps[i] * ro[2][i];
                        s1 += dyg[i];
                        s2 += dyg[i] * xh[i];
                    }
                    s1 = rsum(s1) * (1.0f / H);
                    s2 = rsum(s2) * (1.0f / H);
                    if constexpr (KEEPROWS) {
                        // dn of layer l - 1 stays in registers and goes straight through that layer's stage D (gate-2 weights in
                        // ro[6..8], prefetched by stage E); then its stage E operands: LN2 gamma -> ro[2], g1 (3) -> ro[3..5]
                        float dnv[HC];
#pragma unroll
                        for (int i = 0; i < HC; ++i) dnv[i] = resbuf[rs_o + LPR * i] + rstd * (dyg[i] - s1 - xh[i] * s2);
                        stage_D(l - 1, dnv, std::integral_constant<int, 6>{}, rrow, sub);
                        ro_load(2, m.layer[l - 1].ln2_g, sub);
                        ro_load3(3, m.layer[l - 1].g1, sub);
                    } else {
#pragma unroll
                    for (int i = 0; i < HC; ++i) resbuf[rs_o + LPR * i] += rstd * (dyg[i] - s1 - xh[i] * s2);
                    if (l > 0) {   // stage D operands of layer l-1
                        const DffLayerDev& lp = m.layer[l - 1];
                        const gfloat* const sp = stash + (size_t)(l - 1) * sl.layer_stride;
                        ro_load(0, (const float*)(sp + sl.attn_out + rrow * H), sub);
                        ro_load(1, (const float*)((l == 1 ? l0e : sp) + sl.nodes_in + rrow * H), sub);
                        ro_load(2, (const float*)(sp + sl.ff + rrow * H), sub);
                        ro_load3(3, lp.g1, sub); ro_load3(6, lp.g2, sub);
                    }
                    }
                }
                __syncthreads();
            }
            pf.tick(10); DFF_MARK(10);
        }
        if constexpr (KEEPROWS && MODE == DFF_MODE_DDPM) {
            if (hoistA) {   // the next step's LayerNorm rows of layer 0 (+ their fp16 pieces): `nx` is free now that layer 0's attention backward is through
                DFF_ROW_CONSTS
                if (ract) {
                    float nva[HC];
                    keep_get(LSel{true, false}, KL{}, nva);
#pragma unroll
                    for (int i = 0; i < HC; ++i) n_put(i, nva[i]);
                }
            }
        }
        { const int tq = tid_id();   // (opaque: the per-lane addresses below are re-derived every step, not hoisted and spilled)
        // dxs = sum of the waves' partial x-gradients (the force head wrote dxs itself); supplied / not yet drawn noise -> xib
        if (tq < 64 && m.conservative) {
            const lfloat* d0 = sm + LL::dxw;
            float pv[NWR];
#pragma unroll
            for (int w = 0; w < NWR; ++w) pv[w] = d0[w * 128 + tq];
            __builtin_amdgcn_sched_barrier(0);   // all reads in flight before the first add
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWR; ++w) t += pv[w];
            if constexpr (PAIR) t = pair_dx(t, tq);   // + the partner's four heads (wave 0 only: no workgroup barrier)
            dxs[tq] = t;
        }
        if constexpr (PAIR) { if (m.conservative) ++xseq; }
        if (MODE != DFF_MODE_SCORE && !xi_pre && tq < rows * 4 && (tq & 3) < 3) {
            const int row = tq >> 2, cc = tq & 3, g = row / N, i = row - g * N;
            const size_t item = (size_t)b0 + g;
            if (a.noise) xib[tq] = a.noise[(((size_t)step * a.B + item) * N + i) * 3 + cc];
            else xib[tq] = philox_normal(seed_now(), a.item_offset + item, MODE == DFF_MODE_DDPM ? (uint64_t)t_int : a.step_offset + step, i, cc);
        }
        // Everything the update reads was written by wave 0 itself just above (dxs, supplied noise: LDS operations of one wave
        // complete in order) or before the barrier that ended the backward sweep (the waves' partials, the pre-drawn normals): no
        // workgroup barrier here.  The GEN variants zero their partials for the next step, which must wait for wave 0's reads.
        if constexpr (GEN) {
            __syncthreads();
            { const int ln = lane_id(); dxw[ln] = 0.f; dxw[64 + ln] = 0.f; }   // (this wave's dx partials of the NEXT step start from zero)
        } else __builtin_amdgcn_wave_barrier();
        if constexpr (GEN) {
            if (full0 && m.conservative) {   // absolute coordinates: dE/dx_i += d(nodes_0)_i . W_node[:, x columns]
                DFF_ROW_CONSTS
                if (ract) {
                    float f3[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < HC; ++i) {
                        const int cl = sub + LPR * i;
                        const float dn = resbuf[rs_o + LPR * i];
#pragma unroll
                        for (int c3 = 0; c3 < 3; ++c3) f3[c3] += dn * m.WnT[(N + c3) * H + cl];
                    }
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) {
                        f3[c3] = rsum(f3[c3]);
                        if (sub == 0) dxs[rrow * 4 + c3] += f3[c3];
                    }
                }
                __syncthreads();
            }
        }

        // =============================== update (as dff_fused_kernel) ===============================
        if (MODE == DFF_MODE_SCORE) {
            if (tq < rows * 4 && (tq & 3) < 3)
                if (hf == 0) a.force_out[((size_t)b0 * N + (tq >> 2)) * 3 + (tq & 3)] = -dxs[tq];
        } else if (MODE == DFF_MODE_LANGEVIN) {
            const bool save = ((step + 1) % a.save_interval) == 0;
            const int fi = (step + 1) / a.save_interval - 1;
            if (tq < rows * 4 && (tq & 3) < 3) {
                const int row = tq >> 2, cc = tq & 3;
                const int g = row / N, i = row - g * N;
                const size_t item = (size_t)b0 + g;
                const float xi = xib[tq];
                const float f = dxs[tq] * a.force_scale;
                const float x = xcb[tq];        // the centred x_old (langevin.py:75-92 centres before the force call)
                float xn, vn = 0.f;
                if (a.overdamped) {
                    xn = x + f * a.dtau + a.brown_sigma * xi;
                } else {
                    vn = vst[tq] + (a.dt * f) / mass_i;
                    xn = x + (vn * a.dt) / 2.0f;
                    const float nz = nsig_i * xi;
                    vn = vn * a.vscale;
                    vn = vn + a.noisescale * nz;
                    xn = xn + (vn * a.dt) / 2.0f;
                }
                xst[tq] = xn;
                vst[tq] = vn;
                if (save && a.frames && hf == 0) a.frames[(((size_t)fi * a.B + item) * N + i) * 3 + cc] = xn;   // (PAIR: both blocks hold the same x)
            }
            if (step + 1 < a.n_steps) { __builtin_amdgcn_wave_barrier(); centre(); }   // (wave 0 reads back what it just wrote)
            if (save && a.ke && !a.overdamped) {
                __syncthreads();
                if (tq < gcnt) {
                    float ke = 0.f;
                    for (int i = 0; i < N; ++i) {
                        const lfloat* vp = vst + (tq * N + i) * 4;
                        ke += a.mass[i] * (vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]);
                    }
                    if (hf == 0) a.ke[(size_t)fi * a.B + b0 + tq] = 0.5f * ke;
                }
            }
        } else {
            // Reverse DDPM step (ddpm.py:195-232) + clamp + centring (:248-251).  Four per-protein means sit between the
            // element-wise stages; every (bead, component) thread -- all in wave 0, so its loads of the old xst precede every
            // store of the new one -- runs the whole chain for its protein's column in registers and takes the means itself
            // (summed in bead order as bead_mean() does): same arithmetic per element as the stage-by-stage form, no barriers.
            if (tq < rows * 4 && (tq & 3) < 3) {
                const int cc = tq & 3, pb0 = ((tq >> 2) / N) * N;
                const float sr = sch_sr, srm1 = sch_srm1;
                const float c1 = sch_c1, c2 = sch_c2;
                const float sig = ((t_int == 0) ? 0.f : 1.f) * expf(0.5f * sch_lv);
                const float invn = (float)N;
                float ev[16], zv[16], xv[16];
                const int o0 = pb0 * 4 + cc;
#pragma unroll
                for (int i = 0; i < 16; ++i) {   // (unclamped: rows past the protein's are never summed, see colsum16)
                    ev[i] = -dxs[o0 + 4 * i]; zv[i] = xib[o0 + 4 * i]; xv[i] = xst[o0 + 4 * i];
                }
                float e_own = -dxs[tq], z_own = xib[tq];
                const float x_own = xst[tq];
                // (four dependent means over 80 live values: here the early-exit form spills; the conditions are taken per lane
                // against N in a VECTOR register instead -- one v_cmp each, still no kernel-lifetime lane masks)
                int nv_ = N;
                asm volatile("" : "+v"(nv_));
                auto colmean = [&](const float (&v)[16]) {
                    float sacc = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc += i < nv_ ? v[i] : 0.f;
                    return sacc / invn;
                };
                const float me = colmean(ev), mz = colmean(zv);
                float x0v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { ev[i] -= me; zv[i] -= mz; x0v[i] = sr * xv[i] - srm1 * ev[i]; }
                e_own -= me; z_own -= mz;
                float x0_own = sr * x_own - srm1 * e_own;
                const float m0 = colmean(x0v);
                float xnv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float mu = c1 * (x0v[i] - m0) + c2 * xv[i];
                    xnv[i] = fminf(fmaxf(mu + sig * zv[i], -1000.f), 1000.f);
                }
                x0_own -= m0;
                const float mu_own = c1 * x0_own + c2 * x_own;
                float xn = mu_own + sig * z_own;
                if (xn > 1000.f || xn < -1000.f) {
                    if (a.clamp_flag) atomicOr(a.clamp_flag, 1);
                    xn = fminf(fmaxf(xn, -1000.f), 1000.f);
                }
                xst[tq] = xn - colmean(xnv);
            }
            if (step + 1 < a.n_steps) { __builtin_amdgcn_wave_barrier(); centre(); }
        }
        // The next step's first weight units and layer-0 head rows: requested here, where seven waves have nothing to do but
        // wait for wave 0's integrator update (wave 0 requests its own once it is through); the ring and the Q region are
        // dead since the last attention block.
        if constexpr (SPW) {
            if (nxt && !early_done) step_prefetch(c0n, l0n);   // (models whose layer 0 has a VJP, force-head models: the backward sweep did not)
        }
        __syncthreads();
        }
        // end of a reverse chain: assert_center_zero(mol) (models/ddpm.py:252) on the device -> bit 1 of the flag word
        if (MODE == DFF_MODE_DDPM && step == a.n_steps - 1 && a.clamp_flag) {
            bead_mean(c, (float*)xst, (float*)cm);
            __syncthreads();
            if (tid < gcnt * 4 && (tid & 3) < 3 && !(fabsf(cm[tid]) < 1e-3f)) atomicOr(a.clamp_flag, 2);
        }
        pf.tick(11); DFF_MARK(11);
    }
    if (pf.on)
        for (int i = 0; i < DFF_NPROF; ++i) a.prof[i] = pf.acc[i];
    if (MODE != DFF_MODE_SCORE && tid < rows * 4 && (tid & 3) < 3 && hf == 0) {
        const size_t gi = ((size_t)b0 * N + (tid >> 2)) * 3 + (tid & 3);
        a.x_io[gi] = xst[tid];
        if (MODE == DFF_MODE_LANGEVIN && !a.overdamped) a.v_io[gi] = vst[tid];
    }
}

// kernel lookup for the host dispatcher (dff_host.hip); taking the address instantiates the variant.  One translation unit
// per sampler mode: this one exports dff_small_pick_m<DFF_SMALL_MODE>.
#ifndef DFF_SMALL_MODE
#error "compile dff_small.hip with -DDFF_SMALL_MODE=0|1|2 (DFF_MODE_SCORE / LANGEVIN / DDPM): build.sh builds all three"
#endif
#if DFF_SMALL_MODE == 0
int dff_small_f16_level() { return DFF_F16; }
#endif
#define DFF_CAT2(a, b) a##b
#define DFF_CAT(a, b) DFF_CAT2(a, b)
bool DFF_CAT(dff_small_pick_m, DFF_SMALL_MODE)(int H, int NW, bool gen, bool spw, const void** fn, unsigned* lds_floats, const char** name, bool fold,
                                               bool pair) {
    constexpr int MD = DFF_SMALL_MODE;
    if (pair) {   // two workgroups per protein: the FOLD kernel on the fp16 engine, sampling loops only
#if DFF_F16 >= 1 && DFF_SMALL_MODE != 0
        if (H == 64 && NW == 8 && spw && fold && !gen) {
            *fn = (const void*)&dff_small_kernel<64, 8, false, true, true, MD, true>;
            *lds_floats = SmallLds<64, 8, true, 4>::total;
            *name = "dff_small_kernel<64,8,split_f16,fold_kv,pair>";
            return true;
        }
#endif
        return false;
    }
    if (H == 64 && NW == 8 && spw && fold && !gen) {
        *fn = (const void*)&dff_small_kernel<64, 8, false, true, true, MD>;
        *lds_floats = SmallLds<64, 8, true>::total;
        *name = DFF_F16 ? "dff_small_kernel<64,8,split_f16,fold_kv>" : "dff_small_kernel<64,8,split_bf16,fold_kv>";
        return true;
    }
    if (H == 64 && NW == 8 && spw) {
        *fn = gen ? (const void*)&dff_small_kernel<64, 8, true, true, false, MD> : (const void*)&dff_small_kernel<64, 8, false, true, false, MD>;
        *lds_floats = SmallLds<64, 8>::total;
        *name = DFF_F16 >= 2 ? (gen ? "dff_small_kernel<64,8,gen,split_f16>" : "dff_small_kernel<64,8,split_f16>")
                             : (gen ? "dff_small_kernel<64,8,gen,split_bf16>" : "dff_small_kernel<64,8,split_bf16>");
        return true;
    }
#define SMALL_CASE(H_, NW_)                                                                                          \
    if (H == H_ && NW == NW_ && !spw) {                                                                              \
        *fn = gen ? (const void*)&dff_small_kernel<H_, NW_, true, false, false, MD> : (const void*)&dff_small_kernel<H_, NW_, false, false, false, MD>;  \
        *lds_floats = SmallLds<H_, NW_>::total;                                                                      \
        *name = gen ? "dff_small_kernel<" #H_ "," #NW_ ",gen>" : "dff_small_kernel<" #H_ "," #NW_ ">";               \
        return true;                                                                                                 \
    }
    SMALL_CASE(64, 8)
#ifndef DFF_FAST_BUILD
    SMALL_CASE(64, 4)
    SMALL_CASE(96, 4)
    SMALL_CASE(128, 4)
    SMALL_CASE(96, 8)
#endif
#undef SMALL_CASE
    return false;
}
