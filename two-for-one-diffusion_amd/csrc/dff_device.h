// dff_device.h -- device-side helpers, layouts and types shared by the kernel translation units
// (dff_kernels.hip: the <= 64-row kernel, dff_small.hip: the <= 16-row kernel).  Not part of the ABI.
#pragma once
#include "dff_internal.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEVI __device__ __forceinline__

#define DFF_QKVW 208    // stash row of one head: [q_ext 80 | k 64 | v 64]
// Row padding (bf16 elements) of every split A operand in LDS.  A row stride of 8 dwords mod 64 makes the ds_read_b128
// fragment reads conflict-free: the 16 lanes of a read group are rows {0-3, 12-15} of one 8-element k-group and rows
// {4-11} of the next, i.e. bank slots (2 row + kg) mod 16 = all even | all odd.  (8 elements = 4 dwords mod 64 put row 11
// of k-group 1 on row 12 of k-group 0: 41 % conflict cycles in profiles/r02/villin.)
#define DFF_SPAD 16

// Explicitly address-space-typed pointers: the hot lambdas capture pointers by reference and some
// closures end up in memory, where a plain `float*` loses its provenance and every access turns
// into a FLAT op (which waits on BOTH counters and drains the weight ring).  Typed pointers keep
// ds_* / global_* no matter how they travel.
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(1))) float gfloat;
typedef f32x4 __attribute__((address_space(3))) lf32x4;
typedef f32x4 __attribute__((address_space(1))) gf32x4;
// A weight image as a BUFFER: wp[i] with a wave-uniform index i is one buffer_load_dwordx4 -- resource descriptor (base) and
// byte offset in SGPRs, computed on the scalar unit, ONE VGPR for the lane offset, shared by every stream of the kernel --
// instead of a 64-bit per-lane pointer per stream (a VGPR pair each, v_mad_u64_u32 / v_lshl_add_u64 / quarter-rate
// v_mul_lo_u32 per load, 64-bit VALU adds wherever consecutive loads are further apart than the 4 KB immediate reaches;
// 303 of the 311 wide loads of villin's kernel, 21 % of its dynamic VALU was INT32).  Writing the address as
// (uniform base + uniform index)[32-bit lane] is not enough (DFF_SADDR=1): the compiler merges base, index and lane into one
// vector pointer and offsets that.  DFF_SADDR=0: the per-lane pointers of rounds 1-4.
#ifndef DFF_SADDR
#define DFF_SADDR 2
#endif
template <class T> struct WVal;
template <> struct WVal<f32x4 __attribute__((address_space(1)))> { typedef f32x4 type; };
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
template <> struct WVal<u32x4_ __attribute__((address_space(1)))> { typedef u32x4_ type; };
// MODE: 2 buffer, 1 saddr-shaped pointer arithmetic, 0 per-lane pointer.  DFF_WMODE(MT): what a GEMM core with MT row tiles uses
// -- measured (round 5, us / step, buffer vs per-lane pointers): BBA's shape (96,2,2) 281 vs 291.5, trp-cage's (128,2,2) 300.5
// vs 304.8, villin's (128,3,1) 487-495 vs 487-490, protein G's (128,4,1) 469-472 vs 465-467: the shapes with three and four
// row tiles keep the per-lane pointers (their cores hold more streams' worth of SGPR offsets than they have SGPRs for).
#define DFF_WMODE(MT) (DFF_SADDR == 2 ? ((MT) <= 2 ? 2 : 0) : DFF_SADDR)
template <class GT, int MODE = DFF_SADDR>
struct WPtr;
template <class GT>
struct WPtr<GT, 2> {
    typedef typename WVal<GT>::type V;
    __amdgpu_buffer_rsrc_t rs;
    unsigned lob;     // this lane's byte offset
    DEVI WPtr(const GT* base, unsigned lane) : rs(__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000)), lob(lane * 16u) {}
    DEVI V operator[](size_t i) const {   // i: wave-uniform, in 16-byte elements
        return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, lob, (unsigned)i * 16u, 0));
    }
    DEVI float first_float(size_t i) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lob, (unsigned)i * 16u, 0)); }
};
template <class GT, int MODE>
struct WPtr {
    typedef typename WVal<GT>::type V;
    const GT* base;   // wave-uniform
    unsigned lo;      // this lane's element offset (< 64: cannot wrap)
    DEVI WPtr(const GT* b, unsigned lane) : base(b), lo(lane) {}
    DEVI V operator[](size_t i) const {
        if constexpr (MODE == 1) return (base + i)[lo];
        else return (base + lo)[i];
    }
    DEVI float first_float(size_t i) const { return *(const gfloat*)(base + i + lo); }
};
// Cache policy of the stash traffic that is written once and read once (per head: q_ext | k | v rows, P, gelu'(h_pre) out; the
// backward's head reload and its gelu' rows in).  POL bit 0: non-temporal LOADS, bit 1: non-temporal STORES.  Round 5, us / step
// (sites: 1 head reload, 2 q|k|v out, 4 P out, 8 gelu' out, 16 gelu' in; DFF_STASH_SITES forces a site mask):
//   villin (three row tiles, 1.3 MB of stash per protein and step)   none 493-504 | 1: 478 | 1+16: 477.6 | 1+2 / 1+4 / 1+8: 480-481 | all stores too: 488.5
//   protein G, two workgroups per protein (four row tiles)           none 482-488 | 1: 473-475 | 1+16: 426-430 | 1+2: 428.5 | 1+4: 428-433 | 1+2+4+8: 432 | all: 434
//   protein G at 256 per GPU (one workgroup per protein)             none 818-825 | 1+2+4+8: 745 | all: 748
//   trp-cage / BBA (two row tiles), loads | stores non-temporal: +1 % | +2-3.5 %  -> default policy
// A stash line that is read exactly once should not stay in the XCD's 4 MB L2 next to the weight images every workgroup
// streams: with three / four row tiles the resident workgroups' stash outgrows it (L2 hit rate 0.87 / 0.76 in profiles/r04).
// Both load sites go non-temporal there, the stores keep the default policy (they are what the loads then find in the MALL).
// DFF_STASH_NT >= 0 forces one policy everywhere (0 = default, 1 = loads, 2 = stores, 3 = both).  The <= 16-row kernels keep
// the default (round 2: -3 % with both non-temporal; the headline variant has no stash traffic in the sampling loops at all).
#ifndef DFF_STASH_NT
#define DFF_STASH_NT -1
#endif
#define DFF_STASH_POL(MT) (DFF_STASH_NT >= 0 ? DFF_STASH_NT : ((MT) >= 3 ? 1 : 0))
// per-site form for experiments: DFF_STASH_SITES = bit mask of the sites that go non-temporal (whatever MT) --
//   1 head reload (q_ext | k | v | P in), 2 q_ext | k | v out, 4 P out, 8 gelu' out, 16 gelu' in;  < 0: DFF_STASH_POL(MT) at every site
#ifndef DFF_STASH_SITES
#define DFF_STASH_SITES -1
#endif
#define DFF_SITE_LD(MT, bit) (DFF_STASH_SITES >= 0 ? ((DFF_STASH_SITES & (bit)) ? 1 : 0) : (DFF_STASH_POL(MT) & 1))
#define DFF_SITE_ST(MT, bit) (DFF_STASH_SITES >= 0 ? ((DFF_STASH_SITES & (bit)) ? 2 : 0) : (DFF_STASH_POL(MT) & 2))
template <int POL = 0> DEVI void st_ntg(gfloat* p, float v) {
    if constexpr (POL & 2) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int POL = 0> DEVI float ld_ntg(const gfloat* p) { if constexpr (POL & 1) return __builtin_nontemporal_load(p); else return *p; }
template <int POL = 0> DEVI f32x4 ld_ntg4(const gfloat* p) {
    if constexpr (POL & 1) return __builtin_nontemporal_load((const gf32x4*)p); else return *(const gf32x4*)p;
}
template <int POL = 0> DEVI void st_ntg4(gfloat* p, const f32x4 v) {
    if constexpr (POL & 2) __builtin_nontemporal_store(v, (gf32x4*)p); else *(gf32x4*)p = v;
}
// four consecutive floats (16-byte aligned) into a scalar aux array of an epilogue
template <int POL = 0>
DEVI void ld4_aux(float* aux, const gfloat* p) {
    const f32x4 v = ld_ntg4<POL>(p);
    aux[0] = v[0]; aux[1] = v[1]; aux[2] = v[2]; aux[3] = v[3];
}

// ------------------------------------------------------------------------------------------
// Stash layout (floats, per workgroup).  R = G*N allocated rows, F = 4H.
// ------------------------------------------------------------------------------------------
struct StashLayout {
    unsigned nodes_in, attn_out, ff, h_pre, qkvx, P, m12;  // offsets inside a layer slot
    unsigned PS;          // leading dimension of a stashed probability row (16 * row tiles)
    unsigned layer_stride;
    unsigned dn_spill;   // offset of the (R,H) spill slot for nodes / dn (after all layers)
    unsigned junk;       // 256 floats nobody reads: where the stash stores of pad rows / surplus tiles land, so that every
                         // epilogue issues the same number of stores (the compiler can then count the loads in flight)
    unsigned total;
};

// qkvx: per head an (R x 208) block, row = [q(64) | u(16) | k(64) | v(64)] (the "extended head" of
// dff_internal.h); P: per head an (R x PS) block of softmax rows over ALL rows of the workgroup
// (zeros outside the row's own protein).
// mt = row tiles of the kernel variant that runs (its compile-time MT): a stashed probability row is 16 MT floats, and a
// shape may be larger than the rows need (hidden 128 at 32 rows runs the three-row-tile shape: two do not fit the LDS).
__host__ __device__ inline StashLayout dff_stash_layout(int N, int G, int H, int L, int mt) {
    StashLayout s;
    const unsigned R = (unsigned)(G * N), F = 4u * H;
    unsigned o = 0;
    s.PS = 16u * (unsigned)mt;
    s.nodes_in = o; o += R * H;
    s.attn_out = o; o += R * H;
    s.ff = o;       o += R * H;
    s.h_pre = o;    o += R * F;
    s.qkvx = o;     o += (unsigned)DFF_HEADS * R * DFF_QKVW;
    s.P = o;        o += (unsigned)DFF_HEADS * R * s.PS;
    s.m12 = o;      o += (unsigned)DFF_HEADS * R * 4;   // GEN: [sum_j a x_j (3) | sum_j a |x_j|^2] per head and row
    s.layer_stride = o;
    s.dn_spill = o * (unsigned)L;
    s.junk = (s.dn_spill + R * (H + 4) + 3u) & ~3u;   // dn_spill is indexed with the LDS leading dimension H + 4
    s.total = (s.junk + 256u + 63u) & ~63u;
    return s;
}

// ------------------------------------------------------------------------------------------
// LDS layout (floats).  Computed identically on host (for the launch size) and device.
// Head-group region Rg: four (R x LQ) buffers Q_ext | K_ext | V_ext | G_ext, a head being 80
// columns [64 | 16 extension]; Pbuf / dSbuf: per head of the group a (16MT x PL) tile array.
// ------------------------------------------------------------------------------------------
#ifndef DFF_KVSPLIT
#define DFF_KVSPLIT 1
#endif
template <int H, int MT, int HGS, bool SPILL>
struct LdsLayout {
    static constexpr int LH = H + 4;
    static constexpr int LQ = 80 * HGS + 4;
    static constexpr int F = 4 * H;
    static constexpr int FC = (F % 256 == 0) ? 256 : 128;
    static constexpr int LF = FC + 4;
    static constexpr int NREG = MT < 4 ? 5 : 4;   // a fifth head-group buffer (backward: dQ_ext) where LDS allows
    static constexpr int LHS2 = (H + DFF_SPAD) / 2;   // dwords per row of a bf16 piece of the split A operand (SPW variants)
    // SPW variants with the fifth buffer: dK / dV leave their producer as bf16 pieces (co_dv_dk), [h | m] in place of the
    // fp32 row, dK's l pieces in the extension columns of the R1 / R2 rows, dV's in `lsplit` (LSV dwords per row) where the
    // LDS has room for it (DFF_KVSPLIT, VSP)
    static constexpr bool KVS = NREG == 5 && DFF_KVSPLIT;
    static constexpr bool VSP = KVS;   // (round 4: also BBA's shape (96,2,2) -- the fp32 abuf the split engine no longer has paid for the area)
    static constexpr int LSV = 32 * HGS + 4;
    // P / dS tile arrays.  Default: 16 MT rows x (16 MT + 4) floats per head.  TIGHT (split engine at four row tiles, i.e.
    // protein G: 56 rows; round 4): the workgroup's ALLOCATED rows only, 16 MT - 4 = 60 floats each -- columns 60..63 do not
    // exist (the phases that would touch them are guarded: they hold pad beads, exact zeros) -- and the regions are ordered
    //   [... | asplit | Rg = R0 R1 R2 R3 | dSbuf | Pbuf]
    // so that what outgrows R3 / Rg in this variant continues into buffers that are idle then: the bf16 pieces of o (forward,
    // 3 x R x 40 dwords from R3 on: dSbuf is idle) and of the FFN hidden chunk (3 x R x 136 dwords from R0 on: dSbuf and Pbuf
    // are idle).  That, and the fp32 abuf the split engine no longer has, is what lets the split A operand (48 KB at 56
    // rows) into the 160 KB next to four head buffers.
    static constexpr bool TIGHT_OK = MT == 4 && HGS == 1;
    static constexpr int PL_WIDE = 16 * MT + 4, PL_TIGHT = 16 * MT - 4;
    template <bool SPW> static constexpr int pl() { return (SPW && TIGHT_OK) ? PL_TIGHT : PL_WIDE; }
    unsigned xst, xs, dxs, vst, cm, tn, prof, prow, rsc, dxw, m12, abuf, resbuf, Pbuf, dSbuf, Rg, asplit, lsplit, junk, total;
    unsigned PT;   // floats of one head's P (or dS) tile array
    __host__ __device__ LdsLayout(int N, int G, bool spw = false) {
        const unsigned R = (unsigned)(G * N);
        const bool tight = spw && TIGHT_OK;
        PT = tight ? R * PL_TIGHT : 16u * MT * PL_WIDE;
        unsigned o = 0;
        xst = o;   o += R * 4;
        xs = o;    o += R * 4;
        dxs = o;   o += R * 4;
        vst = o;   o += R * 4;
        cm = o;    o += 16 * 4 * 2;
        tn = o;    o += 16;
        prof = o;  o += 2 * DFF_NPROF;
        prow = o;  o += 64;                    // protein index of each row (-1: pad row)
        rsc = o;   if (spw) o += 64;           // fp16 engine: inverse row scales of the backward chain in flight (round 5)
        dxw = o;   o += DFF_NWAVES * R * 4;   // per-wave partial dE/dx (summed once per step)
        m12 = o;   if (!tight) o += HGS * R * 4;   // GEN: reloaded [m1 | m2] of the head group (backward); no GEN variant is TIGHT
        abuf = o;  if (!spw) o += R * LH;   // (split engine: the row stages write the bf16 pieces themselves, no fp32 copy)
        resbuf = o; if (!SPILL) o += R * LH;
        if (tight) {
            asplit = o; o += 3u * R * LHS2;
            Rg = o;     o += (unsigned)NREG * R * LQ;
            dSbuf = o;  o += HGS * PT;
            Pbuf = o;   o += HGS * PT;
            // (slack: clamped / skipped operand reads of the last tile rows run a few floats past Pbuf)
            lsplit = o;
            junk = o;   o += 64;
            total = o;
            return;
        }
        Pbuf = o;  o += HGS * PT;
        dSbuf = o; o += HGS * PT;
        Rg = o;
        unsigned rsz = (unsigned)NREG * R * LQ;
        if (R * LF > rsz) rsz = R * LF;
        if (R * LH > rsz) rsz = R * LH;
        o += rsz + 64;  // slack: clamped A-fragment reads never leave the allocation
        asplit = o;
        if (spw) o += 3u * R * LHS2;   // [h | m | l] bf16 pieces of abuf (R x H each, row stride H + 8)
        lsplit = o;
        if (spw && VSP) o += 2 * R * LSV;   // l pieces of dV, then of dQ (co_ds; round 4)
        junk = o;  o += 64;                    // landing pad of the L2 warm-up loads (l2_touch)
        total = o;
    }
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// all-reduce over an aligned group of 16 lanes (one DPP "row") without touching LDS:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
template <int CTRL>
DEVI float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// thread index behind an opaque barrier: every stage re-derives its lane / row / address arithmetic from
// a fresh copy, so LICM cannot hoist hundreds of per-stage invariants to the top of the kernel (where
// they spill and come back through scratch loads that drain the weight rings).
DEVI int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
DEVI float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
DEVI float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
DEVI float grp_sum(float v, int np) {
    for (int o = np >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// GELU(erf) value and derivative in one go: the forward FFN epilogue stashes gelu'(h_pre) so that the
// backward epilogue is a single multiply
// One exponential serves both: erfc(z) = exp(-z^2) t P(t), t = 1 / (1 + p z) (the Abramowitz-Stegun 7.1.26 form with a
// degree-7 polynomial, least-squares fit of the absolute error on z in [0, 8]: 1.9e-10 in exact arithmetic), z = |x| / sqrt 2,
// so  cdf = 1 - erfc(z) / 2 (x >= 0) or erfc(z) / 2,  pdf = exp(-z^2) / sqrt(2 pi).  ~24 VALU instructions (one v_rcp_f32,
// one v_exp_f32) instead of ~110 for erff + expf; evaluated in fp32 its error against float64 is that of the libm-based
// form (g: 6.1e-7 at |x| = 12, 3.4e-7 for |x| < 4; g': 2.0e-7 vs 1.5e-7).  The FFN epilogue was the largest single VALU
// item of the sampler kernels (a third of all VALU instructions of a chignolin step).
DEVI void gelu_both(float x, float& g, float& gp) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.505f, z, 1.0f));
    float P = -5.364335749e-02f;
    P = fmaf(P, t, 3.483687174e-01f); P = fmaf(P, t, -8.750633503e-01f); P = fmaf(P, t, 9.397950760e-01f);
    P = fmaf(P, t, -3.177378563e-01f); P = fmaf(P, t, 4.184097744e-01f); P = fmaf(P, t, 2.522358196e-01f);
    P = fmaf(P, t, 2.876351766e-01f);
    const float E = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);   // exp(-x^2 / 2)
    const float q = 0.5f * (t * P) * E;
    const float cdf = x >= 0.f ? 1.0f - q : q;
    g = x * cdf;
    gp = fmaf(x * 0.39894228040143267794f, E, cdf);
}

// Hardware transcendentals (v_exp_f32, v_rcp_f32, v_rsq_f32: ~1 ulp each) for the softmax, the gates and the LayerNorm
// scale: they sit on serial row-stage paths and in the softmax of every head, where libm's expf / IEEE division / sqrtf
// cost 10-25 instructions apiece.  The extra error is of the order of one fp32 rounding (forces vs the reference's
// float64 run stay within the tolerances of tests/test_gpu_parity.py).
#ifndef DFF_LIBM_ROWMATH
#define DFF_LIBM_ROWMATH 0   // 1: expf / division / sqrtf as in round 1
#endif
DEVI float fast_exp(float x) {   // e^x
#if DFF_LIBM_ROWMATH
    return expf(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#endif
}
DEVI float fast_rcp(float x) {
#if DFF_LIBM_ROWMATH
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
DEVI float fast_rsqrt(float x) {
#if DFF_LIBM_ROWMATH
    return 1.0f / sqrtf(x);
#else
    return __builtin_amdgcn_rsqf(x);
#endif
}
DEVI float sigmoid_f(float z) { return fast_rcp(1.0f + fast_exp(-z)); }
DEVI void st_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
DEVI float ld_nt(const float* p) { return __builtin_nontemporal_load(p); }

#ifndef DFF_LIBM_NORMAL
#define DFF_LIBM_NORMAL 0   // 1: logf / sqrtf / sinf / cosf of libm (rounds 1-2)
#endif
// Philox4x32-10 (Salmon et al. 2011), counter-based: the same (key, counter) always gives the
// same 4 words, so a trajectory's noise does not depend on how the batch is sharded.
DEVI void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                        uint32_t c3, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// standard normal number `c` (0..2) for (item, step, bead): Box-Muller on Philox words.
DEVI float philox_normal(uint64_t seed, uint64_t item, uint64_t step, uint32_t bead, int c) {
    uint32_t w[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)item,
                  (uint32_t)(item >> 32) ^ (bead << 8), (uint32_t)step, (uint32_t)(step >> 32), w);
    const float inv = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = ((float)w[(c >> 1) * 2] + 0.5f) * inv;          // (0,1]
    const float u1 = ((float)w[(c >> 1) * 2 + 1] + 0.5f) * inv;
#if DFF_LIBM_NORMAL
    const float r = sqrtf(-2.0f * logf(u0));
    const float th = 6.28318530717958647692f * u1;
    return (c & 1) ? r * sinf(th) : r * cosf(th);
#else
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
    // revolutions: u1 itself) -- ~10 instructions instead of ~200 for libm's logf / sinf / cosf with their range reductions.
    // The draws are standard normals to ~1e-6 either way; every kernel draws through this one function, so trajectories stay
    // independent of sharding, chunking and kernel variant.
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));   // sqrt(-2 ln u0)
    return r * ((c & 1) ? __builtin_amdgcn_sinf(u1) : __builtin_amdgcn_cosf(u1));
#endif
}

// ------------------------------------------------------------------------------------------
// SPW variants (opt-in, DFF_SPLIT_BF16=1): the K = H weight GEMMs on the bf16 matrix pipe at fp32 accuracy.
// An fp32 value is the exact sum of three bf16 pieces (truncation split: h = top 16 bits, r = a - h exactly, ...);
// a.b ~ ah.bh + (am.bh + ah.bm) + (al.bh + ah.bl + am.bm) drops only terms of order 2^-24 (tools_ubench/split_bf16.hip:
// error vs fp64 <= that of v_mfma_f32_16x16x4_f32).  The weights are split on the host (dff_host.hip pack_b_split),
// the activations by split_rows once per GEMM input; six v_mfma_f32_16x16x32_bf16 (16 cycles each, and they leave the
// vector port free: tools_ubench/overlap3.hip) replace eight v_mfma_f32_16x16x4_f32 (32 cycles each).
// ------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned lu32;
typedef __attribute__((address_space(3))) u32x4 lu32x4;
typedef __attribute__((address_space(3))) unsigned short lu16;
typedef __attribute__((address_space(1))) u32x4 gu32x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lu32x2;
DEVI f32x4 mfma_bf16(const u32x4 a, const u32x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- two-piece fp16 split (round 5): x = h + l' / 2048 with h = RN_f16(x), l' = RN_f16((x - h) 2048), both rounded to nearest
// (split2h), and  a.b ~ ah.bh + (ah.bl' + al'.bh) / 2048  -- THREE v_mfma_f32_16x16x32_f16 instead of six bf16 ones and 4 instead
// of 6 bytes per weight.  Dropped: al.bl (2^-22) and the pieces' rounding (<= 2^-22 each, unbiased); measured against float64
// the products are as accurate as an fp32 FMA chain (profiles/r05/f16_engine), and end to end the forces are closer to the
// reference's float64 run than its own float32 run is.  fp16 has 5 exponent bits: the hardware keeps subnormals in
// v_cvt_pk_f16_f32 and in the MFMA inputs (tools_ubench/f16_denorm.hip), so a piece resolves 2^-24 / 2048 = 3e-11 ABSOLUTE
// whatever the element's size -- negligible next to O(1) forward activations and O(0.1) weights; the backward's operands
// (gradients, any magnitude) are scaled row-wise by a power of two first.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
DEVI f32x4 mfma_f16(const u32x4 a, const u32x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#define DFF_F16_LSCALE 2048.0f
#define DFF_F16_LINV (1.0f / 2048.0f)
// one pair of values -> (h pair, l' pair), element 0 in the low half
// (both pieces rounded to NEAREST, v_cvt_pk_f16_f32: |x - h| <= 2^-11 |x| is exact in fp32, l' carries it to 2^-11 of itself, so
// h + l' / 2048 = x (1 + e), |e| <= 2^-22 and unbiased; the truncating v_cvt_pkrtz_f16_f32 this started with leaves e in (-2^-20, 0]
// -- one-sided, so the errors of a long dot product add up instead of cancelling)
DEVI void split2h(float e0, float e1, unsigned& h, unsigned& l) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    const f16x2_ ph = __builtin_convertvector(((f32x2_){e0, e1}), f16x2_);
    const float r0 = e0 - (float)ph[0], r1 = e1 - (float)ph[1];
    h = __builtin_bit_cast(unsigned, ph);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(((f32x2_){r0 * DFF_F16_LSCALE, r1 * DFF_F16_LSCALE}), f16x2_));
}
// eight consecutive fp32 -> the two fp16 operands (the layout of split8 below)
DEVI void split8h(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float e0 = q < 2 ? x0[2 * q] : x1[2 * q - 4], e1 = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
        unsigned hh, ll;
        split2h(e0, e1, hh, ll);
        h[q] = hh; l[q] = ll;
    }
}
// one value -> its two 16-bit pieces (row stages)
DEVI void split1h(float v, unsigned short& h, unsigned short& l) {
    unsigned hh, ll;
    split2h(v, 0.f, hh, ll);
    h = (unsigned short)hh; l = (unsigned short)ll;
}
// power-of-two scale that brings a row maximum `mx` (>= 0) to [16, 32): s = 2^(4 - floor(log2 mx)), and its inverse; exact.
// (the exponent is clamped: a zero / denormal row gets 2^67, whose products with the row are still zero / tiny)
DEVI void pow2_scale(float mx, float& s, float& inv) {
    int ex = (int)((__float_as_uint(mx) >> 23) & 255u);
    ex = ex < 64 ? 64 : (ex > 250 ? 250 : ex);
    s = __uint_as_float((unsigned)(258 - ex) << 23);     // 2^(127 + 4 - (ex - 127) - 127) -> biased 258 - ex
    inv = __uint_as_float((unsigned)(ex - 4) << 23);     // 2^((ex - 127) - 4)            -> biased ex - 4
}

// eight consecutive fp32 (x0 = k 0..3, x1 = k 4..7) -> the three bf16-pair operands of v_mfma_f32_16x16x32_bf16 (~44 VALU)
DEVI void split8(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float e0 = q < 2 ? x0[2 * q] : x1[2 * q - 4], e1 = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
        const unsigned b0 = __float_as_uint(e0), b1 = __float_as_uint(e1);
        const float r0 = e0 - __uint_as_float(b0 & 0xffff0000u), r1 = e1 - __uint_as_float(b1 & 0xffff0000u);
        const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
        const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);
        h[q] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        m[q] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
        l[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
}

// ------------------------------------------------------------------------------------------
// DFF_PROF=1 (./build.sh with DFF_EXTRA_FLAGS=-DDFF_PROF=1; tools_profile_stages.py needs such a build): the stage ticks are
// compiled in.  The product build leaves them out: ~30 exec-masked branch sites per layer in every wave's instruction stream.
#ifndef DFF_PROF
#define DFF_PROF 0
#endif
// optional per-stage cycle accounting (a.prof != null): thread 0 of block 0 accumulates
// s_memtime deltas per stage id at stage boundaries; written out at kernel end.
// ------------------------------------------------------------------------------------------
struct Prof {
    unsigned long long* out;
    unsigned long long last;
    unsigned long long* acc;   // LDS
    bool on;
    DEVI void tick(int id) {
#if !DFF_PROF
        (void)id;
        return;
#endif
        if (on) {
            const unsigned long long t = __builtin_readcyclecounter();
            acc[id] += t - last;
            last = t;
        }
    }
};

// ------------------------------------------------------------------------------------------
// per-workgroup context
// ------------------------------------------------------------------------------------------
struct Ctx {
    int N, G, gcnt, rows, NP, L;
    int b0;
    float *xst, *xs, *dxs, *vst, *cm, *tn, *abuf, *resbuf, *Pbuf, *dSbuf, *Rg;
    float* rsc;    // fp16 engine (LdsLayout::rsc): rsc[row] = 1 / (power-of-two scale of the row's backward GEMM input)
    lu32* asp;     // split engine: the bf16 pieces of the K = H GEMM input (LdsLayout::asplit), RNa allocated rows
    int RNa;
    float* stash;  // this workgroup's slot
    const float* l0;   // layer-0 slot the x-independent nodes_in / q|u|k|v are READ from (table entry or own stash)
    StashLayout sl;
};

// x-independent layer-0 node features: node_embedding([one_hot(i), t])  (graph_transformer.py:

// geometry every attention phase needs
DEVI float quad_sum(float v) {        // all-reduce over the 4 lanes of a DPP quad
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
DEVI float quad_bcast3(float v) { return dpp_mov<0xFF>(v); }   // lane 3 of the quad to all four

// ------------------------------------------------------------------------------------------
// centring helpers (utils.py:65-70): per-protein mean over beads
// ------------------------------------------------------------------------------------------
DEVI void bead_mean(const Ctx& c, const float* src, float* cm) {
    const int tid_ = tid_now();
    if (tid_ < c.gcnt * 4) {
        const int g = tid_ >> 2, cc = tid_ & 3;
        float s = 0.f;
        for (int i = 0; i < c.N; ++i) s += src[(g * c.N + i) * 4 + cc];
        cm[tid_] = s / (float)c.N;
    }
}

// ---- <= 16-row kernel (dff_small.hip): stash layout, shared with the host dispatcher
#define DFF_XH 80       // extended head width
#define DFF_XLD 84      // leading dim of the per-wave head buffers
#define DFF_PLD 20      // leading dim of the per-wave P / dS tiles

struct SmallStash {
    unsigned nodes_in, attn_out, ff, h_pre, qkv, P, m12;
    unsigned layer_stride, total;
};
__host__ __device__ inline SmallStash dff_small_stash(int N, int G, int H, int L) {
    SmallStash s;
    // every array has one extra "dummy" row (index G*N) that absorbs the stores of pad lanes
    // (rows >= real rows of the 16-row MFMA tile), so epilogues need no exec-masked branches;
    // P keeps all 16 rows (its pad rows must read back as exact zeros).
    const unsigned R = (unsigned)(G * N) + 1u, F = 4u * H;
    unsigned o = 0;
    s.nodes_in = o; o += R * H;
    s.attn_out = o; o += R * H;
    s.ff = o;       o += R * H;
    s.h_pre = o;    o += R * F;
    s.qkv = o;      o += DFF_HEADS * R * DFF_QKVW;
    s.P = o;        o += DFF_HEADS * 16 * 16;
    s.m12 = o;      o += DFF_HEADS * 16 * 4;   // GEN: [sum_j a x_j (3) | sum_j a |x_j|^2] per head and row
    s.layer_stride = o;
    s.total = (o * (unsigned)L + 63u) & ~63u;
    return s;
}


// ---- kernel lookup, exported by the kernel translation units to the host dispatcher (dff_host.hip)
struct Variant {
    int H, MT, HGS;
    bool spill, gen, spw;
    const void* fn;
    unsigned (*lds_floats)(int N, int G);
    const char* name;
    bool pair = false;   // two workgroups per protein (dff_fused_kernel<..., PAIR>)
};
const Variant* dff_fused_variants(int* count);                                  // dff_kernels.hip
int dff_debug_gemm_launch(int K, const float* dA, const float* dW, int M, int Nout, float* dO, size_t lds);   // dff_kernels.hip
// the <= 16-row kernel for (H, NW waves, input-branch variant, split-bf16 weight GEMMs); false if not built
// (one translation unit per sampler mode, DFF_MODE_SCORE / LANGEVIN / DDPM = 0 / 1 / 2: dff_small.hip compiled three times)
// (pair: the two-workgroups-per-protein variant -- the FOLD kernel of the sampling loops only; 4 waves per workgroup)
bool dff_small_pick_m0(int H, int NW, bool gen, bool spw, const void** fn, unsigned* lds_floats, const char** name, bool fold, bool pair);
bool dff_small_pick_m1(int H, int NW, bool gen, bool spw, const void** fn, unsigned* lds_floats, const char** name, bool fold, bool pair);
bool dff_small_pick_m2(int H, int NW, bool gen, bool spw, const void** fn, unsigned* lds_floats, const char** name, bool fold, bool pair);
inline bool dff_small_pick(int mode, int H, int NW, bool gen, bool spw, const void** fn, unsigned* lds_floats, const char** name, bool fold = false,
                           bool pair = false) {
    return mode == 0 ? dff_small_pick_m0(H, NW, gen, spw, fn, lds_floats, name, fold, pair)
         : mode == 1 ? dff_small_pick_m1(H, NW, gen, spw, fn, lds_floats, name, fold, pair)
                     : dff_small_pick_m2(H, NW, gen, spw, fn, lds_floats, name, fold, pair);
}
