// dff_host.hip -- host half of libdff_amd.so: weight folding + MFMA packing, schedule tables,
// scratch management, kernel dispatch and the extern "C" ABI declared in include/dff.h.
#include "../../include/dff.h"
#include "dff_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

// the two sampler kernels are their own translation units (dff_kernels.hip, dff_small.hip: they export variant
// lookups, dff_device.h); the small PWD kernels are compiled here
#include "dff_device.h"
#include "dff_pwd.hip"

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return fail(DFF_EHIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

// Every ABI entry runs on the model's device and leaves the caller's current device as it found it (a process that
// drives several GPUs keeps torch's notion of the current device).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (cur != dev) {
            if (hipSetDevice(dev) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(dev)                                                         \
    DeviceGuard dev_guard_(dev);                                               \
    if (!dev_guard_.ok) return fail(DFF_EHIP, "cannot select device %d", (int)(dev))

// ------------------------------------------------------------------------------------------
// packing for the v_mfma_f32_16x16x4_f32 B operand (layout: dff_internal.h)
// ------------------------------------------------------------------------------------------
// xper > 0: k-blocks with kb % xper == 4 are head EXTENSION blocks, of which only rows 0..3 are ever non-zero
// (u | s, xrel | D, du | ds): they are packed with row (lane >> 4) in k-step 0 and zeros in k-steps 1..3, and the
// kernels issue one MFMA for them instead of four (A operand: column lane >> 4 of the block).
static std::vector<float> pack_b(int K, int Nout, const std::function<double(int, int)>& w, int xper = 0) {
    const int KB = (K + 15) / 16, NT = (Nout + 15) / 16;
    std::vector<float> out((size_t)KB * NT * 256, 0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb) {
            const bool ext = xper > 0 && kb % xper == 4;
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < (ext ? 1 : 4); ++s) {
                    const int k = 16 * kb + (ext ? lane >> 4 : 4 * (lane >> 4) + s), n = 16 * nt + (lane & 15);
                    if (k < K && n < Nout) out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + s] = (float)w(k, n);
                }
        }
    return out;
}

// B operand of v_mfma_f32_16x16x32_bf16 for the split-bf16 GEMMs (dff_kernels.hip gemm_wide_split): per (tile,
// 32-row k-block, piece h | m | l, lane) eight bf16 = W[32 kb + 8 (lane >> 4) + j][16 nt + (lane & 15)], j = 0..7,
// two per dword (even j in the low half).  w = h + m + l exactly (truncation split of the fp32 weight).
static std::vector<uint32_t> pack_b_split(int K, int Nout, const std::function<double(int, int)>& w) {
    const int KB = K / 32, NT = (Nout + 15) / 16;
    std::vector<uint32_t> out((size_t)NT * KB * 3 * 64 * 4, 0u);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * (lane >> 4) + j, n = 16 * nt + (lane & 15);
                    const float v = n < Nout ? (float)w(k, n) : 0.f;
                    const uint32_t h = bits(v) & 0xffff0000u;
                    const float r = v - flt(h);
                    const uint32_t mm = bits(r) & 0xffff0000u;
                    const float r2 = r - flt(mm);
                    const uint32_t pc[3] = {h >> 16, mm >> 16, bits(r2) >> 16};
                    for (int p = 0; p < 3; ++p)
                        out[((((size_t)nt * KB + kb) * 3 + p) * 64 + lane) * 4 + (j >> 1)] |= pc[p] << (16 * (j & 1));
                }
    return out;
}

// Split-bf16 images for the <= 16-row kernel's SPW variants (dff_small.hip, split engine): a sequence of UNITS, each one
// 16-column output tile nt x 32 A-columns starting at c0, stored [piece h | m | l][lane][8 bf16].  Element j of lane
// (n = lane & 15, kg = lane >> 4) is W(c0 + 16 (j >> 2) + 4 kg + (j & 3), 16 nt + n): the k order in which that kernel's A
// fragments hold a 32-column block (two ds_read_b128, at columns 4 kg and 16 + 4 kg).  w = h + m + l exactly.
static uint16_t f32_to_f16_rn(float f);
static float f16_to_f32(uint16_t h);
// pack_b_split in the two-piece fp16 format of the round-5 engine: per (tile, 32-row k-block, piece h | l', lane) eight fp16
static std::vector<uint32_t> pack_b_split_f16(int K, int Nout, const std::function<double(int, int)>& w) {
    const int KB = K / 32, NT = (Nout + 15) / 16;
    std::vector<uint32_t> out((size_t)NT * KB * 2 * 64 * 4, 0u);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * kb + 8 * (lane >> 4) + j, n = 16 * nt + (lane & 15);
                    const float v = n < Nout ? (float)w(k, n) : 0.f;
                    const uint16_t h = f32_to_f16_rn(v);
                    const uint16_t l = f32_to_f16_rn((v - f16_to_f32(h)) * 2048.0f);
                    out[((((size_t)nt * KB + kb) * 2 + 0) * 64 + lane) * 4 + (j >> 1)] |= (uint32_t)h << (16 * (j & 1));
                    out[((((size_t)nt * KB + kb) * 2 + 1) * 64 + lane) * 4 + (j >> 1)] |= (uint32_t)l << (16 * (j & 1));
                }
    return out;
}
static std::vector<uint32_t> pack_units(const std::vector<std::pair<int, int>>& units, int Nout,
                                        const std::function<double(int, int)>& w) {
    std::vector<uint32_t> out(units.size() * 3 * 64 * 4, 0u);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (size_t u = 0; u < units.size(); ++u) {
        const int nt = units[u].first, c0 = units[u].second;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3), n = 16 * nt + (lane & 15);
                const float v = n < Nout ? (float)w(c, n) : 0.f;
                const uint32_t h = bits(v) & 0xffff0000u;
                const float r = v - flt(h);
                const uint32_t mm = bits(r) & 0xffff0000u;
                const float r2 = r - flt(mm);
                const uint32_t pc[3] = {h >> 16, mm >> 16, bits(r2) >> 16};
                for (int q = 0; q < 3; ++q) out[((u * 3 + q) * 64 + lane) * 4 + (j >> 1)] |= pc[q] << (16 * (j & 1));
            }
    }
    return out;
}
// fp32 -> fp16 (binary16), round to nearest even, subnormals kept: what v_cvt_pk_f16_f32 does on the device (dff_device.h split2h)
static uint16_t f32_to_f16_rn(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0u));   // inf / nan
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);             // >= 65520 rounds to infinity (the range guard keeps models away)
    const int e = (int)(a >> 23) - 127;
    if (e < -25) return (uint16_t)sign;                                   // < 2^-25: zero
    uint32_t r, rem, half;
    if (e >= -14) { r = (uint32_t)(e + 15) << 10 | ((a >> 13) & 0x3ffu); rem = a & 0x1fffu; half = 0x1000u; }
    else {                                                                // subnormal: value = mant 2^(e - 23) = r 2^-24
        const uint32_t mant = (a & 0x7fffffu) | 0x800000u; const int sh = -e - 1;   // 14 .. 24
        r = mant >> sh; rem = mant & ((1u << sh) - 1u); half = 1u << (sh - 1);
    }
    if (rem > half || (rem == half && (r & 1u))) ++r;                     // (a carry out of the mantissa lands in the exponent: right)
    return (uint16_t)(sign | r);
}
static float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, mnt = h & 0x3ffu;
    float f;
    if (e == 0) { f = (float)mnt * 5.9604644775390625e-08f; uint32_t u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    const uint32_t u = sign | ((e == 31 ? 255u : e + 112u) << 23) | (mnt << 13);
    memcpy(&f, &u, 4);
    return f;
}
// The same unit sequence as TWO fp16 pieces per weight for the <= 16-row FOLD kernel's fp16 engine (dff_small.hip, stream kind
// -1): [piece h | l'][lane][8 fp16], h = RN_f16(w), l' = RN_f16((w - h) 2048): w = h + l' / 2048 to 2^-22 relative (3e-11 absolute).
static std::vector<uint32_t> pack_units_f16(const std::vector<std::pair<int, int>>& units, int Nout,
                                            const std::function<double(int, int)>& w) {
    std::vector<uint32_t> out(units.size() * 2 * 64 * 4, 0u);
    for (size_t u = 0; u < units.size(); ++u) {
        const int nt = units[u].first, c0 = units[u].second;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3), n = 16 * nt + (lane & 15);
                const float v = n < Nout ? (float)w(c, n) : 0.f;
                const uint16_t h = f32_to_f16_rn(v);
                const uint16_t l = f32_to_f16_rn((v - f16_to_f32(h)) * 2048.0f);
                out[((u * 2 + 0) * 64 + lane) * 4 + (j >> 1)] |= (uint32_t)h << (16 * (j & 1));
                out[((u * 2 + 1) * 64 + lane) * 4 + (j >> 1)] |= (uint32_t)l << (16 * (j & 1));
            }
    }
    return out;
}
static std::vector<std::pair<int, int>> units_wide(int K, int Nout) {   // [tile][k-block]
    std::vector<std::pair<int, int>> u;
    for (int nt = 0; nt < (Nout + 15) / 16; ++nt)
        for (int kb = 0; kb < K / 32; ++kb) u.push_back({nt, 32 * kb});
    return u;
}
static std::vector<std::pair<int, int>> units_tall(int K, int Nout) {   // [k-block][tile]
    std::vector<std::pair<int, int>> u;
    for (int kb = 0; kb < K / 32; ++kb)
        for (int nt = 0; nt < (Nout + 15) / 16; ++nt) u.push_back({nt, 32 * kb});
    return u;
}

// ------------------------------------------------------------------------------------------
// model handle
// ------------------------------------------------------------------------------------------
struct dff_model {
    dff_config cfg;
    int device;
    DffModelDev dev;
    std::vector<void*> allocs;
    std::vector<std::vector<float>> sched;  // 12 tables, host fp32
    float* stash = nullptr;
    size_t stash_floats = 0;
    int group_override = 0;
    bool force_generic = false;   // debugging: never use the rows<=16 fast path
    int small_waves = 0;          // 0 auto, 4: never use the 8-wave variant
    bool last_small = false;
    unsigned long long* prof = nullptr;
    bool prof_on = false;
    int prof_wave = 0;
    // last launch
    const char* last_kernel = "";
    int last_grid = 0, last_lds = 0, last_G = 0, last_B = 0;
    unsigned long long last_stride = 0;
    int last_mt = 1;                           // row tiles of the <= 64-row variant that ran last (stash layout of the debug reads)
    // precomputed layer-0 tables (see ensure_l0_table), keyed independently: [0] one entry at tnorm (Langevin),
    // [1] one entry per noise level (DDPM) -- `sample.py --gen_mode langevin` uses both, alternately
    struct L0Table {
        float* tab = nullptr;
        size_t floats = 0;
        bool valid = false;
        int G = 0, waves = -1;
        float tnorm = 0.f;
        const void* variant = nullptr;
        bool split = false;
        unsigned long long used = 0;           // last use (dff_model::l0_clock): the slot that was idle longest is rebuilt
    } l0[2][3];                                // a few per kind: batches that alternate between group sizes / variants (ADVICE r05:
    unsigned long long l0_clock = 0;           // ala2 crossing 128 / 256 / 384 per GPU) keep their tables instead of rebuilding one slot
    bool l0_off = false;                       // debugging: never use the table
    int max_wgs = 2048;                        // workgroups per launch: bounds the stash (grid x stash slot) for big batches
    int last_base = 0;
    bool split = false;                        // split-bf16 images exist, SPW variants preferred (DFF_SPLIT_BF16=0 at model creation: never)
    bool small_split = false;                  // ... and the <= 16-row kernel has an SPW variant for this model
    bool fold_kv = false;                      // H == 64: k = v = LayerNorm output (W_k folded into W_q, W_v into W_o); DFF_FOLD_KV=0: never
    // PAIR variants (two workgroups per protein): partial-tile exchange slots and flags
    float* xchg = nullptr;
    size_t xchg_floats = 0;
    unsigned* xflag = nullptr;
    size_t xflag_n = 0;
    bool pair_off = false;                     // debugging: never use a PAIR variant
    int pair_slow = 0;                         // tests: 1 = PAIR variants always run the cross-XCD (agent-scope) exchange protocol; 2 = partners are
                                               // ADJACENT blocks (different XCDs under round-robin placement), protocol chosen by the handshake
    bool last_pair = false;
    unsigned sticky = 0;                       // the sticky error word as the host last read it (read_status): a non-zero one
                                               // refuses further PAIR launches WITHOUT touching the device
    int n_cus = 0;                             // hipDeviceAttributeMultiprocessorCount of `device`
};

static int upload_u32(dff_model* m, const std::vector<uint32_t>& h, const unsigned** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, h.size() * sizeof(uint32_t)));
    m->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    *out = (const unsigned*)p;
    return DFF_OK;
}

static int upload(dff_model* m, const std::vector<float>& h, const float** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, h.size() * sizeof(float)));
    m->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (const float*)p;
    return DFF_OK;
}
#define UP(vec, dst)                                  \
    do {                                              \
        int rc_ = upload(m, vec, &(dst));             \
        if (rc_) return rc_;                          \
    } while (0)

extern "C" size_t dff_weight_count(const dff_config* c) {
    if (!c) return 0;
    const size_t H = c->hidden, N = c->n_beads, I = DFF_INNER, F = 4 * H;
    const size_t dec = c->conservative ? 1 : 3;   // node_decoder: Linear(H, 1) or Linear(H, 3)
    const size_t nin = N + 1 + (c->use_abs_coords ? 3 : 0);                                  // graph_transformer.py:53
    size_t nfe = (c->use_intrinsic_coords ? 3 : 0) + (c->use_distances ? 1 : 0);             // :54-58
    if (nfe == 0) nfe = 1;
    size_t n = H * nin + H + H * nfe + H + dec * H + dec;
    const size_t per_layer = I * H + I + 2 * I * H + 2 * I + I * H + I + H * I + H + H + H + 3 * H +
                             F * H + F + H * F + H + H + H + 3 * H;
    return n + per_layer * c->n_layers;
}

// cosine schedule + derived tables, float64 -> float32  (utils.py:52-62, models/ddpm.py:52-99)
static void build_schedule(int T, std::vector<std::vector<float>>& out) {
    const double s = 0.008, pi = 3.14159265358979323846;
    std::vector<double> ac0(T + 1), betas(T), alphas(T), ac(T), acp(T);
    for (int i = 0; i <= T; ++i) {
        const double x = (double)i;  // torch.linspace(0, T, T+1) is exact for integer steps
        const double cv = std::cos(((x / T) + s) / (1 + s) * pi * 0.5);
        ac0[i] = cv * cv;
    }
    for (int i = T; i >= 0; --i) ac0[i] = ac0[i] / ac0[0];
    for (int i = 0; i < T; ++i) {
        double b = 1 - (ac0[i + 1] / ac0[i]);
        betas[i] = b < 0 ? 0 : (b > 0.999 ? 0.999 : b);
        alphas[i] = 1.0 - betas[i];
    }
    double cp = 1.0;
    for (int i = 0; i < T; ++i) { cp *= alphas[i]; ac[i] = cp; }
    for (int i = 0; i < T; ++i) acp[i] = i == 0 ? 1.0 : ac[i - 1];
    out.assign(12, std::vector<float>(T));
    for (int i = 0; i < T; ++i) {
        const double pv = betas[i] * (1.0 - acp[i]) / (1.0 - ac[i]);
        out[0][i] = (float)betas[i];
        out[1][i] = (float)ac[i];
        out[2][i] = (float)acp[i];
        out[3][i] = (float)std::sqrt(ac[i]);
        out[4][i] = (float)std::sqrt(1.0 - ac[i]);
        out[5][i] = (float)std::log(1.0 - ac[i]);
        out[6][i] = (float)std::sqrt(1.0 / ac[i]);
        out[7][i] = (float)std::sqrt(1.0 / ac[i] - 1);
        out[8][i] = (float)pv;
        out[9][i] = (float)std::log(pv < 1e-20 ? 1e-20 : pv);
        out[10][i] = (float)(betas[i] * std::sqrt(acp[i]) / (1.0 - ac[i]));
        out[11][i] = (float)((1.0 - acp[i]) * std::sqrt(alphas[i]) / (1.0 - ac[i]));
    }
}

extern "C" int dff_model_create(const dff_config* cfg, const float* w, size_t n_weights, int device,
                                dff_model** out) {
    if (!cfg || !w || !out) return fail(DFF_EINVAL, "null argument");
    auto is01 = [](int v) { return v == 0 || v == 1; };
    if (!(is01(cfg->use_intrinsic_coords) && is01(cfg->use_distances) && is01(cfg->use_abs_coords) && is01(cfg->conservative)))
        return fail(DFF_EINVAL, "use_intrinsic_coords, use_distances, use_abs_coords and conservative must be 0 or 1");
    const int H = cfg->hidden, N = cfg->n_beads, L = cfg->n_layers, I = DFF_INNER, F = 4 * H;
    // hidden sizes with kernel variants: the shipped 64 / 96 / 128, and 256 (the reference's own smoke test,
    // models/graph_transformer.py:332-359; fp32 engine, <= 32 beads)
    if (!(H == 64 || H == 96 || H == 128 || H == 256)) return fail(DFF_EINVAL, "hidden must be 64, 96, 128 or 256 (got %d)", H);
    if (N < 2 || N > DFF_MAX_BEADS) return fail(DFF_EINVAL, "n_beads must be in [2,%d] (got %d)", DFF_MAX_BEADS, N);
    if (H != 128 && N > 32) return fail(DFF_EINVAL, "n_beads > 32 needs hidden = 128 in this build");
    // (four row tiles: 520 floats of LDS per bead row + 36 KB of tile arrays and bookkeeping -- 61 rows are what 160 KB hold;
    // the largest shipped model, protein G, has 56)
    if (N > DFF_MAX_BEADS_LDS) return fail(DFF_EINVAL, "n_beads > %d does not fit the LDS of this build's kernels (got %d)", DFF_MAX_BEADS_LDS, N);
    if (L < 1 || L > DFF_MAX_LAYERS) return fail(DFF_EINVAL, "n_layers must be in [1,%d]", DFF_MAX_LAYERS);
    if (cfg->timesteps < 1) return fail(DFF_EINVAL, "timesteps must be >= 1");
    if (n_weights != dff_weight_count(cfg))
        return fail(DFF_EINVAL, "expected %zu weights, got %zu", dff_weight_count(cfg), n_weights);
    ON_DEVICE(device);
    dff_model* m = new dff_model();
    struct Undo {   // any failure below frees what was uploaded so far
        dff_model* m;
        ~Undo() { if (m) dff_model_destroy(m); }
    } undo{m};
    m->cfg = *cfg;
    m->device = device;
    HIPCHK(hipDeviceGetAttribute(&m->n_cus, hipDeviceAttributeMultiprocessorCount, device));
    {   // weight GEMMs on the fp16 / bf16 pipe via the split of every fp32 operand, wherever a kernel variant exists for the
        // shape (default); DFF_SPLIT_BF16=0: every GEMM on v_mfma_f32_16x16x4_f32
        const char* e = getenv("DFF_SPLIT_BF16");
        m->split = !(e && e[0] == '0');
        // The fp16 pieces of the FORWARD GEMM inputs are taken of the activations as they are (dff_device.h split8h): LayerNorm
        // outputs, attention outputs (averages of v), GELU(hidden).  Their worst-case magnitudes follow from the weights alone;
        // a model that could take them past fp16's range (|x| < 1.6e4 keeps x and its 2^11-scaled low piece finite) runs the
        // fp32 engine instead.  (The backward's inputs are scaled on the device.)  Trained models are orders of magnitude away.
        if (m->split) {
            const float* q = w;
            auto skip = [&](size_t n) { const float* r = q; q += n; return r; };
            const int NI_ = N + 1 + 3 * cfg->use_abs_coords, nfe_ = (3 * cfg->use_intrinsic_coords + cfg->use_distances) ? (3 * cfg->use_intrinsic_coords + cfg->use_distances) : 1;
            skip((size_t)H * NI_); skip(H); skip((size_t)H * nfe_); skip(H); skip((size_t)(cfg->conservative ? 1 : 3) * H); skip(cfg->conservative ? 1 : 3);
            auto amax = [](const float* a, size_t n) { double mx = 0; for (size_t i = 0; i < n; ++i) mx = std::max(mx, (double)fabsf(a[i])); return mx; };
            auto rowl1 = [](const float* a, int rows, int cols) {   // largest L1 norm of a row of a (rows x cols) matrix
                double mx = 0;
                for (int r = 0; r < rows; ++r) { double t = 0; for (int c = 0; c < cols; ++c) t += fabs((double)a[(size_t)r * cols + c]); mx = std::max(mx, t); }
                return mx;
            };
            double worst = 0;
            for (int l = 0; l < L; ++l) {
                const float* Wq = skip((size_t)I * H); const float* bq = skip(I);
                const float* Wkv = skip((size_t)2 * I * H); const float* bkv = skip(2 * I);
                const float* Wek = skip((size_t)I * H); skip(I);
                const float* Wo = skip((size_t)H * I); skip(H);
                const float* ln1g = skip(H); const float* ln1b = skip(H);
                skip(3 * H);
                const float* W1 = skip((size_t)F * H); const float* b1 = skip(F);
                const float* W2 = skip((size_t)H * F); skip(H);
                const float* ln2g = skip(H); const float* ln2b = skip(H);
                skip(3 * H);
                const double m1 = sqrt((double)H) * amax(ln1g, H) + amax(ln1b, H);
                const double m2 = sqrt((double)H) * amax(ln2g, H) + amax(ln2b, H);
                const double vb = rowl1(Wkv + (size_t)I * H, I, H) * m1 + amax(bkv + I, I);   // |v| (and their attention averages)
                const double hb = rowl1(W1, F, H) * m2 + amax(b1, F);                            // |h_pre| >= |GELU(h_pre)|
                worst = std::max(std::max(worst, m1), std::max(std::max(m2, vb), hb));
                // the weights themselves become fp16 pieces too (f32_to_f16_rn maps |w| >= 65520 to infinity)
                worst = std::max(worst, std::max(std::max(amax(Wq, (size_t)I * H), amax(Wkv, (size_t)2 * I * H)),
                                                 std::max(std::max(amax(Wek, (size_t)I * H), amax(Wo, (size_t)H * I)),
                                                          std::max(amax(W1, (size_t)F * H), amax(W2, (size_t)H * F)))));
                if (H == DFF_DH && L <= 3) {
                    // the k / v fold's kernel (round 6) splits two more activations as they are: q' = W_k,h^T (W_q,h n + b_q,h), an operand
                    // of the logits -- |q'| <= (L1 norm of a row of W_k,h^T W_q,h) max|n| + |W_k,h^T b_q,h| -- and the rows of
                    // G = dattn (W_o,h W_v,h), an operand of dA, whose input rows arrive scaled to a maximum below 32
                    double qb = 0, gb = 0;
                    for (int h = 0; h < DFF_HEADS; ++h)
                        for (int e = 0; e < DFF_DH; ++e) {
                            double l1 = 0, cqv = 0, g1 = 0;
                            for (int c = 0; c < H; ++c) {
                                double sq = 0, so = 0;
                                for (int dd = 0; dd < DFF_DH; ++dd) {
                                    sq += (double)Wkv[(size_t)(h * 64 + dd) * H + e] * Wq[(size_t)(h * 64 + dd) * H + c];
                                    so += (double)Wo[(size_t)c * I + h * 64 + dd] * Wkv[(size_t)(I + h * 64 + dd) * H + e];
                                }
                                l1 += fabs(sq); g1 += fabs(so);
                            }
                            for (int dd = 0; dd < DFF_DH; ++dd) cqv += (double)Wkv[(size_t)(h * 64 + dd) * H + e] * bq[h * 64 + dd];
                            qb = std::max(qb, l1 * m1 + fabs(cqv));
                            gb = std::max(gb, 32.0 * g1);
                        }
                    worst = std::max(worst, std::max(qb, gb));
                }
            }
            if (!(worst < 1.6e4)) {
                fprintf(stderr, "dff: forward activations of this model may reach %.3g (> 1.6e4): the fp16 split engine is off, weight GEMMs run on the fp32 matrix pipe\n", worst);
                m->split = false;
            }
        }
        // Two workgroups per protein (PAIR variants) need both blocks of a pair resident at once, and the launches are not cooperative
        // (ADVICE r05): never where the GPU is known to be shared -- more local ranks than visible devices (torchrun's
        // LOCAL_WORLD_SIZE) -- or where the caller says so (DFF_PAIR=0).  dff_debug_pair(m, 1) turns them back on.
        {
            const char* ep = getenv("DFF_PAIR");
            const char* lw = getenv("LOCAL_WORLD_SIZE");
            int ndev = 0;
            if (ep && ep[0] == '0') m->pair_off = true;
            else if (lw && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && atoi(lw) > ndev) {
                m->pair_off = true;
                static bool said = false;
                if (!said) { said = true; fprintf(stderr, "dff: %s local ranks share %d GPU(s): the two-workgroups-per-protein kernels are off\n", lw, ndev); }
            }
        }
        const char* ef = getenv("DFF_FOLD_KV");
        // (the <= 16-row FOLD kernel keeps a model's forward activations in LDS / registers: built for <= 3 layers, which is
        // every shipped configuration; a deeper hidden-64 model runs the unfolded kernels)
        m->fold_kv = H == DFF_DH && L <= 3 && !(ef && ef[0] == '0');
        const void* fn_; unsigned lds_; const char* nm_;
        m->small_split = m->split && N <= 10 && dff_small_pick(DFF_MODE_SCORE, H, 8, false, true, &fn_, &lds_, &nm_);
    }
    memset(&m->dev, 0, sizeof m->dev);
    m->dev.N = N; m->dev.H = H; m->dev.L = L; m->dev.T = cfg->timesteps;

    const float* p = w;
    auto take = [&](size_t n) { const float* r = p; p += n; return r; };
    const int intr = cfg->use_intrinsic_coords, dist = cfg->use_distances, ab = cfg->use_abs_coords;
    const int NI = N + 1 + 3 * ab;                 // node inputs: one-hot (N) | x (3, with abs coords) | t
    const int nfe_used = 3 * intr + dist;          // edge inputs: x_j - x_i (3) | |x_j - x_i|^2 ; neither -> one zero feature
    const int nfe = nfe_used ? nfe_used : 1;
    const float* Wn = take((size_t)H * NI);
    const float* bn = take(H);
    const float* We = take((size_t)H * nfe);
    const float* be = take(H);
    const int dec = cfg->conservative ? 1 : 3;
    const float* wd = take((size_t)dec * H);
    const float* bd = take(dec);
    {
        std::vector<float> WnT((size_t)NI * H);
        for (int c = 0; c < H; ++c)
            for (int i = 0; i < NI; ++i) WnT[(size_t)i * H + c] = Wn[(size_t)c * NI + i];
        m->dev.in_intr = intr; m->dev.in_dist = dist; m->dev.in_abs = ab;
        m->dev.wn_t = NI - 1;
        UP(WnT, m->dev.WnT);
        UP(std::vector<float>(bn, bn + H), m->dev.bn);
        UP(std::vector<float>(wd, wd + (size_t)dec * H), m->dev.wdec);
        m->dev.conservative = cfg->conservative;
        m->dev.bdec = bd[0];
        for (int c2 = 0; c2 < 3; ++c2) m->dev.bdec3[c2] = dec == 3 ? bd[c2] : 0.f;
    }
    for (int l = 0; l < L; ++l) {
        const float* Wq = take((size_t)I * H);  const float* bq = take(I);
        const float* Wkv = take((size_t)2 * I * H); const float* bkv = take(2 * I);
        const float* Wek = take((size_t)I * H); const float* bek = take(I);
        const float* Wo = take((size_t)H * I);  const float* bo = take(H);
        const float* ln1g = take(H); const float* ln1b = take(H);
        const float* g1 = take(3 * H);
        const float* W1 = take((size_t)F * H);  const float* b1 = take(F);
        const float* W2 = take((size_t)H * F);  const float* b2 = take(H);
        const float* ln2g = take(H); const float* ln2b = take(H);
        const float* g2 = take(3 * H);
        DffLayerDev& d = m->dev.layer[l];
        // ---- fold edge_embedding into edges_to_kv (float64): W_c (512x3), b_c (512) ----
        // W_c3 (512x3): the intrinsic-coordinate columns (zero without them); cd (512): the distance column
        std::vector<double> Wc((size_t)I * 3, 0.0), cd(I, 0.0), bc(I);
        for (int r = 0; r < I; ++r) {
            double s0 = 0, s1 = 0, s2 = 0, sd = 0, sb = 0;
            for (int c = 0; c < H; ++c) {
                const double e = Wek[(size_t)r * H + c];
                if (intr) { s0 += e * We[c * nfe + 0]; s1 += e * We[c * nfe + 1]; s2 += e * We[c * nfe + 2]; }
                if (dist) sd += e * We[c * nfe + 3 * intr];
                sb += e * be[c];
            }
            Wc[r * 3 + 0] = s0; Wc[r * 3 + 1] = s1; Wc[r * 3 + 2] = s2;
            cd[r] = sd;
            bc[r] = sb + bek[r];
        }
        // distance term: s_ih = c_h . q_ih -> W_s (8xH), b_s (8) ; D_ih feeds the output through w_od,h = W_o,h c_h (Hx8)
        std::vector<double> Wsd((size_t)8 * H, 0.0), bsd(8, 0.0), Wod((size_t)H * 8, 0.0);
        for (int h = 0; h < DFF_HEADS; ++h) {
            for (int c = 0; c < H; ++c) {
                double s = 0, t = 0;
                for (int dd = 0; dd < DFF_DH; ++dd) {
                    s += cd[h * 64 + dd] * Wq[(size_t)(h * 64 + dd) * H + c];
                    t += (double)Wo[(size_t)c * I + h * 64 + dd] * cd[h * 64 + dd];
                }
                Wsd[(size_t)h * H + c] = s;
                Wod[(size_t)c * 8 + h] = t;
            }
            double s = 0;
            for (int dd = 0; dd < DFF_DH; ++dd) s += cd[h * 64 + dd] * bq[h * 64 + dd];
            bsd[h] = s;
        }
        // W_u (24xH), b_u (24): u_ih = W_c,h^T q_ih ; W_oc (Hx24) = W_o,h W_c,h ; b_o' = b_o + W_o b_c
        std::vector<double> Wu((size_t)24 * H, 0.0), bu(24, 0.0), Woc((size_t)H * 24, 0.0), bof(H);
        for (int h = 0; h < DFF_HEADS; ++h)
            for (int cc = 0; cc < 3; ++cc) {
                for (int c = 0; c < H; ++c) {
                    double s = 0;
                    for (int dd = 0; dd < DFF_DH; ++dd) s += Wc[(h * 64 + dd) * 3 + cc] * Wq[(size_t)(h * 64 + dd) * H + c];
                    Wu[(size_t)(3 * h + cc) * H + c] = s;
                }
                double s = 0;
                for (int dd = 0; dd < DFF_DH; ++dd) s += Wc[(h * 64 + dd) * 3 + cc] * bq[h * 64 + dd];
                bu[3 * h + cc] = s;
                for (int c = 0; c < H; ++c) {
                    double t = 0;
                    for (int dd = 0; dd < DFF_DH; ++dd) t += (double)Wo[(size_t)c * I + h * 64 + dd] * Wc[(h * 64 + dd) * 3 + cc];
                    Woc[(size_t)c * 24 + 3 * h + cc] = t;
                }
            }
        for (int c = 0; c < H; ++c) {
            double s = bo[c];
            for (int r = 0; r < I; ++r) s += (double)Wo[(size_t)c * I + r] * bc[r];
            bof[c] = s;
        }
        // ---- H == head dimension: fold W_k into W_q and W_v into W_o (float64) ----
        // logits: q.k_j = (W_k,h^T q_i).n_j + [constant in j], so with q'_ih = W_k,h^T (W_q,h n_i + b_q,h) the keys are the
        // LayerNorm rows n_j themselves; values: W_o,h sum_j a_ij (W_v,h n_j + b_v,h) = (W_o,h W_v,h) sum_j a_ij n_j + W_o,h b_v,h.
        // The kernels keep their [q | u | k | v] layout: the k and v blocks of the images become identities (exactly
        // representable, so k = v = n bit for bit) and the <= 16-row kernel skips them altogether.  u, s, W_oc, W_od above
        // are built from the ORIGINAL q and W_o: the edge terms are untouched.
        std::vector<double> Mq, cq, Wof;
        if (m->fold_kv) {
            Mq.assign((size_t)I * H, 0.0); cq.assign(I, 0.0); Wof.assign((size_t)H * I, 0.0);
            for (int h = 0; h < DFF_HEADS; ++h)
                for (int e = 0; e < DFF_DH; ++e) {
                    for (int c = 0; c < H; ++c) {
                        double sq = 0;
                        for (int dd = 0; dd < DFF_DH; ++dd)
                            sq += (double)Wkv[(size_t)(h * 64 + dd) * H + e] * Wq[(size_t)(h * 64 + dd) * H + c];
                        Mq[(size_t)(h * 64 + e) * H + c] = sq;
                    }
                    double sc = 0;
                    for (int dd = 0; dd < DFF_DH; ++dd) sc += (double)Wkv[(size_t)(h * 64 + dd) * H + e] * bq[h * 64 + dd];
                    cq[h * 64 + e] = sc;
                    for (int c = 0; c < H; ++c) {
                        double so = 0;
                        for (int dd = 0; dd < DFF_DH; ++dd)
                            so += (double)Wo[(size_t)c * I + h * 64 + dd] * Wkv[(size_t)(I + h * 64 + dd) * H + e];
                        Wof[(size_t)c * I + h * 64 + e] = so;
                    }
                }
            for (int c = 0; c < H; ++c) {
                double sbv = 0;
                for (int r = 0; r < I; ++r) sbv += (double)Wo[(size_t)c * I + r] * bkv[I + r];
                bof[c] += sbv;
            }
        }
        const bool fold = m->fold_kv;
        std::vector<float> bo32(H);
        for (int c = 0; c < H; ++c) bo32[c] = (float)bof[c];

        UP(std::vector<float>(ln1g, ln1g + H), d.ln1_g); UP(std::vector<float>(ln1b, ln1b + H), d.ln1_b);
        UP(bo32, d.bo);
        UP(std::vector<float>(g1, g1 + 3 * H), d.g1);
        UP(std::vector<float>(ln2g, ln2g + H), d.ln2_g); UP(std::vector<float>(ln2b, ln2b + H), d.ln2_b);
        UP(pack_b(H, F, [&](int k, int n) { return (double)W1[(size_t)n * H + k]; }), d.W1_p);
        UP(std::vector<float>(b1, b1 + F), d.b1);
        UP(pack_b(F, H, [&](int k, int n) { return (double)W2[(size_t)n * F + k]; }), d.W2_p);
        UP(std::vector<float>(b2, b2 + H), d.b2);
        UP(std::vector<float>(g2, g2 + 3 * H), d.g2);
        // transposed orientation for the VJP
        UP(pack_b(H, F, [&](int k, int n) { return (double)W2[(size_t)k * F + n]; }), d.W2T_p);
        UP(pack_b(F, H, [&](int k, int n) { return (double)W1[(size_t)k * H + n]; }), d.W1T_p);
        // extended-head images: column e of head h: [0,64) q, [64,67) u, [67,80) 0, [80,144) k, [144,208) v
        auto wqkvx = [&](int col, int c) -> double {
            const int h = col / 208, e = col % 208;
            if (e < 64) return fold ? Mq[(size_t)(h * 64 + e) * H + c] : (double)Wq[(size_t)(h * 64 + e) * H + c];
            if (e < 80) return e < 67 ? Wu[(size_t)(3 * h + e - 64) * H + c] : e == 67 ? Wsd[(size_t)h * H + c] : 0.0;
            if (fold) return (e < 144 ? e - 80 : e - 144) == c ? 1.0 : 0.0;
            if (e < 144) return Wkv[(size_t)(h * 64 + e - 80) * H + c];
            return Wkv[(size_t)(I + h * 64 + e - 144) * H + c];
        };
        std::vector<float> bqkvx(8 * 208, 0.f);
        for (int col = 0; col < 8 * 208; ++col) {
            const int h = col / 208, e = col % 208;
            bqkvx[col] = e < 64 ? (fold ? (float)cq[h * 64 + e] : bq[h * 64 + e]) : e < 67 ? (float)bu[3 * h + e - 64] : e == 67 ? (float)bsd[h]
                         : e < 80 ? 0.f : fold ? 0.f : e < 144 ? bkv[h * 64 + e - 80] : bkv[I + h * 64 + e - 144];
        }
        // output projection with the xrel rows: row e of head h: [0,64) W_o[:, h*64+e], [64,67) W_oc[:, 3h+e-64]
        auto wox = [&](int krow, int c) -> double {
            const int h = krow / 80, e = krow % 80;
            if (e < 64) return fold ? Wof[(size_t)c * I + h * 64 + e] : (double)Wo[(size_t)c * I + h * 64 + e];
            return e < 67 ? Woc[(size_t)c * 24 + 3 * h + e - 64] : e == 67 ? Wod[(size_t)c * 8 + h] : 0.0;
        };
        UP(pack_b(H, 8 * 208, [&](int k, int n) { return wqkvx(n, k); }), d.Wqkvx_p);
        d.Wqkvx_s = d.W1_s = d.W2T_s = d.WoxT_s = d.W2_s = d.W1T_s = d.Wox_s = d.WqkvxT_s = nullptr;
        if (m->split) {
            int rc_;
            // image format per GEMM group: two fp16 pieces where the kernels' engine takes them (dff_fused_f16_mask)
            const int f16m = dff_fused_f16_mask();
            auto PB = [&](int bit, int K_, int Nout_, const std::function<double(int, int)>& wf) {
                return (f16m & bit) ? pack_b_split_f16(K_, Nout_, wf) : pack_b_split(K_, Nout_, wf);
            };
            if ((rc_ = upload_u32(m, PB(1, H, 8 * 208, [&](int k, int n) { return wqkvx(n, k); }), &d.Wqkvx_s))) return rc_;
            if ((rc_ = upload_u32(m, PB(1, H, F, [&](int k, int n) { return (double)W1[(size_t)n * H + k]; }), &d.W1_s))) return rc_;
            if ((rc_ = upload_u32(m, PB(2, H, F, [&](int k, int n) { return (double)W2[(size_t)k * F + n]; }), &d.W2T_s))) return rc_;
            if ((rc_ = upload_u32(m, PB(4, H, 8 * 80, [&](int k, int n) { return wox(n, k); }), &d.WoxT_s))) return rc_;
            if ((rc_ = upload_u32(m, PB(1, F, H, [&](int k, int n) { return (double)W2[(size_t)n * F + k]; }), &d.W2_s))) return rc_;
            if ((rc_ = upload_u32(m, PB(2, F, H, [&](int k, int n) { return (double)W1[(size_t)k * H + n]; }), &d.W1T_s))) return rc_;
            // the 64 regular rows of every head of [W_o ; W_oc] (the extension rows stay on the fp32 image)
            if ((rc_ = upload_u32(m, PB(1, 8 * 64, H, [&](int k, int n) { return wox((k / 64) * 80 + k % 64, n); }), &d.Wox_s))) return rc_;
            // ... and the 192 regular rows [q | k | v] of every head of QKV_ext^T
            if ((rc_ = upload_u32(m, PB(8, 8 * 192, H, [&](int c, int n) {
                     const int h = c / 192, cc = c % 192;
                     return wqkvx(h * 208 + (cc < 64 ? cc : cc + 16), n); }), &d.WqkvxT_s))) return rc_;
        }
        d.Wqkvx_w = d.W1_w = d.W2T_w = d.WoxT_w = d.Wox_t = d.W2_t = d.W1T_t = d.WqkvxT_t = nullptr;
        if (m->small_split) {
            int rc_;
            // fold_kv on the shipped input branch: the <= 16-row kernel's FOLD variant streams [q' | u] (5 tiles per head) and
            // back-projects dQ only (2 k-blocks per head); k = v = LayerNorm output never goes through a GEMM there
            const bool sfold = m->fold_kv && intr && !dist && !ab;
            // the FOLD variant's engine decides the image format: two fp16 pieces (round 5) or three bf16 pieces
            const bool f16img = dff_small_f16_level() >= 2 || (sfold && dff_small_f16_level() == 1);
            auto PU = [&](const std::vector<std::pair<int, int>>& un, int nout, const std::function<double(int, int)>& wf) {
                return f16img ? pack_units_f16(un, nout, wf) : pack_units(un, nout, wf);
            };
            if (sfold) {
                if ((rc_ = upload_u32(m, PU(units_wide(H, 8 * 80), 8 * 80, [&](int k, int n) { return wqkvx((n / 80) * 208 + n % 80, k); }), &d.Wqkvx_w))) return rc_;
            } else
            if ((rc_ = upload_u32(m, PU(units_wide(H, 8 * 208), 8 * 208, [&](int k, int n) { return wqkvx(n, k); }), &d.Wqkvx_w))) return rc_;
            if ((rc_ = upload_u32(m, PU(units_wide(H, F), F, [&](int k, int n) { return (double)W1[(size_t)n * H + k]; }), &d.W1_w))) return rc_;
            if ((rc_ = upload_u32(m, PU(units_wide(H, F), F, [&](int k, int n) { return (double)W2[(size_t)k * F + n]; }), &d.W2T_w))) return rc_;
            if ((rc_ = upload_u32(m, PU(units_wide(H, 8 * 80), 8 * 80, [&](int k, int n) { return wox(n, k); }), &d.WoxT_w))) return rc_;
            // Nout = H: the 64 regular rows of every head of [W_o ; W_oc] / [q | k | v] (extension rows: fp32 k-step off the fp32 images)
            if ((rc_ = upload_u32(m, PU(units_tall(8 * 64, H), H, [&](int c, int n) { return wox((c / 64) * 80 + c % 64, n); }), &d.Wox_t))) return rc_;
            if ((rc_ = upload_u32(m, PU(units_tall(F, H), H, [&](int k, int n) { return (double)W2[(size_t)n * F + k]; }), &d.W2_t))) return rc_;
            if ((rc_ = upload_u32(m, PU(units_tall(F, H), H, [&](int k, int n) { return (double)W1[(size_t)k * H + n]; }), &d.W1T_t))) return rc_;
            if (sfold) {
                if ((rc_ = upload_u32(m, PU(units_tall(8 * 64, H), H, [&](int c, int n) { return wqkvx((c / 64) * 208 + c % 64, n); }), &d.WqkvxT_t))) return rc_;
            } else
            if ((rc_ = upload_u32(m, PU(units_tall(8 * 192, H), H, [&](int c, int n) {
                     const int h = c / 192, cc = c % 192;
                     return wqkvx(h * 208 + (cc < 64 ? cc : cc + 16), n); }), &d.WqkvxT_t))) return rc_;
        }
        UP(bqkvx, d.bqkvx);
        UP(pack_b(8 * 80, H, [&](int k, int n) { return wox(k, n); }, 5), d.Wox_p);
        UP(pack_b(H, 8 * 80, [&](int k, int n) { return wox(n, k); }), d.WoxT_p);
        UP(pack_b(8 * 208, H, [&](int k, int n) { return wqkvx(k, n); }, 13), d.WqkvxT_p);
    }
    build_schedule(cfg->timesteps, m->sched);
    UP(m->sched[6], m->dev.sqrt_recip_ac);
    UP(m->sched[7], m->dev.sqrt_recipm1_ac);
    UP(m->sched[10], m->dev.post_c1);
    UP(m->sched[11], m->dev.post_c2);
    UP(m->sched[9], m->dev.post_logvar);
    undo.m = nullptr;
    *out = m;
    return DFF_OK;
}

extern "C" void dff_model_destroy(dff_model* m) {
    if (!m) return;
    DeviceGuard g(m->device);
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->stash) (void)hipFree(m->stash);
    if (m->xchg) (void)hipFree(m->xchg);
    if (m->xflag) (void)hipFree(m->xflag);
    if (m->prof) (void)hipFree(m->prof);
    for (auto& k : m->l0) for (auto& t : k) if (t.tab) (void)hipFree(t.tab);
    delete m;
}

extern "C" int dff_schedule(const dff_model* m, int which, float* out_host) {
    if (!m || !out_host || which < 0 || which >= 12) return fail(DFF_EINVAL, "bad schedule request");
    memcpy(out_host, m->sched[which].data(), m->sched[which].size() * sizeof(float));
    return DFF_OK;
}

extern "C" int dff_set_group(dff_model* m, int g) {
    if (!m || g < 0) return fail(DFF_EINVAL, "bad group");
    m->group_override = g;
    return DFF_OK;
}

extern "C" int dff_debug_force_generic(dff_model* m, int on) {
    if (!m) return fail(DFF_EINVAL, "null model");
    m->force_generic = on != 0;
    return DFF_OK;
}

extern "C" int dff_debug_max_workgroups(dff_model* m, int n) {
    if (!m || n < 1) return fail(DFF_EINVAL, "bad workgroup limit");
    m->max_wgs = n;
    return DFF_OK;
}

static int clear_status(dff_model* m);
extern "C" int dff_debug_pair(dff_model* m, int on) {
    if (!m) return fail(DFF_EINVAL, "null model");
    m->pair_off = on == 0;
    m->pair_slow = on == 2 ? 1 : on == 3 ? 2 : 0;
    // turning the PAIR variants off is how a caller recovers from a partner timeout: the one-workgroup kernels do not look
    // at the word, so it is cleared here (it would otherwise fail every later status check of a model that works again)
    // (unconditionally -- ADVICE r04: the DEVICE word may be set before the host has looked at it; clear_status synchronises,
    // which a debug / recovery call may do)
    if (m->pair_off && m->xflag) return clear_status(m);
    return DFF_OK;
}

// Sticky status of the model's launches so far: bit 0 = a PAIR launch (two workgroups per protein) gave up waiting for a
// partner workgroup (bounded spin instead of a hang; the results of that launch are invalid).  The word lives at
// xflag[0], is only ever OR-ed by the kernels and is never cleared by a launch, so an error in ANY launch of a chunked
// run is still there when the caller looks.  Synchronises the device -- which is why only the status calls come here: the
// launch path never does (a PAIR kernel that finds the word set at entry leaves at once, and launch() refuses PAIR on the
// host's cached copy), so dff_score / dff_langevin_run / dff_ddpm_run stay asynchronous on the caller's stream.
static int read_status(dff_model* m, unsigned* word, bool clear) {
    *word = 0;
    if (!m->xflag) return DFF_OK;
    ON_DEVICE(m->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(word, m->xflag, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (clear && *word) HIPCHK(hipMemset(m->xflag, 0, sizeof(unsigned)));
    m->sticky = clear ? 0u : *word;
    return DFF_OK;
}
static int clear_status(dff_model* m) {
    unsigned w = 0;
    return read_status(m, &w, true);
}

extern "C" int dff_model_status(dff_model* m, unsigned* status) {
    if (!m || !status) return fail(DFF_EINVAL, "null argument");
    return read_status(m, status, false);
}

extern "C" int dff_model_status_clear(dff_model* m) {
    if (!m) return fail(DFF_EINVAL, "null model");
    return clear_status(m);
}

// tests: set the DEVICE word behind the host's back (as a kernel that lost its partner would), host cache untouched
extern "C" int dff_debug_poke_status(dff_model* m, unsigned word) {
    if (!m) return fail(DFF_EINVAL, "null model");
    if (!m->xflag) return fail(DFF_EINVAL, "no two-workgroups launch has run on this model yet");
    ON_DEVICE(m->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(m->xflag, &word, sizeof(unsigned), hipMemcpyHostToDevice));
    return DFF_OK;
}

// debugging form: reads AND clears
extern "C" int dff_debug_pair_status(dff_model* m, int* status) {
    if (!m || !status) return fail(DFF_EINVAL, "null argument");
    unsigned w = 0;
    const int rc = read_status(m, &w, true);
    *status = (int)w;
    return rc;
}

extern "C" int dff_debug_l0_table(dff_model* m, int on) {
    if (!m) return fail(DFF_EINVAL, "null model");
    m->l0_off = on == 0;
    return DFF_OK;
}

extern "C" int dff_debug_small_waves(dff_model* m, int waves) {
    if (!m || !(waves == 0 || waves == 4 || waves == 8)) return fail(DFF_EINVAL, "waves must be 0 (auto), 4 or 8");
    m->small_waves = waves;
    return DFF_OK;
}

extern "C" int dff_last_launch(const dff_model* m, const char** name, int* grid, int* lds) {
    if (!m) return fail(DFF_EINVAL, "null model");
    if (name) *name = m->last_kernel;
    if (grid) *grid = m->last_grid;
    if (lds) *lds = m->last_lds;
    return DFF_OK;
}

static int ensure_stash(dff_model* m, size_t need) {
    if (need > m->stash_floats) {
        if (m->stash) HIPCHK(hipFree(m->stash));
        m->stash = nullptr; m->stash_floats = 0;
        HIPCHK(hipMalloc((void**)&m->stash, need * sizeof(float)));
        m->stash_floats = need;
    }
    return DFF_OK;
}

// rows <= 16: one-head-per-wave kernel (dff_small.hip)
static int ensure_pair_buffers(dff_model* m, int npairs, size_t need, hipStream_t stream);
// pair: two workgroups per protein (dff_small_kernel<..., PAIR>: blocks b and b + 8 share a protein, 16 blocks per 8 proteins, ONE
// launch -- the caller has checked that every block gets a CU of its own)
static bool small_pair_available(const dff_model* m, int mode, int G) {
    const int N = m->cfg.n_beads, H = m->cfg.hidden;
    const void* fn; unsigned lds; const char* name;
    const bool gen = !(m->cfg.use_intrinsic_coords == 1 && m->cfg.use_distances == 0 && m->cfg.use_abs_coords == 0);
    // OPT-IN (DFF_SMALL_PAIR=1, read per launch): built and measured in round 6 -- chignolin at 128 / 100 / 32 per GPU 45.8 / 45.9 / 45.4 us
    // per step against 44.4 with one workgroup per protein.  The wave-private blocks do shrink (-19 k cycles per step: four waves,
    // one per SIMD), but each of the 12 exchanges costs ~1.8 k cycles inside its row stage -- a store's s_waitcnt vmcnt(0) also
    // waits for the weight units the wave has just requested for the next block (loads and stores retire through one in-order
    // counter), 0.25 - 0.45 us in tools_ubench/pair_exchange.hip without that -- and the 16-lanes-per-row stages issue more per thread
    // (profiles/r06/pair_small/).  Kept as a tested variant; not the default.
    const char* const en = getenv("DFF_SMALL_PAIR");
    if (!(en && en[0] == '1')) return false;
    return H == 64 && G * N <= 10 && m->small_waves == 0 && m->small_split && m->fold_kv && !gen && m->cfg.conservative &&
           dff_small_pick(mode, H, 8, false, true, &fn, &lds, &name, true, true);
}
static int launch_small(dff_model* m, DffRunArgs& a, int G, hipStream_t stream, bool pair = false) {
    const int N = m->cfg.n_beads, H = m->cfg.hidden, L = m->cfg.n_layers;
    const void* fn; unsigned lds; const char* name; int nthreads = 256;
    // 8 waves (two per SIMD, one head per wave) when the rows fit its 11-row head buffers
    const bool eight = (H == 64 || H == 96) && G * N <= 10 && m->small_waves != 4;
    const bool gen = !(m->cfg.use_intrinsic_coords == 1 && m->cfg.use_distances == 0 && m->cfg.use_abs_coords == 0);
    const int NW = eight ? 8 : 4;
    const bool spw = eight && m->small_split;
    if (!dff_small_pick(a.mode, H, NW, gen, spw, &fn, &lds, &name, m->fold_kv, pair))
        return fail(DFF_EINVAL, "no <= 16-row kernel for hidden=%d waves=%d in this build", H, NW);
    nthreads = pair ? 256 : NW * 64;
    lds *= (unsigned)sizeof(float);
    if (lds > 160 * 1024) return fail(DFF_EINVAL, "LDS budget exceeded (%u bytes)", lds);
    const int ngroups = (a.B + G - 1) / G;
    const int npairs = pair ? 8 * ((ngroups + 7) / 8) : 0;
    const int grid_all = pair ? 2 * npairs : ngroups;
    const int grid_max = pair ? grid_all : (grid_all < m->max_wgs ? grid_all : m->max_wgs);
    const SmallStash sl = dff_small_stash(N, G, H, L);
    int rc = ensure_stash(m, (size_t)grid_max * sl.total);
    if (rc) return rc;
    a.xchg = nullptr; a.xflag = nullptr; a.xpairs = npairs; a.xslow = m->pair_slow;
    if (pair) {
        rc = ensure_pair_buffers(m, npairs, (size_t)npairs * 4 * 1024, stream);   // [pair][half][parity][256 threads x 4 floats]
        if (rc) return rc;
        a.xchg = m->xchg; a.xflag = m->xflag;
    }
    m->last_pair = pair;
    a.G = G;
    a.prof = m->prof_on ? m->prof : nullptr;
    a.prof_wave = m->prof_wave < NW ? m->prof_wave : 0;
    a.stash = m->stash;
    a.stash_stride = sl.total;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // independent proteins: a batch larger than max_wgs workgroups runs as consecutive launches (same stream,
    // same stash), each over the whole step range
    for (int w0 = 0; w0 < grid_all; w0 += grid_max) {
        const int grid = grid_all - w0 < grid_max ? grid_all - w0 : grid_max;
        a.b_base = pair ? 0 : w0 * G;
        void* args[] = {(void*)&m->dev, (void*)&a};
        HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(nthreads), args, lds, stream));
        m->last_grid = grid; m->last_base = a.b_base;
        a.prof = nullptr;   // per-stage cycles: first launch only
    }
    m->last_kernel = name; m->last_lds = (int)lds; m->last_G = G; m->last_B = a.B;
    m->last_stride = sl.total; m->last_small = true;
    return DFF_OK;
}

// PAIR launches (either kernel): the exchange slots (`need` floats) and the flag words -- error word, [pair][half] sequence flags,
// [pair][half] XCD ids; the sequence numbers restart at every launch, the error word at [0] is NOT touched
static int ensure_pair_buffers(dff_model* m, int npairs, size_t need, hipStream_t stream) {
    const size_t nflag = (size_t)4 * npairs + 1;
    if (need > m->xchg_floats) {
        if (m->xchg) HIPCHK(hipFree(m->xchg));
        m->xchg = nullptr; m->xchg_floats = 0;
        HIPCHK(hipMalloc((void**)&m->xchg, need * sizeof(float)));
        m->xchg_floats = need;
    }
    if (!m->xflag) {   // once per model, sized for the largest grid a PAIR launch can have (one block per CU): no
                       // re-allocation -- and so no synchronisation -- on the launch path afterwards
        const size_t cap = 2 * (size_t)(m->n_cus > 256 ? m->n_cus : 256) + 1;   // 4 words per pair, at most n_cus / 2 pairs
        HIPCHK(hipMalloc((void**)&m->xflag, cap * sizeof(unsigned)));
        HIPCHK(hipMemsetAsync(m->xflag, 0, sizeof(unsigned), stream));
        m->xflag_n = cap;
    }
    if (nflag > m->xflag_n) return fail(DFF_EINVAL, "PAIR grid of %d pairs exceeds the flag array", npairs);
    HIPCHK(hipMemsetAsync(m->xflag + 1, 0, (nflag - 1) * sizeof(unsigned), stream));
    return DFF_OK;
}

// generic kernel (rows <= 64), variant already chosen
static int launch_generic(dff_model* m, DffRunArgs& a, int G, const Variant* v, hipStream_t stream) {
    const int N = m->cfg.n_beads, H = m->cfg.hidden, L = m->cfg.n_layers;
    const unsigned lds = v->lds_floats(N, G) * (unsigned)sizeof(float);
    if (lds > 160 * 1024) return fail(DFF_EINVAL, "LDS budget exceeded (%u bytes) for N=%d G=%d H=%d", lds, N, G, H);
    // PAIR variants: blocks b and b + 8 share a protein (dff_kernels.hip), 16 blocks per 8 proteins, ONE launch: the caller
    // has checked that every block gets a CU of its own (both blocks of a pair must be resident at once)
    const int ngroups = (a.B + G - 1) / G;   // (a pair shares a GROUP of G proteins: ala2 two or three to a 16-row tile)
    const int npairs = v->pair ? 8 * ((ngroups + 7) / 8) : 0;
    const int grid_all = v->pair ? 2 * npairs : ngroups;
    const int grid_max = v->pair ? grid_all : (grid_all < m->max_wgs ? grid_all : m->max_wgs);
    const StashLayout sl = dff_stash_layout(N, G, H, L, v->MT);
    { int rc = ensure_stash(m, (size_t)grid_max * sl.total); if (rc) return rc; }
    a.xchg = nullptr; a.xflag = nullptr; a.xpairs = npairs; a.xslow = m->pair_slow;
    if (v->pair) {
        const int rc = ensure_pair_buffers(m, npairs, (size_t)npairs * 4 * (size_t)(G * N) * (H + 4), stream);
        if (rc) return rc;
        a.xchg = m->xchg; a.xflag = m->xflag;
    }
    m->last_small = false;
    m->last_pair = v->pair;
    m->last_mt = v->MT;
    a.G = G;
    a.prof = m->prof_on ? m->prof : nullptr;
    a.stash = m->stash;
    a.stash_stride = sl.total;
    HIPCHK(hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int w0 = 0; w0 < grid_all; w0 += grid_max) {   // see launch_small
        const int grid = grid_all - w0 < grid_max ? grid_all - w0 : grid_max;
        a.b_base = w0 * G;
        void* args[] = {(void*)&m->dev, (void*)&a};
        HIPCHK(hipLaunchKernel(v->fn, dim3(grid), dim3(DFF_NTHREADS), args, lds, stream));
        m->last_grid = grid; m->last_base = a.b_base;
        a.prof = nullptr;
    }
    m->last_kernel = v->name; m->last_lds = (int)lds; m->last_G = G; m->last_B = a.B;
    m->last_stride = sl.total;
    return DFF_OK;
}

// Layer-0 inputs do not depend on x (node features are [one-hot, t]): nodes_in and the
// [q|u|k|v] rows of layer 0 are functions of the noise level only.  They are computed ONCE per noise
// level by running the sampling kernel itself in score mode (one workgroup per level, x = 0) and
// copying the layer-0 slot of each workgroup's stash into a table, which the sampling loops then read
// instead of running layer 0's QKV GEMM: bit-identical values (same code produced them), one shared
// L2-resident copy instead of one per workgroup.  Langevin: one entry (its fixed t); DDPM: T entries,
// built in chunks so that the stash does not grow beyond what sampling needs anyway.
// v == nullptr: rows<=16 kernel, else that variant of the generic kernel.
static int ensure_l0_table(dff_model* m, int kind, float t_norm, int G, const Variant* v, hipStream_t stream, const float** out) {
    const int N = m->cfg.n_beads, H = m->cfg.hidden, L = m->cfg.n_layers, T = m->cfg.timesteps;
    dff_model::L0Table* slot = nullptr;
    for (auto& c : m->l0[kind - 1]) {
        if (c.valid && c.G == G && c.waves == m->small_waves && c.variant == (const void*)v && c.split == m->small_split &&
            (kind == 2 || c.tnorm == t_norm)) {
            c.used = ++m->l0_clock;
            *out = c.tab;
            return DFF_OK;
        }
        if (!slot || (!c.valid && slot->valid) || (c.valid == slot->valid && c.used < slot->used)) slot = &c;
    }
    dff_model::L0Table& tb = *slot;
    *out = nullptr;
    const int nent = kind == 2 ? T : 1;
    size_t layer_stride, total;
    if (v) { const StashLayout sl = dff_stash_layout(N, G, H, L, v->MT); layer_stride = sl.layer_stride; total = sl.total; }
    else   { const SmallStash sl = dff_small_stash(N, G, H, L); layer_stride = sl.layer_stride; total = sl.total; }
    const size_t need = (size_t)nent * layer_stride;
    tb.valid = false;   // invalid until filled
    if (need > tb.floats) {
        if (tb.tab) HIPCHK(hipFree(tb.tab));
        tb.tab = nullptr; tb.floats = 0;
        HIPCHK(hipMalloc((void**)&tb.tab, need * sizeof(float)));
        tb.floats = need;
    }
    int CH = nent < 256 ? nent : 256;         // noise levels (= workgroups) per build launch ...
    if (CH > m->max_wgs) CH = m->max_wgs;     // ... each of which must be ONE kernel launch (its stash is copied out)
    const int Bmax = CH * G;
    struct Tmp {   // freed on every path
        float *xz = nullptr, *tnd = nullptr, *fo = nullptr;
        ~Tmp() { (void)hipFree(xz); (void)hipFree(fo); (void)hipFree(tnd); }
    } tmp;
    HIPCHK(hipMalloc((void**)&tmp.xz, (size_t)Bmax * N * 3 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&tmp.fo, (size_t)Bmax * N * 3 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&tmp.tnd, (size_t)Bmax * sizeof(float)));
    HIPCHK(hipMemsetAsync(tmp.xz, 0, (size_t)Bmax * N * 3 * sizeof(float), stream));
    std::vector<float> tn((size_t)Bmax);
    for (int e0 = 0; e0 < nent; e0 += CH) {
        const int ne = nent - e0 < CH ? nent - e0 : CH;
        for (int e = 0; e < ne; ++e)
            for (int g = 0; g < G; ++g)
                tn[(size_t)e * G + g] = kind == 2 ? (1.0f * (float)(e0 + e)) / (float)T : t_norm;   // as the kernel forms t/T
        HIPCHK(hipMemcpyAsync(tmp.tnd, tn.data(), (size_t)ne * G * sizeof(float), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));   // tn is reused by the next chunk
        DffRunArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = DFF_MODE_SCORE; a.B = ne * G; a.n_steps = 1; a.save_interval = 1;
        a.x_in = tmp.xz; a.tnorm = tmp.tnd; a.force_out = tmp.fo;
        const int rc = v ? launch_generic(m, a, G, v, stream) : launch_small(m, a, G, stream);
        if (rc) return rc;
        // only the layer-0 slot of each workgroup's stash is kept
        HIPCHK(hipMemcpy2DAsync(tb.tab + (size_t)e0 * layer_stride, layer_stride * sizeof(float), m->stash,
                                total * sizeof(float), layer_stride * sizeof(float), (size_t)ne,
                                hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    tb.valid = true; tb.G = G; tb.waves = m->small_waves; tb.tnorm = t_norm; tb.variant = (const void*)v; tb.split = m->small_split;
    tb.used = ++m->l0_clock;
    *out = tb.tab;
    return DFF_OK;
}

// choose proteins-per-workgroup and the kernel variant, make sure scratch is large enough, launch
static int launch(dff_model* m, DffRunArgs& a, hipStream_t stream) {
    const int N = m->cfg.n_beads, H = m->cfg.hidden, L = m->cfg.n_layers;
    ON_DEVICE(m->device);
    const int mt_min = (N + 15) / 16;
    // Proteins per workgroup, by measurement (profiles/r02/packing.jsonl).  Packing only ever pays inside the padded rows a
    // protein occupies anyway: ala2 (5 beads) fits three times into its 16-row tile, and a workgroup takes the same time for
    // one as for three, so as soon as the batch exceeds one workgroup per CU (256) three per workgroup win (P = 512:
    // 120 vs 179 us / step).  Spilling into a second row tile never pays: chignolin three to a 32-row tile of the generic
    // kernel runs at 0.20-0.26 of the fp32 roof against 0.40 for one per workgroup on the <= 16-row kernel, at every batch.
    int G = m->group_override;
    const bool auto_group = G <= 0;
    const int cap = (16 * mt_min) / N;  // proteins that fit the padded rows anyway
    if (auto_group) {
        G = 1;
        if (cap > 1 && a.B > 256) G = cap;
    }
    if (G > 16) G = 16;
    int mt = (G * N + 15) / 16;
    if (mt > 4) { G = 64 / N; mt = (G * N + 15) / 16; }
    // input branches other than the shipped one run the GEN variants of the generic kernel; with absolute
    // coordinates layer 0 depends on x, so there is no layer-0 table
    const bool gen = !(m->cfg.use_intrinsic_coords == 1 && m->cfg.use_distances == 0 && m->cfg.use_abs_coords == 0);
    const bool want_tab = a.mode != DFF_MODE_SCORE && !m->l0_off && !m->cfg.use_abs_coords;
    a.l0_tab = nullptr;
    const void* sfn_; unsigned slds_; const char* snm_;
    const bool have_small = dff_small_pick(DFF_MODE_SCORE, H, 4, gen, false, &sfn_, &slds_, &snm_) ||
                            dff_small_pick(DFF_MODE_SCORE, H, 8, gen, false, &sfn_, &slds_, &snm_);   // (hidden = 256: generic kernel only)
    // Hidden 96 (ala2) and 128: the <= 16-row kernel has no split variant for them (its split engine is written for hidden 64's unit
    // counts) and is bound by the fp32 matrix pipe; the one-row-tile split_f16 variant of the <= 64-row kernel is faster
    // (round 5: 81 vs 86 us / step at 256 per GPU, 89 vs 138 at 768 -- three proteins per workgroup --, 60 vs 85 at 128, where it
    // runs as two workgroups per protein; hidden 128 at 5 / 10 / 16 rows: 90 / 150 / 157 vs 184 / 300 / 307).  DFF_SMALL_H96=1 keeps the
    // <= 16-row kernel.
    static const bool small_h96 = [] { const char* e = getenv("DFF_SMALL_H96"); return e && e[0] == '1'; }();
    const bool prefer_generic = (H == 96 || H == 128) && m->split && !small_h96 && m->small_waves == 0;   // (dff_debug_small_waves(m, 4 | 8) asks for the <= 16-row kernel)
    if (G * N <= 16 && !m->force_generic && have_small && !prefer_generic) {
        if (want_tab) {
            int rc = ensure_l0_table(m, a.mode == DFF_MODE_DDPM ? 2 : 1, a.t_norm, G, nullptr, stream, &a.l0_tab);
            if (rc) return rc;
        }
        // Two workgroups per protein when one per protein would leave at least half the CUs idle (round 6: the reference's published
        // protocol runs 100 trajectories, evaluate/sampling_commands.md:13; config 2 over 8 GPUs is 32 per GPU) -- same conditions as
        // the <= 64-row PAIR variants below: the whole grid resident one block per CU, no failure word seen, not switched off.
        const int cu_cap_s = m->n_cus < m->max_wgs ? m->n_cus : m->max_wgs;
        const int ngr_s = (a.B + G - 1) / G;
        const bool spair = a.mode != DFF_MODE_SCORE && !m->pair_off && !m->sticky && 2 * 8 * ((ngr_s + 7) / 8) <= cu_cap_s &&
                           small_pair_available(m, a.mode, G);
        return launch_small(m, a, G, stream, spair);
    }
    const Variant* v = nullptr;
    auto pick = [&](int mt_) {
        const Variant* r = nullptr;
        int nv = 0;
        const Variant* vs = dff_fused_variants(&nv);
        for (int q = 0; q < nv; ++q) {
            const Variant& c = vs[q];
            // (a variant must also fit the 160 KB of LDS at this row count: e.g. the four-row-tile split variant does up to
            // 56 rows -- protein G --, beyond that the fp32 engine of the same shape runs; a shape that does not fit at all
            // -- hidden 128 at 32 rows in two row tiles -- hands over to the next larger one below)
            if (c.H == H && c.MT == mt_ && c.gen == gen && !c.pair && (!c.spw || m->split) && (!r || c.spw) &&
                c.lds_floats(N, G) * sizeof(float) <= 160u * 1024u) r = &c;
        }
        return r;
    };
    auto pick_pair = [&](int mt_, bool spw_) -> const Variant* {
        int nv = 0;
        const Variant* vs = dff_fused_variants(&nv);
        for (int q = 0; q < nv; ++q)
            if (vs[q].pair && vs[q].H == H && vs[q].MT == mt_ && vs[q].spw == spw_) return &vs[q];
        return nullptr;
    };
    auto pick_up = [&](int mt0) -> const Variant* {   // the smallest shape from mt0 row tiles up that exists and fits
        for (int mt_ = mt0; mt_ <= 4; ++mt_)
            if (const Variant* r = pick(mt_)) { mt = mt_; return r; }
        return nullptr;
    };
    v = pick_up(mt);
    if (!v) {  // fall back to one protein per workgroup
        G = 1;
        v = pick_up(mt_min);
    }
    if (!v) return fail(DFF_EINVAL, "no kernel variant for hidden=%d rows=%d", H, G * N);
    // Two workgroups per protein when one per protein would leave at least half the CUs idle (protein G at 128 per GPU:
    // 1087 -> ~700 us / step).  Both blocks of a pair must be RESIDENT at once, and the launch is not cooperative: the
    // variant is used only when the whole grid fits the device's CUs one block each (hipDeviceAttributeMultiprocessorCount
    // of THIS device -- a partitioned or CU-masked GPU reports fewer than 256 and gets the one-workgroup variant), and
    // only on the conservative, shipped input branch.  A previous PAIR launch that gave up on a partner (co-tenancy: another
    // process held the CUs) is reported here, before anything else is launched on top of its garbage.  (The layer-0 table
    // below is built by the one-workgroup variant: the stash layout is the same.)
    // Proteins that share a row tile anyway (ala2: up to three in 16 rows) are paired as a GROUP: the smallest group size
    // whose pairs fit the CUs -- ala2 at 256 per GPU runs as 128 groups of two on 256 workgroups instead of 256 single
    // proteins on 256 workgroups that each do all the heads.
    const int cu_cap = m->n_cus < m->max_wgs ? m->n_cus : m->max_wgs;
    const Variant* vp = nullptr;
    // (no device access here: the launch path stays asynchronous.  A failure word the HOST has seen -- an earlier PAIR launch
    // timed out waiting for its partner: the GPU is shared or partitioned -- takes the PAIR variants out of the choice: the model
    // keeps working on the one-workgroup kernels (it used to refuse every launch until dff_model_status_clear).  One
    // the host has not seen yet makes a PAIR kernel leave at entry with NaN outputs, and the next status call reports it.)
    if (m->sticky && !m->pair_off) {
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "dff: an earlier two-workgroups-per-protein launch timed out waiting for its partner workgroup (its results were "
                            "invalid and have been reported): continuing on the one-workgroup kernels; dff_model_status_clear(m) re-arms\n");
        }
    }
    if (!gen && m->cfg.conservative && !m->pair_off && !m->sticky) {
        const Variant* const cand = pick_pair(mt, v->spw);
        const int g_lo = auto_group ? 1 : G, g_hi = auto_group ? (cap > 1 ? cap : 1) : G;
        const int G0 = G;
        for (int g = g_lo; cand && g <= g_hi; ++g) {
            const int ngr = (a.B + g - 1) / g;
            if (2 * 8 * ((ngr + 7) / 8) > cu_cap || (g * N + 15) / 16 > mt || cand->lds_floats(N, g) * sizeof(float) > 160u * 1024u) continue;
            // the one-workgroup variant of the same shape builds the layer-0 table at THIS group size: it must fit there too
            // (it had been picked and LDS-checked at the old G)
            G = g;
            const Variant* const v1 = pick(mt);
            if (!v1) { G = G0; continue; }
            v = v1; vp = cand;
            break;
        }
    }
    if (want_tab) {
        int rc = ensure_l0_table(m, a.mode == DFF_MODE_DDPM ? 2 : 1, a.t_norm, G, v, stream, &a.l0_tab);
        if (rc) return rc;
    }
    if (vp) v = vp;
    return launch_generic(m, a, G, v, stream);
}

extern "C" int dff_score(dff_model* m, const float* x, const float* tnorm, int batch, float* force,
                         float* energy, void* stream) {
    if (!m || !x || !tnorm || !force) return fail(DFF_EINVAL, "null argument");
    if (batch <= 0) return batch == 0 ? DFF_OK : fail(DFF_EINVAL, "negative batch");
    if (energy && !m->cfg.conservative) return fail(DFF_EINVAL, "a non-conservative model has no energy (graph_transformer.py:106-113)");
    DffRunArgs a;
    memset(&a, 0, sizeof a);
    a.mode = DFF_MODE_SCORE; a.B = batch; a.n_steps = 1;
    a.x_in = x; a.tnorm = tnorm; a.force_out = force; a.energy_out = energy;
    a.save_interval = 1;
    return launch(m, a, (hipStream_t)stream);
}

extern "C" int dff_langevin_run(dff_model* m, const dff_langevin_params* p, int n_traj, float* x, float* v,
                                const float* noise, uint64_t seed, uint64_t traj_offset, uint64_t step_offset,
                                int n_steps, int save_interval, float* frames, float* ke, void* stream) {
    if (!m || !p || !x) return fail(DFF_EINVAL, "null argument");
    if (!p->overdamped && !v) return fail(DFF_EINVAL, "v_dev required unless overdamped");
    if (n_traj <= 0 || n_steps < 0) return fail(DFF_EINVAL, "bad sizes");
    if (n_steps == 0) return DFF_OK;
    if (save_interval <= 0) save_interval = n_steps;
    // "The save_interval must be a factor of the simulation length" (langevin_cgnet.py:305-309)
    if (frames && n_steps % save_interval != 0)
        return fail(DFF_EINVAL, "The save_interval must be a factor of the simulation length");
    DffRunArgs a;
    memset(&a, 0, sizeof a);
    a.mode = DFF_MODE_LANGEVIN; a.B = n_traj; a.n_steps = n_steps;
    a.x_io = x; a.v_io = v; a.noise = noise; a.seed = seed; a.item_offset = traj_offset; a.step_offset = step_offset;
    a.t_norm = p->t_norm; a.force_scale = p->force_scale; a.dt = p->dt; a.vscale = p->vscale;
    a.noisescale = p->noisescale; a.dtau = p->dtau; a.overdamped = p->overdamped;
    a.save_interval = save_interval; a.frames = frames; a.ke = ke;
    const float inv_beta = (float)(1.0 / (double)p->beta);
    for (int i = 0; i < m->cfg.n_beads; ++i) {
        const float mass = p->masses[i];
        if (!(mass > 0.f) && !p->overdamped) return fail(DFF_EINVAL, "masses must be positive");
        a.mass[i] = mass;
        a.inv_mass[i] = 1.0f / mass;
        a.noise_sigma[i] = sqrtf(inv_beta / mass);  // torch.sqrt(1.0 / beta / masses)  :465
    }
    a.brown_sigma = (float)std::sqrt(2.0 * (double)p->dtau / (double)p->beta);  // :497
    return launch(m, a, (hipStream_t)stream);
}

extern "C" int dff_ddpm_run(dff_model* m, int batch, float* x, const float* noise, uint64_t seed,
                            uint64_t sample_offset, int t_start, int t_end, int init_prior, int* clamp_flag,
                            void* stream) {
    if (!m || !x) return fail(DFF_EINVAL, "null argument");
    if (batch <= 0) return batch == 0 ? DFF_OK : fail(DFF_EINVAL, "negative batch");
    if (t_start >= m->cfg.timesteps || t_end < 0 || t_end > t_start) return fail(DFF_EINVAL, "bad timestep range");
    DffRunArgs a;
    memset(&a, 0, sizeof a);
    a.mode = DFF_MODE_DDPM; a.B = batch; a.n_steps = t_start - t_end + 1;
    a.x_io = x; a.noise = noise; a.seed = seed; a.item_offset = sample_offset;
    a.t_start = t_start; a.init_prior = init_prior; a.clamp_flag = clamp_flag; a.save_interval = 1;
    return launch(m, a, (hipStream_t)stream);
}

extern "C" int dff_debug_gemm(int device, const float* A, const float* W, int M, int K, int Nout, float* out) {
    if (!A || !W || !out || M < 1 || M > 64 || Nout % 16 || !(K == 64 || K == 128))
        return fail(DFF_EINVAL, "dff_debug_gemm: M<=64, K in {64,128}, Nout%%16==0");
    ON_DEVICE(device);
    std::vector<float> Wp = pack_b(K, Nout, [&](int k, int n) { return (double)W[(size_t)k * Nout + n]; });
    float *dA, *dW, *dO;
    HIPCHK(hipMalloc((void**)&dA, (size_t)M * K * 4));
    HIPCHK(hipMalloc((void**)&dW, Wp.size() * 4));
    HIPCHK(hipMalloc((void**)&dO, (size_t)M * Nout * 4));
    HIPCHK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(64 * (K + 4) + 64) * 4;
    HIPCHK((hipError_t)dff_debug_gemm_launch(K, dA, dW, M, Nout, dO, lds));
    HIPCHK(hipMemcpy(out, dO, (size_t)M * Nout * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dO);
    return DFF_OK;
}

extern "C" int dff_debug_stash(dff_model* m, int b, int layer, int what, float* out, size_t n) {
    if (!m || !out || !m->stash || m->last_G <= 0) return fail(DFF_EINVAL, "no stash (run dff_score first)");
    if (m->last_pair) return fail(DFF_EINVAL, "the last launch split every protein over two workgroups (dff_debug_pair(m, 0) turns that off)");
    const int N = m->cfg.n_beads, H = m->cfg.hidden, L = m->cfg.n_layers, G = m->last_G;
    if (b < 0 || b >= m->last_B || layer < 0 || layer >= L) return fail(DFF_EINVAL, "bad sample / layer");
    ON_DEVICE(m->device);
    HIPCHK(hipDeviceSynchronize());
    if (b < m->last_base) return fail(DFF_EINVAL, "sample %d was not part of the last launch (it started at %d)", b, m->last_base);
    const int wg = (b - m->last_base) / G, g = (b - m->last_base) % G;
    // both kernels stash the same per-head blocks; they differ in offsets, allocated rows and P stride
    unsigned o_nodes, o_attn, o_ff, o_hpre, o_qkv, o_P, lstride, total;
    int R, PS;
    if (m->last_small) {
        const SmallStash ss = dff_small_stash(N, G, H, L);
        o_nodes = ss.nodes_in; o_attn = ss.attn_out; o_ff = ss.ff; o_hpre = ss.h_pre; o_qkv = ss.qkv; o_P = ss.P;
        lstride = ss.layer_stride; total = ss.total;
        R = G * N + 1;   // arrays carry one dummy row
        PS = 16;
    } else {
        const StashLayout sl = dff_stash_layout(N, G, H, L, m->last_mt);
        o_nodes = sl.nodes_in; o_attn = sl.attn_out; o_ff = sl.ff; o_hpre = sl.h_pre; o_qkv = sl.qkvx; o_P = sl.P;
        lstride = sl.layer_stride; total = sl.total;
        R = G * N;
        PS = (int)sl.PS;
    }
    const int Prows = m->last_small ? 16 : R;   // rows of one head's P block
    const float* sb = m->stash + (size_t)wg * total + (size_t)layer * lstride;
    auto rowsS = [&](unsigned off, int width) -> int {
        if (n != (size_t)N * width) return fail(DFF_EINVAL, "expected %d values", N * width);
        HIPCHK(hipMemcpy(out, sb + off + (size_t)g * N * width, (size_t)N * width * 4, hipMemcpyDeviceToHost));
        return DFF_OK;
    };
    auto heads = [&](int c0, int w, int dstw) -> int {
        // per head h: rows (g*N .. g*N+N) x [c0, c0+w) of its (R x 208) block -> out[row][h*w ..]
        std::vector<float> tmp((size_t)N * DFF_QKVW);
        for (int h = 0; h < DFF_HEADS; ++h) {
            HIPCHK(hipMemcpy(tmp.data(), sb + o_qkv + ((size_t)h * R + (size_t)g * N) * DFF_QKVW, tmp.size() * 4, hipMemcpyDeviceToHost));
            for (int r = 0; r < N; ++r)
                for (int c2 = 0; c2 < w; ++c2) out[(size_t)r * dstw + h * w + c2] = tmp[(size_t)r * DFF_QKVW + c0 + c2];
        }
        return DFF_OK;
    };
    switch (what) {
        case 0: return rowsS(o_nodes, H);
        case 1: return rowsS(o_attn, H);
        case 2: return rowsS(o_ff, H);
        case 3: return rowsS(o_hpre, 4 * H);
        case 4: if (n != (size_t)N * 512) return fail(DFF_EINVAL, "size"); return heads(0, 64, 512);
        case 5: if (n != (size_t)N * 512) return fail(DFF_EINVAL, "size"); return heads(80, 64, 512);
        case 6: if (n != (size_t)N * 512) return fail(DFF_EINVAL, "size"); return heads(144, 64, 512);
        case 8: {
            if (n != (size_t)N * 32) return fail(DFF_EINVAL, "size");
            memset(out, 0, n * 4);
            return heads(64, 3, 32);
        }
        case 7: {
            if (n != (size_t)DFF_HEADS * N * N) return fail(DFF_EINVAL, "size");
            std::vector<float> tmp((size_t)N * PS);
            for (int h = 0; h < DFF_HEADS; ++h) {
                HIPCHK(hipMemcpy(tmp.data(), sb + o_P + ((size_t)h * Prows + (size_t)g * N) * PS, tmp.size() * 4, hipMemcpyDeviceToHost));
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) out[((size_t)h * N + i) * N + j] = tmp[(size_t)i * PS + g * N + j];
            }
            return DFF_OK;
        }
    }
    return fail(DFF_EINVAL, "unknown stash item %d", what);
}

extern "C" int dff_debug_profile(dff_model* m, int enable) {
    if (!m) return fail(DFF_EINVAL, "null model");
#if !DFF_PROF
    if (enable) return fail(DFF_EINVAL, "this library was built without the stage ticks (./build.sh with DFF_EXTRA_FLAGS=-DDFF_PROF=1)");
#endif
    ON_DEVICE(m->device);
    if (enable && !m->prof) {
        HIPCHK(hipMalloc((void**)&m->prof, DFF_NPROF * sizeof(unsigned long long)));
        HIPCHK(hipMemset(m->prof, 0, DFF_NPROF * sizeof(unsigned long long)));
    }
    m->prof_on = enable != 0;
    m->prof_wave = enable > 1 ? enable - 1 : 0;   // enable = 1 + wave whose view is recorded (<= 16-row kernel)
    if (enable) HIPCHK(hipMemset(m->prof, 0, DFF_NPROF * sizeof(unsigned long long)));
    return DFF_OK;
}

extern "C" int dff_debug_profile_read(dff_model* m, unsigned long long* out) {
    if (!m || !out || !m->prof) return fail(DFF_EINVAL, "profiling was never enabled");
    ON_DEVICE(m->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, m->prof, DFF_NPROF * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DFF_OK;
}

// ---------------------------------------------------------------------------------------------
// PWD histograms (dff_pwd.hip)
// ---------------------------------------------------------------------------------------------
extern "C" int dff_pwd_num_pairs(int n_beads, int offset) {
    if (n_beads < 1 || offset < 0) return 0;
    int np = 0;
    for (int i = 0; i < n_beads; ++i) np += (n_beads - i - offset > 0) ? n_beads - i - offset : 0;
    return np;
}

static int pwd_check(int device, const float* x, long long n, int N, int offset, int& npairs) {
    if ((!x && n > 0) || n < 0) return fail(DFF_EINVAL, "pwd: null input / negative count");
    if (N < 2 || N > DFF_MAX_BEADS) return fail(DFF_EINVAL, "pwd: n_beads must be 2..%d", DFF_MAX_BEADS);
    npairs = dff_pwd_num_pairs(N, offset);
    if (npairs <= 0) return fail(DFF_EINVAL, "pwd: no bead pairs at offset %d", offset);
    return DFF_OK;
}

// structures per workgroup: a multiple of the tile, enough workgroups to fill the chip, and long
// enough that the per-workgroup flush stays small next to the streaming part
static long long pwd_chunk(long long n, long long want_wgs, long long min_chunk) {
    long long chunk = (n + want_wgs - 1) / want_wgs;
    if (chunk < min_chunk) chunk = min_chunk;
    chunk = (chunk + DFF_PWD_TILE - 1) / DFF_PWD_TILE * DFF_PWD_TILE;
    return chunk;
}

extern "C" int dff_pwd_max(int device, const float* x, long long n, int N, int offset, float* max_out, void* stream_) {
    int npairs;
    int rc = pwd_check(device, x, n, N, offset, npairs);
    if (rc) return rc;
    ON_DEVICE(device);
    if (!max_out) return fail(DFF_EINVAL, "pwd: null output");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipMemsetAsync(max_out, 0, (size_t)npairs * sizeof(float), stream));
    if (n == 0) return DFF_OK;
    const long long chunk = pwd_chunk(n, 1024, 4 * DFF_PWD_TILE);
    const int grid = (int)((n + chunk - 1) / chunk);
    const unsigned lds = (unsigned)(DFF_PWD_TILE * 3 * N * sizeof(float) + 16);
    int pc_log2 = 0;
    while ((1 << pc_log2) < npairs && pc_log2 < 8) ++pc_log2;
    if (npairs > (DFF_PWD_MAXG << pc_log2)) return fail(DFF_EINVAL, "pwd: too many pairs (%d)", npairs);
    const int vec4 = ((uintptr_t)x % 16) == 0;
    hipLaunchKernelGGL(dff_pwd_max_kernel, dim3(grid), dim3(DFF_PWD_THREADS), lds, stream, x, n, N, offset, npairs,
                       pc_log2, chunk, (unsigned*)max_out, vec4);
    HIPCHK(hipGetLastError());
    return DFF_OK;
}

extern "C" int dff_pwd_hist(int device, const float* x, long long n, int N, int offset, const int32_t* nbins,
                            const float* hmax, int max_bins, int ld, uint32_t* hist, void* stream_) {
    int npairs;
    int rc = pwd_check(device, x, n, N, offset, npairs);
    if (rc) return rc;
    ON_DEVICE(device);
    if (!nbins || !hmax || !hist) return fail(DFF_EINVAL, "pwd: null argument");
    if (max_bins < 1 || ld < max_bins) return fail(DFF_EINVAL, "pwd: need 1 <= max_bins <= ld");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)npairs * ld * sizeof(uint32_t), stream));
    if (n == 0) return DFF_OK;
    // LDS: one tile of structures + the privatised histograms.  The tile shrinks (64 -> 32 -> 16 structures) when
    // that lets twice as many pairs keep their histograms in LDS (fewer passes over the structures).
    const int ldl = max_bins | 1;   // odd leading dimension: pairs land in different LDS banks
    int tile_n = DFF_PWD_TILE, pc_log2 = -1, tile_bytes = 0;
    for (int tn = DFF_PWD_TILE; tn >= 16; tn >>= 1) {
        const int tb = tn * 3 * N * (int)sizeof(float);
        int slots = (160 * 1024 - 64 - tb) / (int)sizeof(unsigned);
        if (slots > DFF_PWD_LDS_BINS) slots = DFF_PWD_LDS_BINS;
        if (ldl > slots) continue;
        int lg = 8;                 // pair lanes per workgroup: the largest power of two whose histograms fit
        while (lg > 0 && ((1 << lg) * ldl > slots || (1 << (lg - 1)) >= npairs)) --lg;
        if (lg > pc_log2) { pc_log2 = lg; tile_n = tn; tile_bytes = tb; }
    }
    if (pc_log2 < 0) return fail(DFF_EINVAL, "pwd: %d bins per pair do not fit in LDS", max_bins);
    const int PC = 1 << pc_log2;
    const int npc = (npairs + PC - 1) / PC;
    // each workgroup flushes up to PC * ldl bins: give it at least ~2x that many (pair, structure) items
    const long long min_chunk = (2LL * ldl + DFF_PWD_TILE - 1) / DFF_PWD_TILE * DFF_PWD_TILE;   // multiple of every tile size
    const long long chunk = pwd_chunk(n, (2048 + npc - 1) / npc, min_chunk);
    long long nsc = (n + chunk - 1) / chunk;
    nsc = (nsc + 7) / 8 * 8;       // whole XCD rounds (empty chunks return at once)
    const long long grid = nsc * npc;
    if (grid > 0x7fffffffLL) return fail(DFF_EINVAL, "pwd: grid too large");
    const unsigned lds = (unsigned)(tile_bytes + (size_t)PC * ldl * sizeof(unsigned) + 16);
    if (lds > 160 * 1024) return fail(DFF_EINVAL, "pwd: LDS budget exceeded (%u bytes)", lds);
    HIPCHK(hipFuncSetAttribute((const void*)&dff_pwd_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int vec4 = ((uintptr_t)x % 16) == 0;
    hipLaunchKernelGGL(dff_pwd_hist_kernel, dim3((unsigned)grid), dim3(DFF_PWD_HIST_THREADS), lds, stream, x, n, N, offset,
                       npairs, nbins, hmax, ld, pc_log2, npc, chunk, ldl, hist, vec4, tile_n);
    HIPCHK(hipGetLastError());
    return DFF_OK;
}

extern "C" const char* dff_last_error(void) { return g_err.c_str(); }
// DFF_SRC_SHA (build.sh): sha256 over csrc/* + include/dff.h, 16 hex digits -- ties rocprof evidence to the code it profiled
#ifndef DFF_SRC_SHA
#define DFF_SRC_SHA unknown
#endif
#define DFF_STR2(x) #x
#define DFF_STR(x) DFF_STR2(x)
// DFF_BUILD_FLAGS (build.sh, generated header): DFF_EXTRA_FLAGS / scheduler overrides the translation units were compiled with -- a
// development build names its knobs; they are hashed into DFF_SRC_SHA as well, so that it never shares the product build's hash.
// DFF_EXPERIMENT: the build says of itself that it may compute wrong numbers (timing experiments).
#ifndef DFF_BUILD_FLAGS
#define DFF_BUILD_FLAGS ""
#endif
#ifdef DFF_EXPERIMENT
#define DFF_EXP_TAG " EXPERIMENT"
#else
#define DFF_EXP_TAG ""
#endif
extern "C" const char* dff_version(void) {
    return "dff-amd 0.1 (gfx950; f16-split / f32 MFMA)" DFF_EXP_TAG " flags=[" DFF_BUILD_FLAGS "] src=" DFF_STR(DFF_SRC_SHA);
}
