// dff_internal.h -- structures shared by the host API (dff_host.hip) and the device code
// (dff_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DFF_MAX_LAYERS 8
#define DFF_HEADS 8
#define DFF_DH 64
#define DFF_INNER 512
#ifndef DFF_NTHREADS
#define DFF_NTHREADS 512   // generic kernel: 8 waves = two per SIMD (epilogue VALU of one overlaps MFMA of the other)
#define DFF_NWAVES 8
#endif
#define DFF_NPROF 24       // per-stage cycle counters (debug)
#define DFF_SMALL_LD 36   // leading dim of the 32-column u / xrel / r / du buffers

// Packed B-operand layout for v_mfma_f32_16x16x4_f32 (see pack_b in dff_host.hip):
//   block (nt, kb) = 256 floats holding W[16kb .. 16kb+15][16nt .. 16nt+15];
//   inside a block lane l = (kk = l>>4, nn = l&15) owns 4 consecutive floats
//   s = 0..3  <->  W[16kb + 4kk + s][16nt + nn];   blocks ordered [nt][kb].
// MFMA step s of k-block kb therefore contracts k = 16kb + 4kk + s (a permutation of the
// natural k order inside each 16-block; A fragments are read with the same permutation).

struct DffLayerDev {
    // forward
    const float *ln1_g, *ln1_b;
    const float *bo;              // folded: bo + Wo b_c
    const float *g1;              // (3H) gate weights [x | res | x-res]
    const float *ln2_g, *ln2_b;
    const float *W1_p, *b1;       // K=H, Nout=4H
    const float *W2_p, *b2;       // K=4H, Nout=H
    const float *g2;
    // backward (transposed orientation)
    const float *W2T_p;           // K=H, Nout=4H   dh  = dff  W2
    // opt-in (DFF_SPLIT_BF16=1): the K = H images as three bf16 pieces per weight (dff_host.hip pack_b_split)
    const unsigned *Wqkvx_s, *W1_s, *W2T_s, *WoxT_s, *W2_s, *W1T_s, *Wox_s, *WqkvxT_s;
    const float *W1T_p;           // K=4H, Nout=H   df  = dhp  W1
    // split-bf16 images of all eight weight GEMMs for the <= 16-row kernel (dff_small.hip SPW variants; dff_host.hip
    // pack_units): *_w K = H, units ordered [tile][k-block]; *_t Nout = H, units ordered [k-block][tile]
    const unsigned *Wqkvx_w, *W1_w, *W2T_w, *WoxT_w, *Wox_t, *W2_t, *W1T_t, *WqkvxT_t;
    // "extended head" images: per head 80 = 64 + 16 extension columns / rows ([u (3) | s | 0...], [xrel (3) | D | 0...])
    const float *Wqkvx_p, *bqkvx; // K=H, Nout=8*208, per head [q 64 | ext 16 | k 64 | v 64]
    const float *Wox_p;           // K=8*80 per head [o 64 | ext 16], Nout=H
    const float *WoxT_p;          // K=H, Nout=8*80
    const float *WqkvxT_p;        // K=8*208, Nout=H
};

struct DffModelDev {
    int N, H, L, T;
    const float *WnT;   // (N+1, H): node_embedding.weight transposed (one-hot columns, then t)
    const float *bn;    // (H)
    const float *wdec;  // (H) conservative: energy head ; (3,H) otherwise: force head
    float bdec;         // conservative
    float bdec3[3];     // non-conservative
    int conservative;   // 1: forces = -dE/dx (hand-written VJP) ; 0: forces = node_decoder(nodes) (forward only)
    // input branches (graph_transformer.py:53-58,99-102,116-140).  The shipped checkpoints are intr=1, dist=0,
    // abs=0; anything else runs the GEN variants of the generic kernel.
    int in_intr, in_dist, in_abs;
    int wn_t;           // row of WnT that multiplies t: N, or N + 3 with absolute coordinates (rows N..N+2 multiply x)
    DffLayerDev layer[DFF_MAX_LAYERS];
    // schedule tables, float32 (T each)
    const float *sqrt_recip_ac, *sqrt_recipm1_ac, *post_c1, *post_c2, *post_logvar;
};

enum DffMode { DFF_MODE_SCORE = 0, DFF_MODE_LANGEVIN = 1, DFF_MODE_DDPM = 2 };

struct DffRunArgs {
    int mode;
    int B;            // proteins (samples / trajectories) of the whole call (stride of noise / frames / ke)
    int b_base;       // first protein of this launch: big batches run as several launches over one bounded stash
    int G;            // proteins per workgroup
    int n_steps;
    // state
    const float* x_in;    // SCORE: (B,N,3) input
    float* x_io;          // LANGEVIN/DDPM: (B,N,3) in/out
    float* v_io;          // LANGEVIN: (B,N,3) in/out
    const float* tnorm;   // SCORE: (B)
    float* force_out;     // SCORE: (B,N,3)
    float* energy_out;    // SCORE: (B,N) or null
    // noise
    const float* noise;   // (n_steps,B,N,3) or null -> philox
    uint64_t seed, item_offset, step_offset;
    // langevin
    float t_norm, force_scale, dt, vscale, noisescale, dtau, brown_sigma;
    int overdamped, save_interval;
    float inv_mass[64];     // 1/m
    float noise_sigma[64];  // sqrt(1/beta/m)
    float mass[64];
    float* frames;          // (n_steps/save, B, N, 3) or null
    float* ke;              // (n_steps/save, B) or null
    // ddpm
    int t_start, init_prior;
    int* clamp_flag;
    // scratch
    unsigned long long* prof;   // optional: DFF_NPROF per-stage cycle totals of block 0
    int prof_wave;              // ... as seen by lane 0 of this wave (<= 16-row kernel)
    float* stash;               // per-workgroup stash slots
    unsigned long long stash_stride; // floats per workgroup
    // optional table of precomputed layer-0 inputs (nodes_in, q|u|k|v), one stash-layer-shaped entry per
    // noise level: entry 0 (Langevin, fixed t) or entry t (DDPM).  Rows-<=16 kernel only.
    const float* l0_tab;
    // PAIR variants of the <= 64-row kernel (two workgroups per protein): partial-tile exchange slots
    // [pair][half][parity][rows x (H + 4)]; xflag = one sticky error word at [0], then flags [pair][half]
    float* xchg;
    unsigned* xflag;
    int xpairs;
    int xslow;       // tests: never take the same-XCD fast path of the exchanges (the agent-scope protocol a cross-XCD pair runs)
};

// which split variants of the <= 16-row kernel were compiled with the two-piece fp16 engine (dff_small.hip DFF_F16): 0 none,
// 1 the FOLD variant, 2 all -- the host packs their weight images accordingly (dff_host.hip pack_units_f16)
int dff_small_f16_level();

// bit mask of the <= 64-row split variants' GEMM groups that take two-piece fp16 images (dff_kernels.hip DFF_F16G):
// 1 = forward (Wqkvx_s, Wox_s, W1_s, W2_s), 2 = FFN backward (W2T_s, W1T_s), 4 = G_ext (WoxT_s), 8 = QKV_ext^T (WqkvxT_s)
int dff_fused_f16_mask();
