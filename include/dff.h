/*
 * dff.h -- C ABI of the MI355X-native denoising-force-field sampler (libdff_amd.so).
 *
 * The reference (microsoft/two-for-one-diffusion) is pure Python/PyTorch and has no FFI; the
 * seams this library sits behind are its Python call signatures.  Each
 * entry point below names the reference interface it replaces.  Plain pointers and sizes only:
 * no torch types cross this boundary.  The host-side mirror of the reference classes that
 * binds these symbols (ctypes) is two-for-one-diffusion_amd/{binding,score,ddpm,langevin}.py;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every *_dev pointer is device memory on the model's GPU, fp32, row-major, owned by the
 *     caller (e.g. torch-ROCm tensors); the library owns only its packed weights and scratch;
 *   - calls enqueue on `stream` (a hipStream_t, NULL = default stream) and do not synchronise;
 *   - return 0 on success, a DFF_E* / hipError_t-derived code otherwise; dff_last_error()
 *     gives the message.  No exceptions cross the ABI;
 *   - one handle per device; not thread-safe per handle (the reference is single-threaded per
 *     replica, sample.py:176-190).
 */
#ifndef DFF_H
#define DFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFF_OK 0
#define DFF_EINVAL 1      /* bad argument / unsupported configuration */
#define DFF_EHIP 2        /* a HIP runtime call failed (see dff_last_error) */
#define DFF_ENOMEM 3

#define DFF_MAX_BEADS 64      /* array bound of the ABI structs */
#define DFF_MAX_BEADS_LDS 61  /* largest n_beads dff_model_create accepts: what the kernels' 160 KB of LDS hold (hidden 128) */

typedef struct dff_model dff_model; /* opaque */

/* Hyper-parameters of GraphTransformer.__init__ as sample.py passes them
 * (models/__init__.py:4-15, models/graph_transformer.py:23-75).  Every combination of the 0/1 input
 * flags is implemented: use_intrinsic_coords (edge features x_j - x_i), use_distances (edge feature
 * |x_j - x_i|^2), use_abs_coords (node features include x) -- :53-58,99-102,116-140 -- and
 * conservative (1: forces = -dE/dx through the hand-written VJP; 0: the force head of :62-65,112-113,
 * node_decoder = Linear(H,3), forces = its output, no energy).  All shipped checkpoints are
 * (1, 0, 0, conservative 1) and run the specialised kernels; main_train.py's defaults (0, 1, 1) and the
 * other combinations run the general ("gen") variants of the <= 64-row kernel. */
typedef struct {
    int32_t n_beads;              /* num_beads, 2..DFF_MAX_BEADS_LDS */
    int32_t hidden;               /* hidden_features_gnn: 64, 96, 128 (n_beads <= 61 with 128, <= 32 otherwise) or 256 (<= 32 beads) */
    int32_t n_layers;             /* num_layers_gnn, 1..8 */
    int32_t timesteps;            /* diffusion_steps (GaussianDiffusion timesteps), e.g. 1000 */
    int32_t use_intrinsic_coords; /* 0 / 1 */
    int32_t use_distances;        /* 0 / 1 */
    int32_t use_abs_coords;       /* 0 / 1 */
    int32_t conservative;         /* 1: forces = -dE/dx ; 0: forces = node_decoder(nodes) */
} dff_config;

/* Number of fp32 values dff_model_create expects for this config. */
size_t dff_weight_count(const dff_config* cfg);

/* Build a model from the GraphTransformer parameters, given as ONE flat host fp32 array in the
 * reference's state_dict() order: node_embedding.{weight (H, N+1+3*abs), bias},
 * edge_embedding.{weight (H, 3*intrinsic+distances, or 1 if neither), bias}, node_decoder.{weight (D,H), bias (D)} with D = 1
 * (conservative) or 3, then per layer
 * l: attn to_q.{weight (512,H), bias}, to_kv.{weight (1024,H), bias}, edges_to_kv.{weight
 * (512,H), bias}, to_out.{weight (H,512), bias}, norm.{weight, bias}, gate proj.0.weight
 * (1,3H), ff fn.0.{weight (4H,H), bias}, fn.2.{weight (H,4H), bias}, norm.{weight, bias}, gate
 * proj.0.weight (1,3H).  Replaces get_model + load_state_dict (sample.py:142-167): the
 * library folds edge_embedding into edges_to_kv, packs the GEMM operands for MFMA and builds
 * the cosine schedule tables of GaussianDiffusion.__init__ (models/ddpm.py:45-99). */
int dff_model_create(const dff_config* cfg, const float* weights_host, size_t n_weights,
                     int device, dff_model** out);
void dff_model_destroy(dff_model* m);

/* GaussianDiffusion schedule buffer `which` (models/ddpm.py:61-99) copied to out_host[timesteps]:
 * 0 betas, 1 alphas_cumprod, 2 alphas_cumprod_prev, 3 sqrt_alphas_cumprod,
 * 4 sqrt_one_minus_alphas_cumprod, 5 log_one_minus_alphas_cumprod, 6 sqrt_recip_alphas_cumprod,
 * 7 sqrt_recipm1_alphas_cumprod, 8 posterior_variance, 9 posterior_log_variance_clipped,
 * 10 posterior_mean_coef1, 11 posterior_mean_coef2. */
int dff_schedule(const dff_model* m, int which, float* out_host);

/* The score op: GraphTransformer.forward (models/graph_transformer.py:77-114) including
 * compute_forces (:143-159).  x_dev (batch,N,3) need not be centred; tnorm_dev (batch) is the
 * normalised time t/T per sample; force_dev (batch,N,3) receives -d(sum E)/d(x_centred);
 * energy_dev (batch,N), optional, receives the per-bead energies (return_energy=True). */
int dff_score(dff_model* m, const float* x_dev, const float* tnorm_dev, int batch,
              float* force_dev, float* energy_dev, void* stream);

/* Langevin.simulate driven by ForcesWrapper (dynamics/langevin_cgnet.py:686-792,447-500;
 * dynamics/langevin.py:75-92), n_steps steps in ONE launch.  Scalars are what
 * LangevinDiffusion.__init__ / Langevin._input_option_checks compute on the host
 * (langevin.py:131-184, langevin_cgnet.py:329-330,343). */
typedef struct {
    float t_norm;      /* noise level t / diffusion_steps (langevin.py:68) */
    float force_scale; /* 1 / (kbt_inv * sqrt_one_minus_alphas_cumprod[t]) (langevin.py:79-87) */
    float dt;          /* time step */
    float vscale;      /* exp(-friction dt); ignored when overdamped */
    float noisescale;  /* sqrt(1 - vscale^2); ignored when overdamped */
    float beta;        /* kb_inv / temp_sim */
    float dtau;        /* diffusion * dt, overdamped only (langevin_cgnet.py:343) */
    int32_t overdamped;            /* 1: friction is None -> Brownian step (:481-500) */
    float masses[DFF_MAX_BEADS];   /* per-bead masses (first n_beads used) */
} dff_langevin_params;

/* x_dev (n_traj,N,3) in/out, normalised units (init_mol / norm_factor); v_dev (n_traj,N,3)
 * in/out (ignored when overdamped).  noise_dev: (n_steps,n_traj,N,3) standard normals to use
 * (parity mode), or NULL to draw them in-kernel from Philox4x32-10 keyed by (seed; trajectory
 * traj_offset+i, step step_offset+s, bead).  Every save_interval steps the UN-centred x_new is
 * written to frames_dev (n_steps/save_interval, n_traj, N, 3) and 0.5 sum m v^2 to ke_dev
 * (n_steps/save_interval, n_traj) -- the layout of Langevin.simulated_coords /
 * .kinetic_energies before _swap_and_export (langevin_cgnet.py:410-425,502-542).  Either may be
 * NULL.  n_steps must be a multiple of save_interval when frames_dev != NULL. */
int dff_langevin_run(dff_model* m, const dff_langevin_params* p, int n_traj, float* x_dev,
                     float* v_dev, const float* noise_dev, uint64_t seed, uint64_t traj_offset,
                     uint64_t step_offset, int n_steps, int save_interval, float* frames_dev,
                     float* ke_dev, void* stream);

/* GaussianDiffusion.p_sample_loop body (models/ddpm.py:195-254): reverse steps t_start,
 * t_start-1, ..., t_end (inclusive) in ONE launch, each followed by the +-1000 clamp and
 * centring.  x_dev (batch,N,3) in/out in normalised units; if init_prior != 0 x is first set to
 * center_zero(randn) in-kernel (ddpm.py:242).  noise_dev: (t_start-t_end+1,batch,N,3) draws for
 * randn_like (ddpm.py:228), or NULL for in-kernel Philox keyed by (seed; sample
 * sample_offset+i, t).  *clamp_flag_dev (optional) is set to 1 if any coordinate was clamped
 * (the reference's "Large molecule encountered" warning, ddpm.py:248-250). */
int dff_ddpm_run(dff_model* m, int batch, float* x_dev, const float* noise_dev, uint64_t seed,
                 uint64_t sample_offset, int t_start, int t_end, int init_prior,
                 int* clamp_flag_dev, void* stream);

/* Sticky status word of everything this model has launched so far (synchronises the device).  0 = fine.  Bit 0: a launch
 * of a two-workgroups-per-protein kernel variant (chosen automatically for batches that would leave half the CUs idle, see
 * dff_debug_pair) gave up waiting for a partner workgroup -- possible only when the GPU is shared with another process or
 * partitioned below the CU count the driver reports; the results of that launch are invalid.  The samplers call this at
 * their host synchronisation points (end of LangevinDiffusion.simulate, GaussianDiffusion.check_clamp, the CLI) and raise;
 * once the host has seen the word, further launches on the same model run the one-workgroup-per-protein kernels (round 6;
 * they used to be refused with DFF_EHIP) until dff_model_status_clear re-arms; one queued before that
 * leaves at kernel entry (the word is read on the device) with its OUTPUTS set to NaN (forces / energies, frames / kinetic
 * energies, samples; the Langevin state x, v is left alone), so nothing runs on top of invalid results and a caller that
 * never checks cannot mistake an unwritten buffer for forces.  The entry points
 * themselves never read it: they only enqueue on the caller's stream (round 4; they used to synchronise the device before
 * every two-workgroups launch).  The reference has no counterpart (one process, one kernel per op).  NOTE: which variant runs depends on the per-call batch (<= n_CUs / 2 proteins), and the two variants sum
 * in different orders: trajectories are bit-reproducible for a fixed per-rank batch, not across batch splits that cross
 * that threshold. */
int dff_model_status(dff_model* m, unsigned* status);
/* Clear the sticky word (synchronises the device): re-arms the two-workgroups variants after the caller has dealt with a
 * reported failure (e.g. the co-tenant that held the CUs is gone).  dff_debug_pair(m, 0) clears it as well (whether or not
 * the host has read it yet). */
int dff_model_status_clear(dff_model* m);

/* ---- introspection / debugging (used by tests and bench.py, not by samplers) ---- */

/* Proteins handled per workgroup for this model (0 = choose automatically from batch). */
int dff_set_group(dff_model* m, int proteins_per_workgroup);
/* Debugging: on != 0 disables the rows<=16 fast-path kernel so the generic kernel runs. */
int dff_debug_force_generic(dff_model* m, int on);
/* Debugging: waves per workgroup of the rows<=16 kernel: 0 auto (8 where it applies), 4 or 8.  A non-zero value also keeps
 * hidden-96 / 128 models on that kernel (by default they run the one-row-tile split variant of the <= 64-row kernel, which is faster). */
int dff_debug_small_waves(dff_model* m, int waves);
/* Workgroups per kernel launch (default 2048).  Proteins are independent, so a batch that needs more
 * workgroups runs as consecutive launches over one bounded scratch "stash" (n x stash slot) instead
 * of a scratch allocation that grows with the batch.  Results do not depend on the limit. */
int dff_debug_max_workgroups(dff_model* m, int n);
/* Debugging: on == 0 makes the sampling loops recompute layer 0 every step instead of reading the
 * precomputed per-noise-level table of layer-0 inputs (results are bit-identical either way). */
int dff_debug_l0_table(dff_model* m, int on);
/* Debugging: on == 2 = on, and the exchanges always run the agent-scope protocol of a pair whose blocks sit on different XCDs
 * (never observed: blocks b and b + 8 share one; the kernel checks at run time and takes an L2-local path when they do).
 * on == 3 = on, with partners placed on ADJACENT blocks (b, b + 1): under the hardware's round-robin placement they sit on
 * different XCDs, so the run-time check itself selects the agent-scope protocol on pairs that really span two L2s.
 * on == 0 never splits a protein over two workgroups (the PAIR variants of the <= 64-row kernel, chosen
 * automatically when one workgroup per protein would leave at least half the CUs idle, e.g. protein G at 128 per GPU). */
int dff_debug_pair(dff_model* m, int on);
/* dff_model_status for tests: *status = the sticky word, which is then CLEARED.  Synchronises the device. */
int dff_debug_pair_status(dff_model* m, int* status);
/* Tests: overwrite the sticky word ON THE DEVICE as a kernel that lost its partner would (the host's cached copy is not
 * touched), to exercise the failure path: queued two-workgroups launches leave at entry, dff_model_status reports. */
int dff_debug_poke_status(dff_model* m, unsigned word);
/* Name of the kernel the last call launched, grid size and dynamic LDS bytes. */
int dff_last_launch(const dff_model* m, const char** kernel_name, int* grid, int* lds_bytes);
/* Run one MFMA GEMM stage out(M,Nout) = A(M,K) W(K,Nout) through the same device routine and
 * weight packing the score kernel uses (M <= 64; K, Nout multiples of 16).  Host pointers. */
int dff_debug_gemm(int device, const float* A_host, const float* W_host, int M, int K, int Nout,
                   float* out_host);
/* Copy one stashed forward intermediate of the LAST dff_score call for sample `b`, layer `l`
 * to out_host: what 0 nodes_in (N,H), 1 attn_out (N,H), 2 ff (N,H), 3 gelu'(h_pre) (N,4H; what the backward needs),
 * 4 q (N,512), 5 k (N,512), 6 v (N,512), 7 P (8,N,N), 8 u (N,32). */
int dff_debug_stash(dff_model* m, int b, int layer, int what, float* out_host, size_t n);

/* Per-stage cycle accounting of workgroup 0 (s_memtime deltas accumulated at the stage
 * boundaries of the fused kernel): enable != 0 makes subsequent launches record; _read copies
 * the DFF_NPROF (=24) totals of the last launch (shader-clock cycles) to out_host. */
int dff_debug_profile(dff_model* m, int enable);
int dff_debug_profile_read(dff_model* m, unsigned long long* out_host);

/* ---- pairwise-distance (PWD) histograms for the Jensen-Shannon sample-quality metric ----
 * Replaces, for structures already resident on the GPU, evaluate/evaluators.py: get_pwd_triu_batch
 * (:934-948), the per-pair maximum (:239, :259) and the per-pair torch.histc (:241-247, :261-263) of
 * PwdEvaluator -- without materialising the (n, n_pairs) distance matrix.  Pairs are (i, j >= i+offset)
 * in torch.triu_indices order; distances and bin selection reproduce torch's float32 arithmetic, so
 * the counts equal the reference's.  The (n_pairs x bins) JS reduction stays on the host
 * (two-for-one-diffusion_amd/evaluate.py). */
int dff_pwd_num_pairs(int n_beads, int offset);
/* max_out_dev[p] = max over the n structures of d_p (0 when n == 0).  x_dev: (n, n_beads, 3) fp32. */
int dff_pwd_max(int device, const float* x_dev, long long n, int n_beads, int offset,
                float* max_out_dev, void* stream);
/* hist_dev[p * ld + b], b < nbins_dev[p]: number of structures whose d_p falls into bin b of
 * torch.histc(d_p, bins=nbins[p], min=0, max=hmax_dev[p]); the call zeroes hist_dev (n_pairs * ld)
 * first.  max_bins >= every nbins[p]; ld >= max_bins. */
int dff_pwd_hist(int device, const float* x_dev, long long n, int n_beads, int offset,
                 const int32_t* nbins_dev, const float* hmax_dev, int max_bins, int ld,
                 uint32_t* hist_dev, void* stream);

const char* dff_last_error(void);
const char* dff_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DFF_H */
