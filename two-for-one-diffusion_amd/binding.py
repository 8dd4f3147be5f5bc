"""ctypes binding of libdff_amd.so (C ABI: include/dff.h).

The HIP library is the product path: there is NO CPU / PyTorch fallback.  If the shared
library is missing or fails to load, importing a sampler raises ``DffLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFF_LIB_PATH") or os.path.join(_HERE, "libdff_amd.so")   # DFF_LIB_PATH: development builds
DFF_MAX_BEADS = 64

SCHEDULE_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


class DffLibraryError(RuntimeError):
    pass


class DffConfig(C.Structure):
    _fields_ = [("n_beads", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32),
                ("timesteps", C.c_int32), ("use_intrinsic_coords", C.c_int32),
                ("use_distances", C.c_int32), ("use_abs_coords", C.c_int32),
                ("conservative", C.c_int32)]


class DffLangevinParams(C.Structure):
    _fields_ = [("t_norm", C.c_float), ("force_scale", C.c_float), ("dt", C.c_float),
                ("vscale", C.c_float), ("noisescale", C.c_float), ("beta", C.c_float),
                ("dtau", C.c_float), ("overdamped", C.c_int32),
                ("masses", C.c_float * DFF_MAX_BEADS)]


# every symbol include/dff.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "dff_weight_count": (C.c_size_t, [C.POINTER(DffConfig)]),
    "dff_model_create": (C.c_int, [C.POINTER(DffConfig), _P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "dff_model_destroy": (None, [_P]),
    "dff_schedule": (C.c_int, [_P, C.c_int, _P]),
    "dff_score": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P]),
    "dff_langevin_run": (C.c_int, [_P, C.POINTER(DffLangevinParams), C.c_int, _P, _P, _P, C.c_uint64,
                                   C.c_uint64, C.c_uint64, C.c_int, C.c_int, _P, _P, _P]),
    "dff_ddpm_run": (C.c_int, [_P, C.c_int, _P, _P, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int,
                               _P, _P]),
    "dff_set_group": (C.c_int, [_P, C.c_int]),
    "dff_debug_force_generic": (C.c_int, [_P, C.c_int]),
    "dff_debug_small_waves": (C.c_int, [_P, C.c_int]),
    "dff_debug_l0_table": (C.c_int, [_P, C.c_int]),
    "dff_debug_max_workgroups": (C.c_int, [_P, C.c_int]),
    "dff_last_launch": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dff_debug_gemm": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "dff_debug_stash": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t]),
    "dff_debug_profile": (C.c_int, [_P, C.c_int]),
    "dff_debug_profile_read": (C.c_int, [_P, _P]),
    "dff_pwd_num_pairs": (C.c_int, [C.c_int, C.c_int]),
    "dff_pwd_max": (C.c_int, [C.c_int, _P, C.c_longlong, C.c_int, C.c_int, _P, _P]),
    "dff_pwd_hist": (C.c_int, [C.c_int, _P, C.c_longlong, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P]),
    "dff_last_error": (C.c_char_p, []),
    "dff_debug_pair": (C.c_int, [_P, C.c_int]),
    "dff_debug_pair_status": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "dff_model_status": (C.c_int, [_P, C.POINTER(C.c_uint)]),
    "dff_model_status_clear": (C.c_int, [_P]),
    "dff_debug_poke_status": (C.c_int, [_P, C.c_uint]),
    "dff_version": (C.c_char_p, []),
}

_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libdff_amd.so and type every exported entry point; loud failure if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise DffLibraryError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or ./build.sh).  There is no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise DffLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DffLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.dff_last_error().decode(errors="replace")
        if rc == 1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Model:
    """Owner of one ``dff_model`` handle (packed weights + scratch on one GPU)."""

    def __init__(self, n_beads: int, hidden: int, n_layers: int, flat_weights: np.ndarray,
                 timesteps: int = 1000, device: int = 0, use_intrinsic_coords=True,
                 use_distances=False, use_abs_coords=False, conservative=True):
        self.lib = load_library()
        self.cfg = DffConfig(n_beads, hidden, n_layers, timesteps, int(bool(use_intrinsic_coords)),
                             int(bool(use_distances)), int(bool(use_abs_coords)), int(bool(conservative)))
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        self.handle = C.c_void_p()
        rc = self.lib.dff_model_create(C.byref(self.cfg), w.ctypes.data_as(C.c_void_p), w.size, device,
                                       C.byref(self.handle))
        _check(self.lib, rc, "dff_model_create")
        self.device = device
        self.n_beads, self.hidden, self.n_layers, self.timesteps = n_beads, hidden, n_layers, timesteps

    def close(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            self.lib.dff_model_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: ctypes may already be torn down
            pass

    # ---- helpers
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev_tensor(self, t, shape=None, name="tensor"):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")
        if t.device.index != self.device:
            raise ValueError(f"{name} is on {t.device}, model is on cuda:{self.device}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t

    def schedule(self, name: str) -> np.ndarray:
        out = np.empty(self.timesteps, np.float32)
        _check(self.lib, self.lib.dff_schedule(self.handle, SCHEDULE_NAMES.index(name), out.ctypes.data_as(C.c_void_p)),
               "dff_schedule")
        return out

    def set_group(self, g: int):
        _check(self.lib, self.lib.dff_set_group(self.handle, int(g)), "dff_set_group")

    def force_generic(self, on: bool = True):
        _check(self.lib, self.lib.dff_debug_force_generic(self.handle, int(on)), "dff_debug_force_generic")

    def small_waves(self, waves: int = 0):
        _check(self.lib, self.lib.dff_debug_small_waves(self.handle, int(waves)), "dff_debug_small_waves")

    def max_workgroups(self, n: int = 2048):
        _check(self.lib, self.lib.dff_debug_max_workgroups(self.handle, int(n)), "dff_debug_max_workgroups")

    def pair(self, on=True):
        """True / False: allow / never use the two-workgroups-per-protein variants; 2: allow, and force their cross-XCD exchange protocol;
        3: allow, with the two workgroups of a protein on adjacent blocks (different XCDs: the protocol is chosen by the kernel's own check)."""
        _check(self.lib, self.lib.dff_debug_pair(self.handle, int(on)), "dff_debug_pair")

    def pair_status(self) -> int:
        st = C.c_int(0)
        _check(self.lib, self.lib.dff_debug_pair_status(self.handle, C.byref(st)), "dff_debug_pair_status")
        return st.value

    def status(self) -> int:
        """Sticky status word of every launch so far (dff_model_status; synchronises the device)."""
        st = C.c_uint(0)
        _check(self.lib, self.lib.dff_model_status(self.handle, C.byref(st)), "dff_model_status")
        return st.value

    def check(self):
        """Raise if any launch so far reported a failure the kernels cannot return synchronously (bit 0: a
        two-workgroups-per-protein launch lost its partner workgroup).  Called at the samplers' host sync points."""
        w = self.status()
        if w:
            # reported once: the word is cleared, so the model is usable again (a transient co-tenant must not poison every
            # later run); the launches since the failure produced nothing -- the kernels leave at entry while the word is set
            self.status_clear()
            raise RuntimeError(f"libdff_amd: device-side failure word {w:#x}: a two-workgroups-per-protein kernel launch timed "
                               f"out waiting for its partner workgroup (GPU shared or partitioned?); the results since then "
                               f"are invalid.  The word has been cleared; Model.pair(False) selects the one-workgroup kernels.")

    def poke_status(self, word: int):
        _check(self.lib, self.lib.dff_debug_poke_status(self.handle, int(word)), "dff_debug_poke_status")

    def status_clear(self):
        _check(self.lib, self.lib.dff_model_status_clear(self.handle), "dff_model_status_clear")

    def l0_table(self, on: bool = True):
        _check(self.lib, self.lib.dff_debug_l0_table(self.handle, int(on)), "dff_debug_l0_table")

    def last_launch(self):
        name, grid, lds = C.c_char_p(), C.c_int(), C.c_int()
        _check(self.lib, self.lib.dff_last_launch(self.handle, C.byref(name), C.byref(grid), C.byref(lds)), "dff_last_launch")
        return (name.value or b"").decode(), grid.value, lds.value

    # ---- the three entry points
    def score(self, x, tnorm, return_energy=False):
        import torch
        B = x.shape[0]
        self._dev_tensor(x, (B, self.n_beads, 3), "x")
        self._dev_tensor(tnorm, (B,), "t")
        force = torch.empty_like(x)
        energy = torch.empty(B, self.n_beads, device=x.device, dtype=torch.float32) if return_energy else None
        rc = self.lib.dff_score(self.handle, _ptr(x), _ptr(tnorm), B, _ptr(force), _ptr(energy), self._stream())
        _check(self.lib, rc, "dff_score")
        return (force, energy) if return_energy else force

    def langevin_run(self, params: DffLangevinParams, x, v, n_steps: int, save_interval: int,
                     noise=None, seed: int = 0, traj_offset: int = 0, step_offset: int = 0,
                     frames=None, ke=None):
        P = x.shape[0]
        self._dev_tensor(x, (P, self.n_beads, 3), "x")
        if v is not None:
            self._dev_tensor(v, (P, self.n_beads, 3), "v")
        if noise is not None:
            self._dev_tensor(noise, (n_steps, P, self.n_beads, 3), "noise")
        nf = n_steps // save_interval if save_interval > 0 else 0
        if frames is not None:
            self._dev_tensor(frames, (nf, P, self.n_beads, 3), "frames")
        if ke is not None:
            self._dev_tensor(ke, (nf, P), "ke")
        rc = self.lib.dff_langevin_run(self.handle, C.byref(params), P, _ptr(x), _ptr(v), _ptr(noise),
                                       seed & (2 ** 64 - 1), traj_offset, step_offset, n_steps, save_interval,
                                       _ptr(frames), _ptr(ke), self._stream())
        _check(self.lib, rc, "dff_langevin_run")

    def ddpm_run(self, x, t_start: int, t_end: int = 0, noise=None, seed: int = 0, sample_offset: int = 0,
                 init_prior: bool = False, clamp_flag=None):
        B = x.shape[0]
        self._dev_tensor(x, (B, self.n_beads, 3), "x")
        if noise is not None:
            self._dev_tensor(noise, (t_start - t_end + 1, B, self.n_beads, 3), "noise")
        rc = self.lib.dff_ddpm_run(self.handle, B, _ptr(x), _ptr(noise), seed & (2 ** 64 - 1), sample_offset,
                                   t_start, t_end, int(init_prior), _ptr(clamp_flag), self._stream())
        _check(self.lib, rc, "dff_ddpm_run")

    # ---- debugging
    PROFILE_STAGES = ("centre", "embed+ln1", "gemm_u", "gemm_qkv", "softmax", "pv+xrel", "gemm_wo", "gate1+ln2",
                      "gemm_w1+gelu", "gemm_w2", "gate2", "b_gate2", "b_gemm_w2T", "b_gemm_w1T", "b_ln2+gate1",
                      "b_gemm_woc+u", "b_reload+gemm_woT", "b_ds", "b_dx+dqkv", "b_gemm_qkvT", "b_ln1", "update",
                      "x22", "x23")

    def profile(self, enable: bool = True):
        _check(self.lib, self.lib.dff_debug_profile(self.handle, int(enable)), "dff_debug_profile")

    def profile_read(self) -> dict:
        out = np.zeros(24, np.uint64)
        _check(self.lib, self.lib.dff_debug_profile_read(self.handle, out.ctypes.data_as(C.c_void_p)), "dff_debug_profile_read")
        return {n: int(v) for n, v in zip(self.PROFILE_STAGES, out) if n != "-"}

    def debug_stash(self, b: int, layer: int, what: str) -> np.ndarray:
        N, H = self.n_beads, self.hidden
        items = dict(nodes_in=(0, (N, H)), attn_out=(1, (N, H)), ff=(2, (N, H)), h_pre=(3, (N, 4 * H)),
                     q=(4, (N, 512)), k=(5, (N, 512)), v=(6, (N, 512)), P=(7, (8, N, N)), u=(8, (N, 32)))
        code, shape = items[what]
        out = np.empty(shape, np.float32)
        rc = self.lib.dff_debug_stash(self.handle, b, layer, code, out.ctypes.data_as(C.c_void_p), out.size)
        _check(self.lib, rc, "dff_debug_stash")
        return out


def debug_gemm(A: np.ndarray, W: np.ndarray, device: int = 0) -> np.ndarray:
    lib = load_library()
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = A.shape
    K2, Nout = W.shape
    assert K == K2
    out = np.empty((M, Nout), np.float32)
    rc = lib.dff_debug_gemm(device, A.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), M, K, Nout,
                            out.ctypes.data_as(C.c_void_p))
    _check(lib, rc, "dff_debug_gemm")
    return out


# ---- PWD histograms (dff_pwd_*): stateless entry points, no model handle ----
def _coords(x):
    import torch
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and x.dim() == 3 and x.shape[-1] == 3):
        raise ValueError("structures must be a contiguous float32 CUDA tensor of shape (n, n_beads, 3)")
    return x


def pwd_num_pairs(n_beads: int, offset: int) -> int:
    return int(load_library().dff_pwd_num_pairs(int(n_beads), int(offset)))


def pwd_max(x, offset: int):
    """Per-pair maximum distance over the structures x (n, N, 3) -> float32 CUDA tensor (n_pairs,)."""
    import torch
    lib = load_library()
    x = _coords(x)
    n, N = int(x.shape[0]), int(x.shape[1])
    out = torch.empty(pwd_num_pairs(N, offset), dtype=torch.float32, device=x.device)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _check(lib, lib.dff_pwd_max(x.device.index, _ptr(x), n, N, int(offset), _ptr(out), stream), "dff_pwd_max")
    return out


def pwd_hist(x, offset: int, nbins, hmax):
    """Per-pair histograms (torch.histc semantics, min=0, max=hmax[p], bins=nbins[p]) of the
    pairwise distances of x (n, N, 3) -> int32 CUDA tensor (n_pairs, max(nbins)) of counts."""
    import torch
    lib = load_library()
    x = _coords(x)
    n, N = int(x.shape[0]), int(x.shape[1])
    npairs = pwd_num_pairs(N, offset)
    nb = torch.as_tensor(nbins, dtype=torch.int32).reshape(-1)
    hm = torch.as_tensor(hmax, dtype=torch.float32).reshape(-1)
    if nb.numel() != npairs or hm.numel() != npairs:
        raise ValueError(f"nbins / hmax must have {npairs} entries")
    if int(nb.min()) < 1:
        raise ValueError("nbins must be >= 1")
    max_bins = int(nb.max())
    nb_d, hm_d = nb.to(x.device), hm.to(x.device)
    out = torch.empty((npairs, max_bins), dtype=torch.int32, device=x.device)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _check(lib, lib.dff_pwd_hist(x.device.index, _ptr(x), n, N, int(offset), _ptr(nb_d), _ptr(hm_d), max_bins,
                                 max_bins, _ptr(out), stream), "dff_pwd_hist")
    return out
