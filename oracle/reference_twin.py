"""PyTorch-CPU twin of the reference's sampling hot path.  TEST INFRASTRUCTURE -- the oracle.

Op-for-op restatement, in the reference's own *materialised* formulation (the (B*8,N,N,64)
edge key/value tensors are built, the force is ``torch.autograd.grad`` of the summed energy),
of

* score network ............ models/graph_transformer.py:77-159,178-329
* schedule / DDPM sampler .. utils.py:33-39,52-70 ; models/ddpm.py:45-99,140-161,195-263
* force wrapper ............ dynamics/langevin.py:46-92,135-168
* BAOAB / Brownian step .... dynamics/langevin_cgnet.py:329-330,447-500,502-542,686-792

Functional (plain dict of tensors, no nn.Module) so that the same code runs float32 (the
parity oracle, and the CPU baseline ``bench.py`` times) and float64 (the accuracy yardstick).
Random draws are never made here: every stochastic step takes its noise as an argument, so
that the HIP path can be compared on identical draws.

Pinned against the reference itself by tests/golden/make_golden.py (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

HEADS = 8
DIM_HEAD = 64
KB = 0.83144626181  # dynamics/langevin.py:9  (g/mol, Angstrom, ps, K)
KBOLTZMANN = 1.38064852e-23  # dynamics/langevin.py:6-8
AVOGADRO = 6.022140857e23
JPERKCAL = 4184


# ----------------------------------------------------------------------------- helpers
def center_zero(x: torch.Tensor) -> torch.Tensor:
    """utils.py:65-70."""
    assert x.dim() == 3 and x.shape[-1] == 3
    return x - x.mean(dim=1, keepdim=True)


def to_torch(params: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in params.items()}


# ----------------------------------------------------------------------------- score net
def _attention(p, pre, nodes, edges):
    """graph_transformer.py:229-258 (mask is all-ones there: a no-op, :104,249-253)."""
    b, n, _ = nodes.shape
    q = F.linear(nodes, p[pre + "to_q.weight"], p[pre + "to_q.bias"])
    kv = F.linear(nodes, p[pre + "to_kv.weight"], p[pre + "to_kv.bias"])
    k, v = kv.chunk(2, dim=-1)
    e_kv = F.linear(edges, p[pre + "edges_to_kv.weight"], p[pre + "edges_to_kv.bias"])

    def split(t):  # "b ... (h d) -> (b h) ... d"
        lead = t.shape[1:-1]
        t = t.reshape(b, *lead, HEADS, DIM_HEAD)
        t = t.movedim(-2, 1)  # b h ... d
        return t.reshape(b * HEADS, *lead, DIM_HEAD)

    q, k, v, e_kv = split(q), split(k), split(v), split(e_kv)
    k = k.unsqueeze(1) + e_kv  # (bh,1,j,d)+(bh,i,j,d)
    v = v.unsqueeze(1) + e_kv
    sim = torch.einsum("b i d, b i j d -> b i j", q, k) * (DIM_HEAD ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("b i j, b i j d -> b i d", attn, v)
    out = out.reshape(b, HEADS, n, DIM_HEAD).permute(0, 2, 1, 3).reshape(b, n, HEADS * DIM_HEAD)
    return F.linear(out, p[pre + "to_out.weight"], p[pre + "to_out.bias"])


def _gated_residual(w, x, res):
    """graph_transformer.py:197-205."""
    gate = torch.sigmoid(F.linear(torch.cat((x, res, x - res), dim=-1), w))
    return x * gate + res * (1 - gate)


def energy(p: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, n_layers: int,
           intermediates: Optional[dict] = None, flags=(True, False, False)) -> torch.Tensor:
    """Per-bead energy (B,N,1) of an ALREADY centred x; graph_transformer.py:90-108,318-329.

    Shipped branch only: use_intrinsic_coords=True, use_distances=False, use_abs_coords=False
    (every saved_models/*/args.pickle), conservative=True.
    """
    b, n, _ = x.shape
    H = p["node_embedding.bias"].shape[0]
    tt = t.reshape(-1, 1, 1).to(x.dtype).repeat(1, n, 1)
    h = torch.eye(n, dtype=x.dtype).unsqueeze(0).repeat(b, 1, 1)
    intr, dist, abs_ = flags   # use_intrinsic_coords, use_distances, use_abs_coords (:53-58, :99-102, :116-140)
    diff = x.unsqueeze(1) - x.unsqueeze(2)  # diff[b,i,j] = x[b,j]-x[b,i]  (:125-129)
    if intr and dist:
        attr = torch.cat([diff, torch.sum(diff ** 2, dim=3, keepdim=True)], dim=3)
    elif dist:
        attr = torch.sum(diff ** 2, dim=3, keepdim=True)
    elif intr:
        attr = diff
    else:
        attr = torch.zeros(b, n, n, 1, dtype=x.dtype)
    edges = F.linear(attr, p["edge_embedding.weight"], p["edge_embedding.bias"])
    nodes_in = torch.cat((h, x, tt), dim=2) if abs_ else torch.cat((h, tt), dim=2)
    nodes = F.linear(nodes_in, p["node_embedding.weight"], p["node_embedding.bias"])
    for l in range(n_layers):
        pre = f"graphtransformer.layers.{l}."
        a = F.layer_norm(nodes, (H,), p[pre + "0.0.norm.weight"], p[pre + "0.0.norm.bias"], 1e-5)
        attn_out = _attention(p, pre + "0.0.fn.", a, edges)
        nodes1 = _gated_residual(p[pre + "0.1.proj.0.weight"], attn_out, nodes)
        f = F.layer_norm(nodes1, (H,), p[pre + "1.0.norm.weight"], p[pre + "1.0.norm.bias"], 1e-5)
        hid = F.gelu(F.linear(f, p[pre + "1.0.fn.0.weight"], p[pre + "1.0.fn.0.bias"]))
        ff = F.linear(hid, p[pre + "1.0.fn.2.weight"], p[pre + "1.0.fn.2.bias"])
        nodes2 = _gated_residual(p[pre + "1.1.proj.0.weight"], ff, nodes1)
        if intermediates is not None:
            intermediates[f"l{l}.nodes_in"] = nodes.detach()
            intermediates[f"l{l}.attn_out"] = attn_out.detach()
            intermediates[f"l{l}.nodes1"] = nodes1.detach()
            intermediates[f"l{l}.ff"] = ff.detach()
            intermediates[f"l{l}.nodes2"] = nodes2.detach()
        nodes = nodes2
    return F.linear(nodes, p["node_decoder.weight"], p["node_decoder.bias"])


def score(p: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, n_layers: int,
          return_energy: bool = False, conservative: bool = True, flags=(True, False, False)):
    """GraphTransformer.forward: graph_transformer.py:77-114 + compute_forces :143-159.

    x (B,N,3) need not be centred; returns -d(sum E)/d(x_centred), detached (eval mode:
    create_graph=False).  ``t`` is the normalised time t/T, shape (B,) / (B,1,1) / scalar.
    """
    xc = center_zero(x.detach()).requires_grad_(True)
    if t.numel() == 1:
        t = t.reshape(1).repeat(x.shape[0])
    if not conservative:   # force head: node_decoder is Linear(H, 3), forces = output (:62-65, :112-113)
        with torch.no_grad():
            return energy(p, xc.detach(), t, n_layers, flags=flags)
    with torch.enable_grad():
        e = energy(p, xc, t, n_layers, flags=flags)
        (grad,) = torch.autograd.grad(e, xc, grad_outputs=torch.ones_like(e))
    if return_energy:
        return -grad.detach(), e.detach()
    return -grad.detach()


# ----------------------------------------------------------------------------- schedule
def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """utils.py:52-62 (float64)."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def make_schedule(timesteps: int = 1000) -> Dict[str, torch.Tensor]:
    """The float32 buffers of GaussianDiffusion, models/ddpm.py:52-99 (computed in float64)."""
    betas = cosine_beta_schedule(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    bufs = dict(
        betas=betas,
        alphas_cumprod=ac,
        alphas_cumprod_prev=ac_prev,
        sqrt_alphas_cumprod=torch.sqrt(ac),
        sqrt_one_minus_alphas_cumprod=torch.sqrt(1.0 - ac),
        log_one_minus_alphas_cumprod=torch.log(1.0 - ac),
        sqrt_recip_alphas_cumprod=torch.sqrt(1.0 / ac),
        sqrt_recipm1_alphas_cumprod=torch.sqrt(1.0 / ac - 1),
        posterior_variance=pv,
        posterior_log_variance_clipped=torch.log(pv.clamp(min=1e-20)),
        posterior_mean_coef1=betas * torch.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
    )
    return {k: v.to(torch.float32) for k, v in bufs.items()}


# ----------------------------------------------------------------------------- DDPM
def p_sample(p, sched, x: torch.Tensor, t: int, noise: torch.Tensor, n_layers: int,
             timesteps: int = 1000) -> torch.Tensor:
    """One reverse step, models/ddpm.py:195-232, with the randn_like draw passed in.

    All samples of a batch share the integer timestep ``t`` (ddpm.py:244-247).
    """
    dt = x.dtype
    tt = torch.full((x.shape[0],), t, dtype=torch.long)
    eps = score(p, x, (1.0 * tt / timesteps).to(dt), n_layers)
    eps = center_zero(eps)
    g = lambda name: sched[name][t].to(dt)
    x0 = g("sqrt_recip_alphas_cumprod") * x - g("sqrt_recipm1_alphas_cumprod") * eps
    x0 = center_zero(x0)
    mean = g("posterior_mean_coef1") * x0 + g("posterior_mean_coef2") * x
    noise = center_zero(noise)
    nonzero = 0.0 if t == 0 else 1.0
    return mean + nonzero * (0.5 * g("posterior_log_variance_clipped")).exp() * noise


def p_sample_loop(p, sched, x_start: torch.Tensor, noises: torch.Tensor, t_start: int,
                  n_layers: int, timesteps: int = 1000) -> torch.Tensor:
    """models/ddpm.py:234-254 from x at time ``t_start`` down to 0; noises[k] feeds step k.

    x_start must already be centred (ddpm.py:242).  Returns normalised units (no norm_factor).
    """
    mol = x_start
    for k, i in enumerate(range(t_start, -1, -1)):
        mol = p_sample(p, sched, mol, i, noises[k], n_layers, timesteps)
        mol = torch.clamp(mol, min=-1000, max=1000)  # ddpm.py:248-250 (identity unless exceeded)
        mol = center_zero(mol)
    return mol


# ----------------------------------------------------------------------------- Langevin
def langevin_constants(norm_factor: float, t: int, sched, temp_data: float, temp_sim: float,
                       masses, friction: Optional[float] = 1.0, dt: Optional[float] = None,
                       kb: str = "consistent") -> dict:
    """Unit bookkeeping of LangevinDiffusion.__init__, dynamics/langevin.py:131-184, and of
    Langevin._input_option_checks, dynamics/langevin_cgnet.py:329-330,343."""
    one_minus_ac = 1 - sched["alphas_cumprod"][t].item()
    if kb == "consistent":
        kb_inv = 1 / KB * norm_factor ** 2
    elif kb == "kcal":
        kb_inv = JPERKCAL / KBOLTZMANN / AVOGADRO * (norm_factor ** 2) / 100
    else:
        raise Exception("Wrong kb value")
    friction_aux = 1 if friction is None else friction
    diffusion = 1 / masses[0] if friction is None else 1
    if dt is None:
        dt = one_minus_ac * friction_aux * masses[0] * kb_inv / temp_data
    out = dict(kb_inv=kb_inv, kbt_inv=kb_inv / temp_data, beta=kb_inv / temp_sim, dt=dt,
               sigma_t=sched["sqrt_one_minus_alphas_cumprod"][t].item(), t_norm=t / 1000.0,
               diffusion=diffusion, friction=friction)
    if friction is not None:
        out["vscale"] = float(np.exp(-dt * friction))
        out["noisescale"] = float(np.sqrt(1 - out["vscale"] * out["vscale"]))
    else:
        out["dtau"] = diffusion * dt
    return out


def forces(p, x: torch.Tensor, c: dict, n_layers: int, t_norm: Optional[float] = None) -> torch.Tensor:
    """ForcesWrapper.forward, dynamics/langevin.py:75-92: -GNN(x)/kbt_inv/sigma_t."""
    tn = torch.tensor([c["t_norm"] if t_norm is None else t_norm], dtype=torch.float32).to(x.dtype)
    return -score(p, x, tn, n_layers) / c["kbt_inv"] / c["sigma_t"]


def langevin_step(x_old, v_old, f, noise, masses: torch.Tensor, c: dict):
    """Langevin._langevin_timestep, dynamics/langevin_cgnet.py:447-479 (BAOA(F)B)."""
    m = masses[:, None]
    v_new = v_old + c["dt"] * f / m
    x_new = x_old + v_new * c["dt"] / 2.0
    nz = torch.sqrt(1.0 / c["beta"] / m) * noise
    v_new = v_new * c["vscale"]
    v_new = v_new + c["noisescale"] * nz
    x_new = x_new + v_new * c["dt"] / 2.0
    return x_new, v_new


def overdamped_step(x_old, f, noise, c: dict):
    """Langevin._overdamped_timestep, dynamics/langevin_cgnet.py:481-500."""
    return x_old + f * c["dtau"] + np.sqrt(2 * c["dtau"] / c["beta"]) * noise


def simulate(p, x0: torch.Tensor, noises: torch.Tensor, masses, c: dict, n_layers: int,
             save_interval: int):
    """Langevin.simulate, dynamics/langevin_cgnet.py:686-792, noise supplied as noises[step].

    x0 in normalised units (init_mol / norm_factor, langevin.py:135).  Returns
    (frames (n_sims, n_frames, N, 3), kinetic energies (n_sims, n_frames) or None, x, v).
    The saved frame is the UN-centred x_new (:521); centring happens at the top of the next
    step (:739).  v0 = 0 (:679).
    """
    length = noises.shape[0]
    assert length % save_interval == 0  # langevin_cgnet.py:305-309
    m = torch.as_tensor(masses, dtype=torch.float32).to(x0.dtype)
    x = x0
    v = None if c["friction"] is None else torch.zeros_like(x0)
    frames, kes = [], []
    for s in range(length):
        x = center_zero(x)
        f = forces(p, x, c, n_layers)
        if c["friction"] is None:
            x = overdamped_step(x, f, noises[s], c)
        else:
            x, v = langevin_step(x, v, f, noises[s], m, c)
        if (s + 1) % save_interval == 0:
            frames.append(x.clone())
            if v is not None:
                kes.append(0.5 * torch.sum(torch.sum(m[:, None] * v ** 2, dim=2), dim=1))
    frames = torch.stack(frames, 0).permute(1, 0, 2, 3).contiguous()
    kes = torch.stack(kes, 0).permute(1, 0).contiguous() if kes else None
    return frames, kes, x, v


def num_to_groups(num: int, divisor: int):
    """evaluate/evaluators.py:891-901."""
    arr = [divisor] * (num // divisor)
    if num % divisor > 0:
        arr.append(num % divisor)
    return arr
