"""Host mirror of ``LangevinDiffusion`` / ``ForcesWrapper`` (dynamics/langevin.py:46-212) and of
the CGnet ``Langevin`` driver it wraps (dynamics/langevin_cgnet.py:168-237,686-792).

Unit bookkeeping stays on the host in float64 exactly as in the reference; the time loop --
centre, score network forward + VJP, force scaling, BAOA(F)B / Brownian update, noise, frame
and kinetic-energy capture -- runs in persistent HIP kernel launches (``dff_langevin_run``),
one launch per ``chunk`` of steps, with no host round trip inside a chunk.  (The reference draws
noise on the CPU generator and copies it H2D every step, langevin_cgnet.py:470-472.)
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import binding

KBOLTZMANN = 1.38064852e-23
AVOGADRO = 6.022140857e23
JPERKCAL = 4184
KB = 0.83144626181  # dynamics/langevin.py:9

# dynamics/langevin.py:11-26
temp_dict = {
    "ALANINE_DIPEPTIDE_FUBERLIN": 300, "ALANINE_DIPEPTIDE_MDSHARE": 300, "CHIGNOLIN": 340, "TRP_CAGE": 290,
    "BBA": 325, "VILLIN": 360, "WW_DOMAIN": 360, "NTL9": 355, "BBL": 298, "PROTEIN_B": 340,
    "HOMEODOMAIN": 360, "PROTEIN_G": 350, "ALPHA3D": 370, "LAMBDA_REPRESSOR": 350,
}


class ForcesWrapper:
    """dynamics/langevin.py:46-92: the diffusion model as a force field,
    ``forward(x) -> (zeros(B) on CPU, -GNN(x, onehot, t/T) / kbt_inv / sigma_t)``."""

    def __init__(self, model_diff, t=10, diffusion_steps=1000, kbt_inv=1.0):
        self.model_gnn = model_diff.model.eval()
        self.t = int(t)
        self.sqrt_one_minus_alphas_cumprod = float(model_diff.sqrt_one_minus_alphas_cumprod[t].item())
        self.t_norm = float(np.float32(t) / np.float32(diffusion_steps))
        self.kbt_inv = kbt_inv
        self.one_hot = model_diff.h
        self.training = False

    @property
    def force_scale(self) -> float:
        return 1.0 / (self.kbt_inv * self.sqrt_one_minus_alphas_cumprod)

    def __call__(self, x_old, embeddings=None):
        tn = torch.full((x_old.shape[0],), self.t_norm, dtype=torch.float32, device=x_old.device)
        forces = -self.model_gnn(x_old, self.one_hot, tn) / self.kbt_inv / self.sqrt_one_minus_alphas_cumprod
        return torch.zeros(x_old.shape[0]), forces


class LangevinDiffusion:
    """dynamics/langevin.py:95-212.  Same constructor arguments and ``sample()`` contract:
    returns a float32 CPU tensor (P * n_timesteps / save_interval, N, 3) in Angstrom,
    simulation-major (all frames of sim 0, then sim 1, ...)."""

    def __init__(self, model_diff, init_mol, n_timesteps=1000000, save_interval=250, t=15,
                 diffusion_steps=1000, temp_data=300, temp_sim=300, dt=2e-3, masses=[12.8] * 5,
                 friction=1, kb="consistent", exchange_interval=5000, seed: int = 0,
                 chunk: Optional[int] = None, verbose: bool = True):
        self.norm_factor = model_diff.norm_factor
        self.device = model_diff.device
        self.native = model_diff.model.native
        self.n_beads = model_diff.num_atoms
        init_sample = torch.as_tensor(init_mol, dtype=torch.float32) / self.norm_factor
        if init_sample.dim() != 3:
            raise ValueError("initial_coordinates shape must be [frames, beads, dimensions]")
        self.one_minus_alphas_cumprod = 1 - model_diff.alphas_cumprod[t].item()
        if kb == "consistent":
            self.kb_inv = 1 / KB * self.norm_factor ** 2
        elif kb == "kcal":
            self.kb_inv = JPERKCAL / KBOLTZMANN / AVOGADRO * (self.norm_factor ** 2) / 100
        else:
            raise Exception("Wrong kb value")
        self.model_forces = ForcesWrapper(model_diff, t, diffusion_steps, kbt_inv=self.kb_inv / temp_data)
        if friction is None:
            friction_aux, diffusion_constant = 1, 1 / masses[0]
        else:
            friction_aux, diffusion_constant = friction, 1
        if dt is None:
            dt = self.one_minus_alphas_cumprod * friction_aux * masses[0] * self.kb_inv / temp_data
        # --- Langevin._input_option_checks (langevin_cgnet.py:275-398) ---
        if n_timesteps % save_interval != 0:
            raise ValueError("The save_interval must be a factor of the simulation length")
        if friction is not None:
            if masses is None:
                raise RuntimeError("if friction is not None, masses must be given")
            if len(masses) != init_sample.shape[1]:
                raise ValueError("mass list length must be number of CG beads")
        if init_sample.shape[1] != self.n_beads:
            raise ValueError("initial coordinates do not match the model's number of beads")
        self.length, self.save_interval = int(n_timesteps), int(save_interval)
        self.friction, self.masses, self.dt = friction, list(masses), float(dt)
        self.beta = self.kb_inv / temp_sim
        self.diffusion = diffusion_constant
        self.n_sims = init_sample.shape[0]
        p = binding.DffLangevinParams()
        p.t_norm = self.model_forces.t_norm
        p.force_scale = self.model_forces.force_scale
        p.dt = self.dt
        p.beta = self.beta
        if friction is not None:
            self.vscale = float(np.exp(-self.dt * friction))                 # langevin_cgnet.py:329
            self.noisescale = float(np.sqrt(1 - self.vscale * self.vscale))   # :330
            p.vscale, p.noisescale, p.overdamped, p.dtau = self.vscale, self.noisescale, 0, 0.0
        else:
            self._dtau = self.diffusion * self.dt                             # :343
            p.vscale, p.noisescale, p.overdamped, p.dtau = 0.0, 0.0, 1, self._dtau
        for i, mval in enumerate(self.masses):
            p.masses[i] = mval
        self.params = p
        self.seed = int(seed)
        self.chunk = chunk
        self.verbose = verbose
        self.x = init_sample.to(self.device).contiguous()
        self.v = torch.zeros_like(self.x)                                     # :679
        self.t = 0
        self.kinetic_energies = None
        if verbose:
            print(f"norm factor:{self.norm_factor}")
            print(f"dt: {self.dt: .8f} (ps)")
            print(f"KbT: {temp_data / self.kb_inv: .4f}")

    def simulate(self, noises=None, traj_offset: int = 0):
        """Langevin.simulate (langevin_cgnet.py:686-792).  Returns a numpy array
        (n_sims, n_frames, N, 3) in normalised units; ``noises`` (length, n_sims, N, 3) switches
        to supplied standard normals (parity mode) instead of in-kernel Philox."""
        n_frames = self.length // self.save_interval
        frames = torch.empty(n_frames, self.n_sims, self.n_beads, 3, device=self.device, dtype=torch.float32)
        ke = None if self.friction is None else torch.empty(n_frames, self.n_sims, device=self.device, dtype=torch.float32)
        chunk = self.chunk or self.length
        chunk = max(self.save_interval, (chunk // self.save_interval) * self.save_interval)
        if noises is not None:
            noises = noises.detach().to(self.device, torch.float32).contiguous()
        done = 0
        while done < self.length:
            n = min(chunk, self.length - done)
            f0 = done // self.save_interval
            self.native.langevin_run(
                self.params, self.x, None if self.friction is None else self.v, n, self.save_interval,
                noise=None if noises is None else noises[done:done + n], seed=self.seed,
                traj_offset=traj_offset, step_offset=self.t,
                frames=frames[f0:f0 + n // self.save_interval],
                ke=None if ke is None else ke[f0:f0 + n // self.save_interval])
            done += n
            self.t += n
            if self.verbose:
                print(f"{done // self.save_interval}/{n_frames} time points saved")
        self.kinetic_energies = None if ke is None else ke.permute(1, 0).cpu().numpy()
        self.simulated_coords = frames.permute(1, 0, 2, 3).cpu().numpy()      # :605-629
        self.native.check()   # the copies above synchronised: any device-side failure word of the launches is final now
        return self.simulated_coords

    def sample(self, noises=None, traj_offset: int = 0):
        traj = torch.from_numpy(np.ascontiguousarray(self.simulate(noises, traj_offset)))
        traj = traj.reshape(-1, traj.size(2), traj.size(3))
        return traj * self.norm_factor                                       # langevin.py:209-211
