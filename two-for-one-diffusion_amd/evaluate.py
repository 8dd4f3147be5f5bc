"""Host-side mirror of the reference's pairwise-distance Jensen-Shannon metric
(evaluate/evaluators.py: PwdEvaluator :195-287, js_divergence :905-915, normalize_histogram
:918-924, kl_divergence :927-931, get_pwd_triu_batch :934-948) on top of the HIP kernels
dff_pwd_max / dff_pwd_hist (csrc/dff_pwd.hip).

Same names, arguments and results as the reference, with one difference of mechanism: the
reference materialises the (n, n_pairs) distance matrix and histograms its columns with
torch.histc on the CPU; here the structures stay on the GPU, only per-pair maxima and integer
histogram counts come back, and the (n_pairs x bins) Jensen-Shannon reduction runs in numpy
exactly as the reference writes it.  There is no CPU fallback: without the HIP library the
evaluator raises DffLibraryError.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from . import binding


# ---- evaluators.py:905-931, verbatim semantics (float32 in, numpy reductions) ----
def normalize_histogram(hist: np.ndarray) -> np.ndarray:
    hist = np.array(hist)
    return hist / np.sum(hist)


def kl_divergence(p1: np.ndarray, p2: np.ndarray):
    return np.sum(p1 * np.log(p1 / p2))


def js_divergence(h1: np.ndarray, h2: np.ndarray):
    p1 = normalize_histogram(h1) + 1e-10
    p2 = normalize_histogram(h2) + 1e-10
    M = (p1 + p2) / 2
    return (kl_divergence(p1, M) + kl_divergence(p2, M)) / 2


def nbins_for(maxval: torch.Tensor, resolution: float) -> torch.Tensor:
    """int(torch.div(m, resolution, rounding_mode="floor") + 1) for every pair (evaluators.py:242,259)."""
    return (torch.div(maxval.float().cpu(), resolution, rounding_mode="floor") + 1).to(torch.int64)


def _to_device(x, device):
    x = torch.as_tensor(x)
    if x.dim() != 3 or x.shape[-1] != 3:
        raise AssertionError("Shape mismatch")          # get_pwd_triu_batch's assert, evaluators.py:945
    return x.to(device=device, dtype=torch.float32).contiguous()


class PwdEvaluator:
    """Drop-in for evaluate.evaluators.PwdEvaluator (same constructor arguments and .eval()).

    val_data / all_mol are (n, N, 3) coordinate tensors in Angstrom (CPU or GPU).  `device` (extra,
    keyword-only) selects the GPU the histograms are computed on.
    """

    def __init__(self, val_data, plots_folder="", mol_name="", offset=0, saved_ref="none", evalset="testset",
                 *, device="cuda:0"):
        self.offset = offset
        self.plots_folder = plots_folder
        self.mol_name = mol_name.lower()
        self.resolution = 0.1
        self.device = torch.device(device)
        binding.load_library()      # fail loudly now rather than at the first eval()
        if saved_ref == "none":
            saved_ref = f"./saved_references/saved_pwd_{mol_name.upper()}_{evalset}_offset_{self.offset}.pickle"
        if os.path.exists(saved_ref):
            with open(saved_ref, "rb") as f:
                data = pickle.load(f)
            self.gt_max = data["gt_max"]
            self.gt_hist = data["gt_hist"]
        else:
            x = _to_device(val_data, self.device)
            self.gt_max = binding.pwd_max(x, self.offset).cpu()
            nb = nbins_for(self.gt_max, self.resolution)
            self.gt_hist = self._histograms(x, nb)
            d = os.path.dirname(saved_ref)
            if d == "" or os.path.isdir(d):
                with open(saved_ref, "wb") as f:
                    pickle.dump({"gt_max": self.gt_max, "gt_hist": self.gt_hist}, f)

    def _histograms(self, x, nbins):
        """list over pairs of float32 CPU tensors, what torch.histc(pwd[:, p], bins, 0, resolution*bins) returns"""
        hmax = torch.tensor([self.resolution * int(b) for b in nbins], dtype=torch.float64).float()
        counts = binding.pwd_hist(x, self.offset, nbins.to(torch.int32), hmax).cpu()
        return [counts[p, : int(b)].to(torch.float32) for p, b in enumerate(nbins)]

    def js_divergence_pwd(self, hist_gt, all_mol, gt_max, resolution):
        """Mean over pairs of JS(gt histogram, histogram of the sampled structures) (evaluators.py:251-270).
        Takes the structures (n, N, 3) where the reference takes their (n, n_pairs) distance matrix."""
        x = _to_device(all_mol, self.device)
        smax = binding.pwd_max(x, self.offset).cpu()
        maxval = torch.maximum(torch.as_tensor(gt_max).float().cpu(), smax)
        nb = (torch.div(maxval, resolution, rounding_mode="floor") + 1).to(torch.int64)
        hmax = torch.tensor([resolution * int(b) for b in nb], dtype=torch.float64).float()
        counts = binding.pwd_hist(x, self.offset, nb.to(torch.int32), hmax).cpu()
        result_js = np.empty(len(hist_gt))
        for i, (hgt, b) in enumerate(zip(hist_gt, nb)):
            b = int(b)
            hist_sampled = counts[i, :b].to(torch.float32)
            if b > len(hgt):
                hgt = torch.cat((hgt, torch.zeros(b - len(hgt))))
            result_js[i] = js_divergence(hgt.numpy(), hist_sampled.numpy())
        return result_js.mean()

    def eval(self, all_mol, plot_pwds=False, milestone=0):
        if plot_pwds:
            raise NotImplementedError("plotting (evaluators.py:289-349) is outside the hot path")
        return self.js_divergence_pwd(self.gt_hist, all_mol, self.gt_max, self.resolution)
