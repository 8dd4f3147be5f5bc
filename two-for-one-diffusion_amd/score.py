"""Host mirror of the reference's score network object (models/graph_transformer.py:18-114).

``GraphTransformer`` here owns a ``dff_model`` handle and exposes the reference's call signature
``model(x, h, t, return_energy=False, alphas=None)``; the arithmetic is the fused HIP kernel
behind ``dff_score`` (include/dff.h).  ``h`` (the bead one-hot) and ``alphas`` are accepted and
ignored exactly as far as the reference ignores them: ``h`` must be the identity the reference
passes (``bead_onehot = eye(N)``, datasets/dataset_utils_empty.py:218), ``alphas`` is unused
there too (graph_transformer.py:83-85).
"""
from __future__ import annotations

from typing import Mapping, Optional

import numpy as np
import torch

from . import binding, weights


class GraphTransformer:
    def __init__(self, num_beads: int, hidden_nf: int, device="cuda", n_layers: int = 4,
                 use_intrinsic_coords: bool = False, use_abs_coords: bool = True,
                 use_distances: bool = True, conservative: bool = True,
                 state_dict: Optional[Mapping[str, object]] = None, timesteps: int = 1000):
        if state_dict is None:
            raise ValueError("this sampler is inference-only: pass the trained parameters as state_dict")
        self.num_beads, self.hidden_nf, self.n_layers = num_beads, hidden_nf, n_layers
        self.use_intrinsic_coords, self.use_distances = use_intrinsic_coords, use_distances
        self.use_abs_coords, self.conservative = use_abs_coords, conservative
        dev = torch.device(device)
        if dev.type != "cuda":
            raise binding.DffLibraryError("the HIP score network needs a GPU device ('cuda[:i]'); there is no CPU path")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        flat = weights.flatten_gnn_params(state_dict, num_beads, hidden_nf, n_layers, conservative,
                                          use_intrinsic_coords, use_distances, use_abs_coords)
        self.native = binding.Model(num_beads, hidden_nf, n_layers, flat, timesteps=timesteps,
                                    device=self.device.index, use_intrinsic_coords=use_intrinsic_coords,
                                    use_distances=use_distances, use_abs_coords=use_abs_coords,
                                    conservative=conservative)
        self.training = False

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, x, h=None, t=None, return_energy: bool = False, alphas=None):
        return self.forward(x, h, t, return_energy=return_energy, alphas=alphas)

    def forward(self, x, h, t, return_energy: bool = False, alphas=None):
        if x.dim() != 3 or x.shape[-1] != 3 or x.shape[1] != self.num_beads:
            raise AssertionError("Dimensionality error")  # utils.py:69
        if h is not None and (tuple(h.shape) != (self.num_beads, self.num_beads)):
            raise ValueError("h must be the (N,N) bead one-hot")
        x = x.detach().to(self.device, torch.float32).contiguous()
        t = torch.as_tensor(t, dtype=torch.float32, device=self.device).reshape(-1)
        if t.numel() == 1 and x.shape[0] != 1:
            t = t.repeat(x.shape[0])
        t = t.contiguous()
        if return_energy:
            if not self.conservative:
                raise ValueError("a non-conservative model has no energy (graph_transformer.py:106-113)")
            _, e = self.native.score(x, t, return_energy=True)
            return e.unsqueeze(-1)
        return self.native.score(x, t)
