"""State-dict handling: the reference checkpoint layout -> the flat fp32 array of the C ABI.

Checkpoint format (trainer.py:181-206, sample.py:154-167): ``torch.save({"step", "model", "ema",
"scaler", "opt", "scheduler", "best_val_loss"})``; sampling loads ``data["ema"]`` into
``EMA(GaussianDiffusion)``.  ``ema_pytorch`` is not a dependency here: the EMA state-dict is a
plain mapping whose ``ema_model.``-prefixed entries are the GaussianDiffusion state-dict
(13 schedule buffers + ``model.<GraphTransformer keys>``); we select those by name.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Tuple

import numpy as np

INNER = 512


def abi_key_order(n_layers: int) -> List[str]:
    """GraphTransformer keys in the order include/dff.h documents for dff_model_create."""
    keys = ["node_embedding.weight", "node_embedding.bias", "edge_embedding.weight",
            "edge_embedding.bias", "node_decoder.weight", "node_decoder.bias"]
    for l in range(n_layers):
        p = f"graphtransformer.layers.{l}."
        keys += [p + "0.0.fn.to_q.weight", p + "0.0.fn.to_q.bias", p + "0.0.fn.to_kv.weight",
                 p + "0.0.fn.to_kv.bias", p + "0.0.fn.edges_to_kv.weight", p + "0.0.fn.edges_to_kv.bias",
                 p + "0.0.fn.to_out.weight", p + "0.0.fn.to_out.bias", p + "0.0.norm.weight",
                 p + "0.0.norm.bias", p + "0.1.proj.0.weight", p + "1.0.fn.0.weight", p + "1.0.fn.0.bias",
                 p + "1.0.fn.2.weight", p + "1.0.fn.2.bias", p + "1.0.norm.weight", p + "1.0.norm.bias",
                 p + "1.1.proj.0.weight"]
    return keys


def expected_shapes(n_beads: int, hidden: int, n_layers: int, conservative: bool = True,
                    use_intrinsic_coords: bool = True, use_distances: bool = False,
                    use_abs_coords: bool = False) -> Dict[str, Tuple[int, ...]]:
    H, N, I, F = hidden, n_beads, INNER, 4 * hidden
    D = 1 if conservative else 3     # node_decoder: energy head or force head (graph_transformer.py:62-65)
    NI = N + 1 + 3 * bool(use_abs_coords)                                       # :53
    NE = 3 * bool(use_intrinsic_coords) + bool(use_distances) or 1              # :54-58
    s = {"node_embedding.weight": (H, NI), "node_embedding.bias": (H,), "edge_embedding.weight": (H, NE),
         "edge_embedding.bias": (H,), "node_decoder.weight": (D, H), "node_decoder.bias": (D,)}
    for l in range(n_layers):
        p = f"graphtransformer.layers.{l}."
        s.update({p + "0.0.fn.to_q.weight": (I, H), p + "0.0.fn.to_q.bias": (I,),
                  p + "0.0.fn.to_kv.weight": (2 * I, H), p + "0.0.fn.to_kv.bias": (2 * I,),
                  p + "0.0.fn.edges_to_kv.weight": (I, H), p + "0.0.fn.edges_to_kv.bias": (I,),
                  p + "0.0.fn.to_out.weight": (H, I), p + "0.0.fn.to_out.bias": (H,),
                  p + "0.0.norm.weight": (H,), p + "0.0.norm.bias": (H,), p + "0.1.proj.0.weight": (1, 3 * H),
                  p + "1.0.fn.0.weight": (F, H), p + "1.0.fn.0.bias": (F,), p + "1.0.fn.2.weight": (H, F),
                  p + "1.0.fn.2.bias": (H,), p + "1.0.norm.weight": (H,), p + "1.0.norm.bias": (H,),
                  p + "1.1.proj.0.weight": (1, 3 * H)})
    return s


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32)


def flatten_gnn_params(params: Mapping[str, object], n_beads: int, hidden: int, n_layers: int,
                       conservative: bool = True, use_intrinsic_coords: bool = True, use_distances: bool = False,
                       use_abs_coords: bool = False) -> np.ndarray:
    """GraphTransformer state-dict (keys without prefix) -> flat float32 array, shape-checked."""
    shapes = expected_shapes(n_beads, hidden, n_layers, conservative, use_intrinsic_coords, use_distances,
                             use_abs_coords)
    chunks = []
    for k in abi_key_order(n_layers):
        if k not in params:
            raise KeyError(f"missing parameter {k!r} in state dict")
        a = _np(params[k])
        if tuple(a.shape) != shapes[k]:
            raise ValueError(f"parameter {k!r} has shape {tuple(a.shape)}, expected {shapes[k]}")
        chunks.append(a.reshape(-1))
    return np.concatenate(chunks)


def gnn_params_from_checkpoint(data: Mapping[str, object]) -> Dict[str, object]:
    """Pick the GraphTransformer parameters out of a reference checkpoint dict.

    Accepts, in order of preference: ``data["ema"]`` with ``ema_model.model.*`` keys (what
    sample.py:154-167 loads), ``data["ema"]`` / ``data["model"]`` with ``model.*`` keys
    (a bare GaussianDiffusion state-dict), or a bare GraphTransformer state-dict.
    """
    cands = []
    if isinstance(data, Mapping):
        for top in ("ema", "model"):
            if top in data and isinstance(data[top], Mapping):
                cands.append(data[top])
        cands.append(data)
    for sd in cands:
        for prefix in ("ema_model.model.", "online_model.model.", "model.", ""):
            sel = {k[len(prefix):]: v for k, v in sd.items() if isinstance(k, str) and k.startswith(prefix)}
            if "node_embedding.weight" in sel and "node_decoder.weight" in sel:
                return sel
    raise KeyError("no GraphTransformer parameters (node_embedding.weight ...) found in checkpoint")
