"""Deterministic synthetic weights in the reference's state-dict layout.

Not part of the oracle: no reference arithmetic here, only a seeded number generator and the reference's parameter names
and shapes.  It lives at the repo root so that bench.py and the tools_* scripts (product side) never import from
oracle/; the parity tests, the golden-vector generators and the oracle reach it through the re-export oracle/synth.py.

The reference's checkpoints (``saved_models/**/model-best.pt``) are not in the mount
(``/root/reference/.MISSING_LARGE_BLOBS:7-15``), so parity is pinned on synthetic weights.
They are produced by a counter-based splitmix64 stream in pure numpy integer arithmetic --
no torch / numpy RNG whose stream could change between versions -- so the golden-vector
generator (run once, next to the reference) and the tests (run anywhere) see identical bits.

Key names and shapes follow ``GaussianDiffusion.state_dict()`` of the reference
(``models/ddpm.py:23-138`` buffers, ``models/graph_transformer.py:23-75,211-227,273-316``
parameters; listed in SURVEY.md section 5 "checkpoint / resume").  Every tensor is perturbed
away from PyTorch's default init (LayerNorm gamma != 1, beta != 0, all biases != 0) so that
affine / bias bugs cannot hide.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

HEADS = 8
DIM_HEAD = 64
INNER = HEADS * DIM_HEAD  # 512, graph_transformer.py:218

# (mol name in args.pickle, n_beads, hidden_features_gnn, num_layers_gnn) of the shipped
# configs, read from saved_models/*/args.pickle (SURVEY.md section 8 header).
SHIPPED_CONFIGS = OrderedDict(
    ala2=("alanine_dipeptide_fuberlin", 5, 96, 2),
    chignolin=("CHIGNOLIN", 10, 64, 3),
    trp_cage=("TRP_CAGE", 20, 128, 3),
    bba=("BBA", 28, 96, 3),
    villin=("VILLIN", 35, 128, 3),
    protein_g=("PROTEIN_G", 56, 128, 3),
)

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser of (seed * golden + counter); uint64 in, uint64 out."""
    with np.errstate(over="ignore"):
        z = counter.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(shape, seed: int, stream: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float64 uniforms in [lo, hi) from stream ``stream`` of ``seed`` (53-bit mantissa)."""
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
    u = (splitmix64(ctr, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).reshape(shape)


def normal(shape, seed: int, stream: int) -> np.ndarray:
    """float64 standard normals (Box-Muller on two uniform streams)."""
    u1 = uniform(shape, seed, 2 * stream + 1_000_000, 0.0, 1.0)
    u2 = uniform(shape, seed, 2 * stream + 1_000_001, 0.0, 1.0)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def param_specs(n_beads: int, hidden: int, n_layers: int, decoder_out: int = 1, node_in: int = 0, edge_in: int = 3):
    """[(state-dict key under 'model.', shape, kind)] in the reference's registration order.

    kinds: 'w' Linear weight (fan_in = shape[1]), 'b' Linear bias (fan_in given), 'g' LN gamma,
    'be' LN beta.  Shapes: SURVEY.md section 5 (probe of the reference's own state_dict()).
    """
    H, N, I, F = hidden, n_beads, INNER, 4 * hidden
    NI = node_in if node_in else N + 1      # N + 1 + 3 * use_abs_coords
    specs = [
        ("node_embedding.weight", (H, NI), "w", NI),
        ("node_embedding.bias", (H,), "b", NI),
        ("edge_embedding.weight", (H, edge_in), "w", edge_in),
        ("edge_embedding.bias", (H,), "b", edge_in),
        ("node_decoder.weight", (decoder_out, H), "w", H),   # 1: energy head (conservative), 3: force head
        ("node_decoder.bias", (decoder_out,), "b", H),
    ]
    for l in range(n_layers):
        p = f"graphtransformer.layers.{l}."
        specs += [
            (p + "0.0.fn.to_q.weight", (I, H), "w", H),
            (p + "0.0.fn.to_q.bias", (I,), "b", H),
            (p + "0.0.fn.to_kv.weight", (2 * I, H), "w", H),
            (p + "0.0.fn.to_kv.bias", (2 * I,), "b", H),
            (p + "0.0.fn.edges_to_kv.weight", (I, H), "w", H),
            (p + "0.0.fn.edges_to_kv.bias", (I,), "b", H),
            (p + "0.0.fn.to_out.weight", (H, I), "w", I),
            (p + "0.0.fn.to_out.bias", (H,), "b", I),
            (p + "0.0.norm.weight", (H,), "g", H),
            (p + "0.0.norm.bias", (H,), "be", H),
            (p + "0.1.proj.0.weight", (1, 3 * H), "w", 3 * H),
            (p + "1.0.fn.0.weight", (F, H), "w", H),
            (p + "1.0.fn.0.bias", (F,), "b", H),
            (p + "1.0.fn.2.weight", (H, F), "w", F),
            (p + "1.0.fn.2.bias", (H,), "b", F),
            (p + "1.0.norm.weight", (H,), "g", H),
            (p + "1.0.norm.bias", (H,), "be", H),
            (p + "1.1.proj.0.weight", (1, 3 * H), "w", 3 * H),
        ]
    return specs


def synth_gnn_params(n_beads: int, hidden: int, n_layers: int, seed: int = 1234,
                     decoder_scale: float = 1.0, decoder_out: int = 1, node_in: int = 0,
                     edge_in: int = 3) -> "OrderedDict[str, np.ndarray]":
    """float32 GraphTransformer parameters keyed as in the reference state-dict (no prefix)."""
    out = OrderedDict()
    for stream, (key, shape, kind, fan_in) in enumerate(param_specs(n_beads, hidden, n_layers, decoder_out, node_in, edge_in)):
        bound = 1.0 / math.sqrt(fan_in)
        if kind in ("w", "b"):
            a = uniform(shape, seed, stream, -bound, bound)
        elif kind == "g":
            a = 1.0 + uniform(shape, seed, stream, -0.17, 0.17)
        else:
            a = uniform(shape, seed, stream, -0.17, 0.17)
        if key == "node_decoder.weight":
            a = a * decoder_scale
        out[key] = a.astype(np.float32)
    return out


def count_params(n_beads: int, hidden: int, n_layers: int) -> int:
    return sum(int(np.prod(s)) for _, s, _, _ in param_specs(n_beads, hidden, n_layers))
