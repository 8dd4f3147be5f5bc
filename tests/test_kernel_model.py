"""The float64 factorised model (oracle/kernel_model.py -- the algorithm the HIP kernels
implement) against the oracle twin and the reference's float64 golden vectors.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import kernel_model as km
from oracle import reference_twin as twin
from oracle import synth


@pytest.mark.parametrize("cfg", ["ala2", "chignolin", "trp_cage", "bba", "villin", "protein_g"])
def test_factorised_equals_materialised_fp64(cfg, golden):
    g = golden(f"score_{cfg}.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    params = synth.synth_gnn_params(N, H, L)
    f, e = km.score(params, g["x"], g["t"], L)
    scale = np.abs(g["forces64"]).max()
    assert np.abs(f - g["forces64"]).max() < 1e-12 * max(1.0, scale) + 1e-13
    assert np.abs(e[..., None] - g["energy64"]).max() < 1e-12


def test_intermediates_match_twin(golden):
    g = golden("layers_chignolin.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    params = synth.synth_gnn_params(N, H, L)
    fw = km.fold_weights(params, L)
    xc = g["x"].astype(np.float64)
    xc -= xc.mean(1, keepdims=True)
    _, st = km.forward(fw, xc, g["t"])
    for l in range(L):
        for name in ("attn_out", "nodes1", "ff", "nodes2"):
            np.testing.assert_allclose(st[l][name], g[f"l{l}.{name}"], rtol=0, atol=5e-6)
