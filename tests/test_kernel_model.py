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


def test_fold_kv_is_the_same_network(golden):
    """hidden == head dimension (chignolin): k = v = LayerNorm output with W_k folded into W_q and W_v into W_o gives the
    same energies and forces in float64 (the library's dff_host.hip does this fold when hidden == 64)."""
    g = golden("layers_chignolin.npz")
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    params = synth.synth_gnn_params(N, H, L, decoder_scale=1.0)
    fw = km.fold_weights(params, L)
    ff = km.fold_kv(fw)
    xc = g["x"].astype(np.float64)
    xc -= xc.mean(1, keepdims=True)
    e0, s0 = km.forward(fw, xc, g["t"])
    e1, s1 = km.forward(ff, xc, g["t"])
    f0, f1 = -km.backward(fw, xc, s0), -km.backward(ff, xc, s1)
    assert np.abs(e1 - e0).max() <= 1e-11 * max(1.0, np.abs(e0).max())
    assert np.abs(f1 - f0).max() <= 1e-10 * np.abs(f0).max()
    for l in range(L):   # keys and values ARE the LayerNorm rows
        np.testing.assert_array_equal(s1[l]["k"], km._heads(np.tile(s1[l]["a"], (1, 1, km.HEADS))))
        np.testing.assert_allclose(s1[l]["attn_out"], s0[l]["attn_out"], rtol=0, atol=1e-11 * np.abs(s0[l]["attn_out"]).max())
