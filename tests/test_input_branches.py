"""The other input branches of the score net: use_intrinsic_coords /
use_distances / use_abs_coords as main_train.py can combine them (its defaults: 0 / 1 / 1).
CPU: the float64 factorised model with the hand-written VJP (oracle/kernel_model_gen.py -- the
algorithm the HIP kernel implements) against the reference's float64 run recorded by
tests/golden/make_golden_inputs.py.  GPU (-m gpu): the HIP path against the same vectors."""
import numpy as np
import pytest
import torch

from oracle import kernel_model_gen as kg
from oracle import synth

COMBOS = [(0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 0)]
CFGS = ["chignolin", "trp_cage"]
SEED = 2468


def params_for(cfg, intr, dist, ab):
    _, N, H, L = synth.SHIPPED_CONFIGS[cfg]
    return synth.synth_gnn_params(N, H, L, seed=SEED, node_in=N + 1 + 3 * ab, edge_in=3 * intr + dist), (N, H, L)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("flags", COMBOS)
def test_factorised_model_matches_reference_float64(cfg, flags, golden):
    intr, dist, ab = flags
    g = golden(f"score_in_{cfg}_{intr}{dist}{ab}.npz")
    p, (N, H, L) = params_for(cfg, intr, dist, ab)
    f, e = kg.score(p, g["x"], g["t"], L, bool(intr), bool(dist), bool(ab))
    assert np.abs(f - g["forces64"]).max() < 1e-11 * max(1.0, np.abs(g["forces64"]).max())
    assert np.abs(e - g["energy32"][..., 0]).max() < 1e-4 * max(1.0, np.abs(e).max())   # float32 reference energies


def test_general_model_reduces_to_the_shipped_branch(golden):
    """flags (1, 0, 0) through the general model == the dedicated model of the shipped checkpoints."""
    from oracle import kernel_model as km
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    p = synth.synth_gnn_params(N, H, L)
    g = golden("score_chignolin.npz")
    f1, e1 = km.score(p, g["x"], g["t"], L)
    f2, e2 = kg.score(p, g["x"], g["t"], L, True, False, False)
    assert np.abs(f1 - f2).max() < 1e-13 and np.abs(e1 - e2).max() < 1e-13
