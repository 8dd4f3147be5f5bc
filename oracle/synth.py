"""Re-export of synth_weights.py (repo root): the deterministic synthetic-weight generator shared by the parity tests,
the golden-vector generators, bench.py and the tools.  It holds no reference arithmetic, so it lives outside oracle/;
this module keeps ``from oracle import synth`` working for the test infrastructure."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from synth_weights import *  # noqa: F401,F403,E402
from synth_weights import SHIPPED_CONFIGS, count_params, normal, param_specs, splitmix64, synth_gnn_params, uniform  # noqa: F401,E402
