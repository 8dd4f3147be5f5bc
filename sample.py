#!/usr/bin/env python3
"""Drop-in for the reference's ``sample.py``: ``python sample.py --model_path saved_models/chignolin
--gen_mode langevin --parallel_sim 256 ...`` (multi-GPU: ``torchrun --nproc-per-node G sample.py ...``)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dff_amd  # noqa: E402,F401
from dff_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
