"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's sampling hot path (microsoft/two-for-one-diffusion):

* ``reference_twin``  -- op-for-op PyTorch-CPU twin of ``models/graph_transformer.py``,
  ``models/ddpm.py`` (sampling half), ``dynamics/langevin.py`` and
  ``dynamics/langevin_cgnet.py`` in the reference's own *materialised* formulation
  (N x N x 512 edge tensor, ``torch.autograd`` for the force).  This is the parity oracle
  and the thing ``bench.py`` times as ``cpu_baseline`` (kind "port").
* ``kernel_model``    -- float64 numpy statement of the *factorised* algorithm the HIP
  kernels implement (edge tensor folded away, hand-written VJP), exposing every
  intermediate the kernel stashes, for stage-by-stage debugging of the device code.
* ``synth``           -- deterministic synthetic weights (splitmix64; no torch RNG) in the
  reference's state-dict layout.  The shipped checkpoints are absent from the reference
  mount (``/root/reference/.MISSING_LARGE_BLOBS:7-15``), so parity is pinned on these.

Pinning: the reference has no tests / golden vectors for this path.  The
twin is pinned against the reference ITSELF, imported in the build container, by
``tests/golden/make_golden.py``; the resulting input/output vectors are committed under
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks the twin against them on
every run (CPU, no reference needed).

Nothing under ``two-for-one-diffusion_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, as the checker.
"""
