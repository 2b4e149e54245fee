"""CPU oracle for the differentiable-MD hot path.  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (PyTorch-CPU / numpy, fp32 or
fp64) of the reference algorithms on the north-star path: neighbour list, pair
potentials and their derivatives, Nose-Hoover-chain / NVE right-hand sides, the
NH-Verlet / Verlet fixed-grid steps, the reference's (non-textbook) adjoint
sweep, the soft-histogram RDF and the SchNet energy.  Every function cites the
reference file:line it restates.

Rules:
  * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
    import this package -- as the checker / timed baseline, never as the thing
    shipped.  Nothing under mdgrad_amd/ imports it; the product path raises
    when the HIP library is missing instead of falling back here.
  * Parity is PINNED: tests/test_oracle_golden.py checks every function here
    against golden vectors captured from the reference itself
    (tests/golden/*.npz, made by tests/golden/make_goldens.py in the build
    container).  Unpinned pieces (no reference implementation exists): the
    Yukawa pair form -- checked by finite differences only.
"""
from .md_oracle import *  # noqa: F401,F403
from .schnet_oracle import *  # noqa: F401,F403
