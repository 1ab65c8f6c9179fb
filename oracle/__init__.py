"""CPU oracle for the DeltaConv message-passing path  --  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (pure PyTorch-CPU, fixed-degree "ELL"
formulation: every per-edge quantity is a dense ``[Nt, k, ...]`` tensor) of the algorithm in
``/root/reference/deltaconv/{geometry,nn,models}`` and ``experiments/utils.py``.  Every function
cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg -- as the checker / the timed CPU baseline, never as the thing shipped.  Nothing under
``deltaconv_amd/`` imports it; the product path raises if the HIP library is missing.

Pinning: the reference's tests hold no golden values for this path, so the
oracle is pinned against outputs of the reference itself, imported in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``), plus the reference's own analytic property tests re-run
against the oracle (``tests/test_oracle_properties.py``).

Third-party semantics that the reference leaves un-pinned and that this oracle DEFINES
(SURVEY.md section 8(c)):
  * kNN order: fp32 squared distance ``((dx*dx + dy*dy) + dz*dz)`` (no FMA), ascending, ties by
    lower point index, self included.
  * max-aggregation ties: the first maximal slot of the k-list receives the gradient.
  * SpMM summation order: ascending slot of the k-list.
  * estimate_basis sign: whatever LAPACK returns here; compared up to sign downstream.
"""
from . import geometry, nn, models, loss, fps  # noqa: F401
