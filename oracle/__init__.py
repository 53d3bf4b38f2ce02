"""CPU oracle for the NLEVP hot path of WavesAndEigenvalues.jl  --  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy restatement of the reference's algorithms
(operator-family evaluation, Beyn contour integral, Householder/MSLP/Newton
iterations, adjoint perturbation recurrences) plus the P1 Helmholtz
discretisation needed to regenerate the reference's tutorial problems.  Every
function cites the reference file:line it follows.

It exists to CHECK the HIP product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``wavesandeigenvalues.jl_amd/`` does.

Parity pin: the reference is Julia and cannot run in the build container (no
``julia`` binary); the reference has no test-suite.  The oracle is pinned by the
executed tutorial outputs the reference ships: see
``tests/test_oracle_golden.py``.  Results not covered by those recorded outputs
(e.g. Beyn on FEM problems beyond "two modes at 272 and 695 Hz") are
"parity unpinned" and documented as such where they are tested.
"""
