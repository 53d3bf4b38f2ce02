#!/usr/bin/env python3
"""Development check: Beyn moments with snapshot-projection guesses vs the plain path on an annulus preset."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices, moments2eigs, pos_test

preset = sys.argv[1] if len(sys.argv) > 1 else "small"
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
d = pb["d"]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
fam = L.ensure_solver()
for rb in [0] + [int(a) for a in sys.argv[2:]]:
    t = time.time()
    A = compute_moment_matrices(L, G, V, K=1, N=32, rb=rb)
    dt = time.time() - t
    if rb == 0:
        A0 = A
    Om, P, S = moments2eigs(A, return_sigma=True)
    Om, P = pos_test(Om, P, G)
    print("rb", rb, "time %.3f" % dt, "relerr vs plain %.2e" % (np.max(np.abs(A - A0)) / np.max(np.abs(A0))),
          {k: (round(v, 3) if isinstance(v, float) else v) for k, v in fam.last_info.items()}, "sigma gap %.1e" % (S[7] / S[8]), flush=True)
