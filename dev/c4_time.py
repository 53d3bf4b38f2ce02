#!/usr/bin/env python3
"""BASELINE.json configs[3] (C4): Bloch unit cell d = 200 000, DOS = 32.  Per wave number: Beyn estimates inside the
contour (snapshot projection), verified, then refined together by householder_many."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.bloch import bloch_family
from wae_amd.nlevp import compute_moment_matrices, moments2eigs, pos_test, householder_many

t0 = time.time()
cell = annulus.build_unit_cell(grid=annulus.PRESETS["C4"], DOS=32, tau=2e-4)
L = bloch_family(cell)
print("built", cell["nsector"], len(L.terms), "terms %.1f s" % (time.time() - t0), flush=True)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
t0 = time.time(); fam = L.ensure_solver(); print("setup %.1f s" % (time.time() - t0), flush=True)
d = cell["nsector"]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 8)) + 0j
for b in (0, 1, 2, 3):
    L.params["b"] = b
    t0 = time.time()
    A = compute_moment_matrices(L, G, V, K=1, N=32)
    t1 = time.time()
    info = dict(fam.last_info)
    Om, P, S = moments2eigs(A, return_sigma=True)
    Om, P = pos_test(Om, P, G)
    r = fam.eig_residuals(np.array([L.coefficients(w) for w in Om]), P=P) if len(Om) else np.zeros(0)
    good = r < 1e-5
    t2 = time.time()
    res = householder_many(L, Om[good], maxiter=8, tol=1e-8, v0s=P[:, good]) if good.any() else []
    t3 = time.time()
    print("b", b, "beyn %.2f s (%d col-its)" % (t1 - t0, info["iters_total"]), "tail %.2f" % (t2 - t1), "refine %.2f s" % (t3 - t2),
          "eigs/Hz", [np.round(s.params["ω"] / 2 / np.pi, 3) for s, n, f in res], "its", [n for s, n, f in res], flush=True)
