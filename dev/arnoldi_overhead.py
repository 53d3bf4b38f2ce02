#!/usr/bin/env python3
"""Development check: fixed cost of one wae_arnoldi_shiftinvert call vs per-step cost (C2)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
L, pb = annulus_family("C2", tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
d = pb["d"]; T = len(L.terms)
z = 2 * np.pi * (600 + 40j)
cA = np.array(L.coefficients(z)); cM = np.zeros(T, dtype=complex); cM[-1] = -1
v0 = np.random.default_rng(0).standard_normal(d) + 0j
for m in (1, 2, 3, 6, 6):
    t = time.time(); H, V = fam.arnoldi(cA, cM, m, v0, tol=1e-10, maxit=400); dt = time.time() - t
    print("m", m, "wall %.3f s" % dt, "device %.3f s" % fam.last_info["seconds"], "its", fam.last_info["iters_total"], flush=True)
for r in (1, 1):
    b = np.random.default_rng(1).standard_normal((d, r)) + 0j
    t = time.time(); x = fam.solve(cA, b, tol=1e-10, maxit=400); dt = time.time() - t
    print("solve r", r, "wall %.3f s" % dt, "device %.3f" % fam.last_info["seconds"], "its", fam.last_info["iters_total"], flush=True)
