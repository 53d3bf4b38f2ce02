#!/usr/bin/env python3
"""Development check: refine all eigenpairs of the C2 contour at once (householder_many) vs one by one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import householder, householder_many

L, pb = annulus_family("C2", tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
starts = 2 * np.pi * np.array([195 + 9j, 428.7 + 9.3j, 429.0 + 10.1j, 737.5 + 2.6j, 774.7 + 9.9j, 775.0 + 10.4j, 846.27 + 13.1j, 846.29 + 13.5j])
t = time.time()
many = householder_many(L, starts, maxiter=10, tol=1e-8)
tm = time.time() - t
print("many: %.2f s" % tm, [(np.round(s.params["ω"] / 2 / np.pi, 4), n, f) for s, n, f in many], flush=True)
t = time.time()
single = [householder(L, z0, maxiter=10, tol=1e-8) for z0 in starts[:3]]
print("single x3: %.2f s" % (time.time() - t), [(np.round(s.params["ω"] / 2 / np.pi, 4), n, f) for s, n, f in single], flush=True)
