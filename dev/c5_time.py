#!/usr/bin/env python3
"""BASELINE.json configs[4] (C5): 500k-DoF annulus, eigenpair from householder(tol=1e-11), then perturb_fast!(sol, L, :τ, 30)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import householder, perturb_fast_, conv_radius

t0 = time.time()
L, pb = annulus_family("C5", tau=2e-4)
print("built", pb["d"], "%.1f s" % (time.time() - t0), flush=True)
L.solver_tol = 1e-12
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 16, "restart": 40, "sweeps": 1}
t0 = time.time(); fam = L.ensure_solver(); print("setup %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
sol, n, flag = householder(L, 2 * np.pi * (195 + 9j), maxiter=10, tol=1e-11)
print("householder ->", sol.params["ω"] / 2 / np.pi, n, flag, "%.2f s" % (time.time() - t0), flush=True)
for N in (10, 30):
    t0 = time.time()
    perturb_fast_(sol, L, "τ", N)
    dt = time.time() - t0
    c = sol.eigval_pert["τ/Taylor"]
    print("perturb_fast order", N, "%.2f s" % dt, fam.last_info, "conv radius estimate (last 3)", conv_radius(c)[-3:], flush=True)
# check: Taylor prediction vs a re-solve at a perturbed delay
eps = 2e-4 * 1.05
w_pred = sol("τ", eps, 15, 15)
L.params["τ"] = eps
sol2, n2, f2 = householder(L, w_pred, maxiter=8, tol=1e-10, v0=sol.v, v0_adj=sol.v_adj)
print("Pade[15/15] prediction at tau*1.05:", w_pred / 2 / np.pi, " re-solved:", sol2.params["ω"] / 2 / np.pi, " rel diff %.2e" % (abs(w_pred - sol2.params["ω"]) / abs(w_pred)), flush=True)
