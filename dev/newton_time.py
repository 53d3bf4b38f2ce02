#!/usr/bin/env python3
"""Development check: where the time of a Newton-type refinement (householder from a Beyn estimate) goes at C2."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import householder, mslp
import wae_amd.nlevp.local_solvers as LS

L, pb = annulus_family("C2", tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
# instrument the device entry points
acc = {}
def wrap(name):
    f = getattr(fam, name)
    def g(*a, **k):
        t = time.time(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.time() - t; acc[name + "#"] = acc.get(name + "#", 0) + 1
        acc[name + "_dev"] = acc.get(name + "_dev", 0.0) + fam.last_info.get("seconds", 0.0) if name in ("arnoldi", "arnoldi_batch", "perturb", "solve") else 0.0
        acc[name + "_its"] = acc.get(name + "_its", 0) + (fam.last_info.get("iters_total", 0) if name in ("arnoldi", "arnoldi_batch", "perturb", "solve") else 0)
        return r
    setattr(fam, name, g)
for nm in ("arnoldi", "arnoldi_batch", "perturb", "solve", "spmv"):
    wrap(nm)
for start in (2 * np.pi * (737 + 3j), 2 * np.pi * (430 + 9j)):
    acc.clear()
    t = time.time()
    sol, n, flag = householder(L, start, maxiter=10, tol=1e-8)
    dt = time.time() - t
    print("householder ->", sol.params["ω"] / 2 / np.pi, n, flag, "%.2f s" % dt, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in acc.items()}, flush=True)
