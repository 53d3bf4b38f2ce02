#!/usr/bin/env python3
"""Development diagnostic: mslp on the full-size C4 unit cell."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.bloch import bloch_family
from wae_amd.nlevp import mslp
import faulthandler
faulthandler.dump_traceback_later(400, exit=True)
cell = annulus.build_unit_cell(grid=annulus.PRESETS["C4"], DOS=32, tau=2e-4)
L = bloch_family(cell)
L.solver_ref = 2 * np.pi * 500.0
L.solver_tol = 1e-10
L.solver_opts = {**L.solver_opts, "batch": 16, "restart": 40, "sweeps": 1}
if "--noexcl" in sys.argv:
    L.solver_opts.pop("shape_exclude")
fam = L.ensure_solver()
d = cell["nsector"]
rng = np.random.default_rng(0)
x = rng.standard_normal((d, 1)) + 0j
for b, f0 in [(0, 200.0), (1, 430.0), (5, 450.0), (16, 450.0)]:
    L.params["b"] = b
    for z in (2 * np.pi * f0, 2 * np.pi * (f0 + 30j)):
        t = time.time()
        L(z).solve(x, tol=1e-10)
        print("solve b", b, "z/2pi", z / 2 / np.pi, fam.last_info, round(time.time() - t, 2), flush=True)
    if b > 1:
        continue
    t = time.time()
    sol, n, flag = mslp(L, 2 * np.pi * f0, maxiter=15, tol=1e-8)
    print("mslp b", b, f0, "->", sol.params["ω"] / 2 / np.pi, n, flag, round(time.time() - t, 1), flush=True)
