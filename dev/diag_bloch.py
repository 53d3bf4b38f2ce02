#!/usr/bin/env python3
"""Development diagnostic: time the device mslp on the small Bloch unit cell per wave number."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.bloch import bloch_family
from wae_amd.nlevp import mslp, householder
import faulthandler
faulthandler.dump_traceback_later(100, exit=True)
cell = annulus.build_unit_cell(grid=(4, 26, 7), DOS=12, tau=2e-4)
L = bloch_family(cell)
L.solver_ref = 2 * np.pi * 400.0
for b in [0, 1, 2, 6]:
    L.params["b"] = b
    t = time.time()
    sol, n, flag = mslp(L, 2 * np.pi * 200.0, maxiter=20, tol=1e-10, output=True) if "-v" in sys.argv else mslp(L, 2 * np.pi * 200.0, maxiter=20, tol=1e-10)
    print(b, sol.params["ω"] / 2 / np.pi, n, flag, time.time() - t, L.device().last_info, flush=True)
