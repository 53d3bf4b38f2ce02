#!/usr/bin/env python3
"""Development check: where the solver set-up time goes (family upload, multigrid set-up)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
for preset in sys.argv[1:] or ["C2"]:
    t0 = time.time(); L, pb = annulus_family(preset, tau=2e-4); t1 = time.time()
    L.solver_tol = 1e-10; L.solver_ref = 2 * np.pi * 500.0
    L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    fam = L.device(); t2 = time.time()
    L.ensure_solver(); t3 = time.time()
    print(preset, "build %.2f s  upload/create %.2f s  solver set-up %.2f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
    L._drop_device()
