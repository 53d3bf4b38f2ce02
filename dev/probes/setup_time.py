"""Where the solver set-up of the C3 family spends its time (WAE_SETUP_DEBUG=1 prints the library's own phase lines)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("WAE_SETUP_DEBUG", "1")
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol, L.solver_ref = 1e-10, 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
t0 = time.perf_counter()
fam = L.device()
t1 = time.perf_counter()
fam = L.ensure_solver()
t2 = time.perf_counter()
print("family upload %.3f s, solver set-up %.3f s" % (t1 - t0, t2 - t1), flush=True)
for rep in range(2):
    L._drop_device()
    t0 = time.perf_counter(); fam = L.device(); t1 = time.perf_counter(); fam = L.ensure_solver(); t2 = time.perf_counter()
    print("again: family upload %.3f s, solver set-up %.3f s" % (t1 - t0, t2 - t1), flush=True)
