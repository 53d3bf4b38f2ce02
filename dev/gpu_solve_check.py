import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from wae_amd.helmholtz.family import annulus_family
L, pb = annulus_family(sys.argv[1] if len(sys.argv) > 1 else "20k", tau=2e-4)
d = pb["d"]
L.solver_ref = 2 * np.pi * 500
if os.environ.get('NOQ'):
    w = 2 * np.pi * 500
    L.solver_ref_coeffs = [w * w, 1.0, w * 1e15, 0.0, 0.0]
rng = np.random.default_rng(0)
br = rng.standard_normal(d) + 0j
for sweeps in (1,):
    L.solver_opts = {"restart": 60, "sweeps": sweeps, "batch": 64}
    fam = L.device(); fam.solver_ready = False
    L.ensure_solver()
    for z in (2 * np.pi * (575 + 150j), 2 * np.pi * (1000 - 100j)):
        for nrhs in (1, 8):
            B = np.tile(br[:, None], (1, nrhs))
            X = L(z).solve(B, tol=1e-10, maxit=300)
            print(f"GPU z={z/2/np.pi:.0f} nu={sweeps} nrhs={nrhs}", fam.last_info, flush=True)
