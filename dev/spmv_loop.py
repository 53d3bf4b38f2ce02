import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd
from wae_amd.helmholtz.family import annulus_family
L, pb = annulus_family(sys.argv[1] if len(sys.argv) > 1 else "C3", tau=2e-4)
fam = L.device()
cz = L.coefficients(2 * np.pi * (500 + 20j))
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < 12:
    ms = fam.bench_spmv(cz, r=64, reps=500)
print("us", ms * 1e3, flush=True)
