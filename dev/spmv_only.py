import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from wae_amd.helmholtz.family import annulus_family
L, pb = annulus_family(sys.argv[1] if len(sys.argv) > 1 else "C2", tau=2e-4)
r = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fam = L.device()
cz = L.coefficients(2 * np.pi * (500 + 20j))
ms = fam.bench_spmv(cz, r=r, reps=50)
print("r", r, "us", ms * 1e3, flush=True)
