"""One process, one C3 (or given) family: the fine-level operator product timed in each fused form (wae_bench_spmv, HIP events).
WAE_LIB_PATH selects an A/B build of the library.  usage: modes_ab.py PRESET r [modes...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
modes = [int(m) for m in sys.argv[3:]] or [0, 1, 2, 6]
L, pb = annulus_family(preset, tau=2e-4)
fam = L.device()
cz = L.coefficients(2 * np.pi * (500 + 20j))
names = {0: "A X", 1: "residual", 2: "Jacobi sweep", 6: "product + first sweep"}
out = {}
for m in modes:
    os.environ["WAE_BENCH_MODE"] = str(m)
    best = min(fam.bench_spmv(cz, r=r, reps=30) for _ in range(3))
    out[names.get(m, m)] = round(best * 1e3, 1)
print(os.environ.get("WAE_LIB_PATH", "default"), preset, r, out, flush=True)
