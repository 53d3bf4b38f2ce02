"""restriction / prolongation of level 0 timed alone (wae_bench_spmv_level which = 1, 2), by-tile kernels against the older ones"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import wae_amd  # noqa
from wae_amd import _lib
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol, L.solver_ref = 1e-10, 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
cz = np.ascontiguousarray(L.coefficients(2 * np.pi * (500 + 20j)), dtype=np.complex128)
for tiles in ("1", "0"):
    os.environ["WAE_XFER_TILES"] = tiles
    for which, name in ((1, "restriction"), (2, "prolongation")):
        ms, by = C.c_double(0), C.c_int64(0)
        best = 1e9
        for _ in range(3):
            _lib.check(_lib.lib().wae_bench_spmv_level(fam.handle, _lib.zptr(cz), which, 0, 64, 30, C.byref(ms), C.byref(by)))
            best = min(best, ms.value)
        print(f"WAE_XFER_TILES={tiles} {name}: {best * 1e3:.1f} us, {by.value / 1e9:.3f} GB algorithmic, {by.value / best / 1e9:.2f} TB/s", flush=True)
