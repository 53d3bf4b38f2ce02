"""Two Beyn passes at bench.py's default configuration with a marker kernel (a small triad) before and after the second one: the
kernel trace of this script (rocprofv3 --kernel-trace) shows where the GPU idles inside a pass (dev/probes/gap_trace.sh)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import wae_amd  # noqa
from wae_amd import _lib
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp.distributed import beyn_moments_distributed_rb, warm_up_dense_linalg
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
l, K, rb = 8, 2, int(os.environ.get('RB', 40))
L, pb = annulus_family(preset, tau=2e-4)
d = pb["d"]
L.solver_tol, L.solver_maxit, L.solver_ref = 1e-10, 400, 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1, "jacobi_weight": float(os.environ.get("JACW", 0.8)), "probe_columns": l, "snapshots": rb}
warm_up_dense_linalg(torch.device("cuda", 0), cols=l, K=K)
L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, l)) + 0j)
zmap = (complex(np.mean(G)), float(np.max(np.abs(G - np.mean(G)))))
def marker():
    g = C.c_double(0.0)
    _lib.check(_lib.lib().wae_bench_triad(0, 1 << 16, 1, C.byref(g)))
for it in range(2):
    marker()
    t0 = time.time()
    ph = {}
    buf, info = beyn_moments_distributed_rb(L, G, V, K, N, rb, timings=ph, zmap=zmap)
    torch.cuda.synchronize()
    print("pass", it, round(time.time() - t0, 4), {k: round(v, 4) for k, v in ph.items()}, info["iters_total"], flush=True)
marker()
