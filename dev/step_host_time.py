"""Where the host time of one Beyn pass goes (C2): Python-side preparation against the library's own clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import gauss_points
from wae_amd.nlevp.beyn import coefficient_table, snapshot_split, spread_order
from wae_amd.nlevp.distributed import beyn_moments_distributed_rb, moments2eigs_device

L, pb = annulus_family("C2", n=1.0, tau=2e-4)
d = pb["d"]
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zs, ws = gauss_points(G, 32)
    t1 = time.perf_counter()
    ct = coefficient_table(L, zs)
    t2 = time.perf_counter()
    ph = {}
    buf, info = beyn_moments_distributed_rb(L, G, V, 1, 32, 40, timings=ph)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    Om, Pd, S = moments2eigs_device(buf, (d, 16, 2))
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"gauss {1e3*(t1-t0):.2f} ms  coeff table {1e3*(t2-t1):.2f} ms  moments call {1e3*(t3-t2):.1f} ms (library clock {1e3*info['seconds']:.1f}, "
          f"phases {({k: round(1e3*v,1) for k,v in ph.items()})})  tail {1e3*(t4-t3):.1f} ms")
