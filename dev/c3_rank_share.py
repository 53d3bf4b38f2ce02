#!/usr/bin/env python3
"""One rank's share of the C3 snapshot phase when the probe columns are split over G GPUs (l/G columns, all 64 snapshot
points), timed on one GPU: what each of the G ranks does concurrently.  usage: c3_rank_share.py [G ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import gauss_points
from wae_amd.nlevp.beyn import coefficient_table, snapshot_split, spread_order

preset = os.environ.get("PRESET", "C3")
L, pb = annulus_family(preset, tau=2e-4)
d = pb["d"]
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
N, l, S = (64, 8, 40) if preset == "C3" else (32, 16, 40)
zs, ws = gauss_points(G, N)
ct = coefficient_table(L, zs)
idx, rest = snapshot_split(len(zs), S)
idx = spread_order(idx)
V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, l)) + 0j)
buf = torch.zeros(d * l * 2 * 2, dtype=torch.float64, device="cuda:0")
for world in [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]:
    ls = l // world
    local = torch.empty(S * d * ls * 2, dtype=torch.float64, device="cuda:0")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V[:, :ls], 0, S, Q_dev=local.data_ptr(), l_total=l, col0=0, K=1, tol=1e-10, maxit=400,
                            out_dev=buf.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    i = fam.last_info
    print(f"G={world}: {ls} column(s) x {S} snapshot points: {dt:.3f} s, column-iterations {i['iters_total']}, max {i['iters_max']}", flush=True)
    del local
