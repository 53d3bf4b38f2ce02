#!/usr/bin/env python3
"""Development check: where the time of the snapshot-projection Beyn pass goes (C2)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp.beyn import coefficient_table, gauss_points, snapshot_split, spread_order

preset = sys.argv[1] if len(sys.argv) > 1 else "C2"
rb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
d = pb["d"]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
fam = L.ensure_solver()
zs, ws = gauss_points(G, 32)
ct = coefficient_table(L, zs)
idx, rest = snapshot_split(len(zs), rb)
idx = spread_order(idx)
cap = len(idx) + (int(sys.argv[3]) if len(sys.argv) > 3 else len(idx))
for rep in range(2):
    t0 = time.time()
    fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V, 0, cap, tol=1e-10, maxit=400)
    t1 = time.time(); i0 = dict(fam.last_info)
    fam.beyn_moments_rb(zs[rest], ws[rest], ct[rest], V, 2, cap, tol=1e-10, maxit=400)
    t2 = time.time(); i1 = dict(fam.last_info)
    print("rep", rep, "snapshots %.3f s (%d col-its)" % (t1 - t0, i0["iters_total"]), "projected %.3f s (%d col-its)" % (t2 - t1, i1["iters_total"]), flush=True)
