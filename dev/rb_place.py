#!/usr/bin/env python3
"""Development check: snapshot placement uniform in arclength instead of uniform in quadrature index."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp.beyn import coefficient_table, gauss_points, snapshot_split, spread_order

L, pb = annulus_family("C2", tau=2e-4)
L.solver_tol = 1e-10
L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
d = pb["d"]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
fam = L.ensure_solver()
zs, ws = gauss_points(G, 32)
ct = coefficient_table(L, zs)
# arclength coordinate of every quadrature point along the closed polygon
s = np.zeros(len(zs)); acc = 0.0
for e in range(4):
    a, b = G[e], G[(e + 1) % 4]
    for i in range(32):
        s[e * 32 + i] = acc + abs(zs[e * 32 + i] - a)
    acc += abs(b - a)
def arclen_idx(S):
    tg = (np.arange(S) + 0.5) * acc / S
    idx = []
    for t in tg:
        order = np.argsort(np.abs(s - t))
        for j in order:
            if j not in idx:
                idx.append(int(j)); break
    return np.array(sorted(idx))
for mode, S in [("index", 32), ("arclen", 32), ("arclen", 28), ("arclen", 24), ("index", 24)]:
    idx = arclen_idx(S) if mode == "arclen" else snapshot_split(len(zs), S)[0]
    rest = np.setdiff1d(np.arange(len(zs)), idx)
    idx = spread_order(idx)
    for rep in range(2):
        t0 = time.time()
        fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V, 0, len(idx), tol=1e-10, maxit=400)
        t1 = time.time(); i0 = dict(fam.last_info)
        fam.beyn_moments_rb(zs[rest], ws[rest], ct[rest], V, 2, len(idx), tol=1e-10, maxit=400)
        t2 = time.time(); i1 = dict(fam.last_info)
    print(mode, S, "snapshots %.3f s (%d)" % (t1 - t0, i0["iters_total"]), "projected %.3f s (%d)" % (t2 - t1, i1["iters_total"]), "total %.3f" % (t2 - t0), flush=True)
