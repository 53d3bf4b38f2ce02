#!/usr/bin/env python3
"""Where along the contour the projected phase spends its iterations (C3): mode 2 called chunk by chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import gauss_points
from wae_amd.nlevp.beyn import coefficient_table, snapshot_split, spread_order

preset = os.environ.get("PRESET", "C3")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L, pb = annulus_family(preset, tau=2e-4)
d = pb["d"]
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
N, l = (64, 8) if preset == "C3" else (32, 16)
zs, ws = gauss_points(G, N)
ct = coefficient_table(L, zs)
P = float(os.environ.get("P", "0"))          # snapshot density ~ |z|^P along the contour (0 = uniform in the point index)
if P > 0:
    wgt = np.abs(zs) ** P
    cum = np.cumsum(wgt) - 0.5 * wgt
    targets = (np.arange(S) + 0.5) * wgt.sum() / S
    idx = np.unique(np.searchsorted(cum, targets).clip(0, len(zs) - 1))
    rest = np.setdiff1d(np.arange(len(zs)), idx)
    S = len(idx)
else:
    idx, rest = snapshot_split(len(zs), S)
idx = spread_order(idx)
V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, l)) + 0j)
buf = torch.zeros(d * l * 2 * 2, dtype=torch.float64, device="cuda:0")
kw = dict(K=1, tol=1e-10, maxit=400, out_dev=buf.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V, 0, S, **kw)
torch.cuda.synchronize(); print(f"P={P} S={S} snapshot phase {time.perf_counter() - t0:.3f} s (first call, includes warm-up)")
torch.cuda.synchronize(); t0 = time.perf_counter()
fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V, 0, S, **kw)
torch.cuda.synchronize(); print(f"P={P} S={S} snapshot phase {time.perf_counter() - t0:.3f} s")
print("snapshot indices", sorted(idx.tolist()))
spc = 64 // l
tot = 0.0
for c in range(0, len(rest), spc):
    pts = rest[c:c + spc]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fam.beyn_moments_rb(zs[pts], ws[pts], ct[pts], None, 2, S, accumulate=True, l_total=l, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot += dt
    i = fam.last_info
    print(f"points {pts[0]:3d}..{pts[-1]:3d}  z/2pi ~ {zs[pts[0]]/2/np.pi:.0f}  its max {i['iters_max']:2d} total {i['iters_total']:4d}  {1e3*dt:6.1f} ms")
print("projected total", tot)
