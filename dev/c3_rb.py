#!/usr/bin/env python3
"""1M-DoF (C3, BASELINE.json configs[2]: N = 64 per edge = 256 points, l = 8) full Beyn pass on ONE GPU with snapshot
projection: memory, time, iteration counts, eigenpair residuals."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices, moments2eigs, pos_test

t0 = time.time()
L, pb = annulus_family("C3", tau=2e-4)
d = pb["d"]
print("built", d, f"{time.time()-t0:.1f}s", flush=True)
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
t0 = time.time(); fam = L.ensure_solver(); print(f"setup {time.time()-t0:.1f}s", flush=True)
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
l = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
V = np.random.default_rng(7).standard_normal((d, l)) + 0j
t0 = time.time()
A = compute_moment_matrices(L, G, V, K=1, N=64, rb=rb)
dt = time.time() - t0
print(f"beyn 256 points x {l} columns, rb={rb}: {dt:.2f}s", fam.last_info, "free GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)
Om, P, S = moments2eigs(A, return_sigma=True)
Om, P = pos_test(Om, P, G)
r = fam.eig_residuals(np.array([L.coefficients(w) for w in Om]), P=P)
print("sigma", np.round(S, 6)); print("inside", np.round(Om / 2 / np.pi, 3)); print("res", r, flush=True)
