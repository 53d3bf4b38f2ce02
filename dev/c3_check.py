"""1M-DoF (C3) and 500k-DoF (C5) sanity runs: set-up time, memory, iteration counts, one short Beyn pass / perturbation."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices, gauss_points, moments2eigs, householder, perturb_fast_, pos_test
import torch
what = sys.argv[1]
t0 = time.time()
L, pb = annulus_family("C3" if what == "c3" else "C5", tau=2e-4)
d = pb["d"]
print("built", d, pb["info"], f"{time.time()-t0:.1f}s", flush=True)
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
t0 = time.time(); fam = L.ensure_solver(); print(f"setup {time.time()-t0:.1f}s", "mem GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)
cz = L.coefficients(2 * np.pi * (500 + 20j))
for r in (1, 8, 64):
    ms = fam.bench_spmv(cz, r=r, reps=10)
    print(f"spmv r={r}: {ms*1e3:.1f} us  {fam.spmv_bytes(r=r, mask=[1,1,1,1,0])/ms/1e6:.0f} GB/s", flush=True)
if what == "c3":
    G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    zs, ws = gauss_points(G, 64)
    V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
    t0 = time.time()
    A = compute_moment_matrices(L, G, V, K=1, N=64, points=(zs[::16], ws[::16]))     # 16 of the 256 points
    print(f"beyn 16 of 256 points: {time.time()-t0:.2f}s", fam.last_info, flush=True)
else:
    t0 = time.time()
    sol, n, flag = householder(L, 2 * np.pi * (195 + 9j), maxiter=10, tol=1e-8, output=True)
    print("householder", sol.params["ω"] / 2 / np.pi, n, flag, f"{time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    perturb_fast_(sol, L, "τ", 10)
    print("perturb_fast 10:", f"{time.time()-t0:.1f}s", fam.last_info, sol.eigval_pert["τ/Taylor"][:4], flush=True)
