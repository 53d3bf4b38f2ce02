"""`householder` (one start value, vectors through host memory) against `householder_many([z])` (device-resident) at C3"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices, householder, householder_many, moments2eigs, pos_test
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol, L.solver_ref = 1e-10, 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
d = pb["d"]
V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, 8)) + 0j)
A = compute_moment_matrices(L, G, V, K=1, N=64 if preset == "C3" else 32)
Om, P = moments2eigs(A)
Om, P = pos_test(Om, P, G)
L.solver_tol = 1e-12
z0, v0 = Om[0] * (1 + 1e-5), np.ascontiguousarray(P[:, 0])
for name, fn in (("householder", lambda: householder(L, z0, maxiter=8, tol=1e-8 * 2 * np.pi, v0=v0, output=False)),
                 ("householder_many([z])", lambda: householder_many(L, [z0], maxiter=8, tol=1e-8 * 2 * np.pi, v0s=v0.reshape(d, 1))[0]),
                 ("householder, default start", lambda: householder(L, z0, maxiter=8, tol=1e-8 * 2 * np.pi, output=False)),
                 ("householder_many([z]), default start", lambda: householder_many(L, [z0], maxiter=8, tol=1e-8 * 2 * np.pi)[0])):
    t0 = time.perf_counter()
    sol, n, flag = fn()
    print("%-40s %.3f s  steps %d flag %d  omega/2pi %.9f%+.9fj" % (name, time.perf_counter() - t0, n, flag, (sol.params[L.eigval] / 2 / np.pi).real,
                                                                 (sol.params[L.eigval] / 2 / np.pi).imag), flush=True)
