"""cProfile of the Newton-type refinement of the C3 Beyn estimates (householder_many): where the HOST time goes between the library calls"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices, householder_many, moments2eigs, pos_test
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
res = fam.eig_residuals(np.array([L.coefficients(w) for w in Om]), P=P)
good = res <= 1e-6
Om, P = Om[good], np.asfortranarray(P[:, good])
L.solver_tol = 1e-12
stats = {}
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
outs = householder_many(L, list(Om), maxiter=6, tol=1e-8 * 2 * np.pi, v0s=P, stats=stats)
pr.disable()
print("householder_many(%d): %.3f s" % (len(Om), time.time() - t0), [(o[1], o[2]) for o in outs], stats, flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
if os.environ.get("NEWTON_SHOW_GRAM"):
    Vn = P / np.linalg.norm(P, axis=0)
    G = Vn.T @ Vn
    np.set_printoptions(precision=2, linewidth=200, suppress=True)
    print("eigenvalues / 2 pi:", Om / 2 / np.pi)
    print("|V^T V| =\n", np.abs(G))
    print("singular values of V^T V:", np.linalg.svd(G, compute_uv=False))
