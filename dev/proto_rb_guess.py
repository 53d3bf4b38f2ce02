#!/usr/bin/env python3
"""Development prototype (CPU): how good is a Galerkin reduced-basis initial guess for the shifted systems of a Beyn
contour?  For one probe column v, solve L(z_j) x_j = v exactly at all quadrature points (sparse LU), then replay the
points in a spread-out order keeping an orthonormal basis Q of the snapshots taken so far: guess = Q (Q^H L(z) Q)^-1 Q^H v.
Prints the relative error of the guess at every point and whether the point had to be added to the basis."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_mg import annulus

preset = sys.argv[1] if len(sys.argv) > 1 else "small"
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
pb = annulus.build(preset, tau=2e-4)
T = pb["terms"]; d = pb["d"]
Y, n, tau = 1e15, 1.0, 2e-4
terms = [T["M"].tocsc(), T["K"].tocsc(), T["C"].tocsc(), T["Q"].tocsc()]
coefs = lambda z: [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
N = 32
xg, wg = np.polynomial.legendre.leggauss(N)
Z = np.concatenate([(xg * (G[(i + 1) % 4] - G[i]) / 2 + (G[i] + G[(i + 1) % 4]) / 2) for i in range(4)])
rng = np.random.default_rng(7)
ncol = 2
V = rng.standard_normal((d, ncol)) + 0j
t0 = time.time()
X = np.zeros((len(Z), d, ncol), dtype=complex)
for j, z in enumerate(Z):
    A = sum(c * t for c, t in zip(coefs(z), terms)).tocsc()
    X[j] = spla.splu(A).solve(V)
print("exact solves", time.time() - t0, flush=True)
# spread-out order: bit reversal over the 128 points
order = sorted(range(len(Z)), key=lambda j: int(format(j, "07b")[::-1], 2))
for joint in (False, True):
    print("joint basis over columns" if joint else "per-column basis")
    Q = [np.zeros((d, 0), dtype=complex) for _ in range(ncol)]
    added = 0
    errs = []
    for cnt, j in enumerate(order):
        z = Z[j]
        A = sum(c * t for c, t in zip(coefs(z), terms)).tocsr()
        for c in range(ncol):
            Qc = np.hstack(Q) if joint else Q[c]
            if Qc.shape[1]:
                if joint:
                    Qc, _ = np.linalg.qr(Qc)
                y = np.linalg.solve(Qc.conj().T @ (A @ Qc), Qc.conj().T @ V[:, c])
                x0 = Qc @ y
                err = np.linalg.norm(x0 - X[j, :, c]) / np.linalg.norm(X[j, :, c])
            else:
                err = 1.0
            errs.append(err)
            if err > thr:
                q = X[j, :, c].copy()
                B = Q[c]
                for _ in range(2):
                    q -= B @ (B.conj().T @ q)
                Q[c] = np.hstack([B, (q / np.linalg.norm(q))[:, None]])
                added += 1
        if cnt % 8 == 7 or cnt < 8:
            print(cnt + 1, "points; basis sizes", [q.shape[1] for q in Q], "last errs", ["%.1e" % e for e in errs[-ncol:]], flush=True)
    errs = np.array(errs)
    print("added", added, "of", len(errs), " median err of non-added", np.median(errs[errs <= thr]) if (errs <= thr).any() else None)
    # iterations saved estimate: GMRES at 0.44/iter from err -> 1e-10
    its = np.log(1e-10 / np.minimum(errs, 1.0)) / np.log(0.44)
    print("estimated GMRES iterations: mean %.1f (vs %.1f from zero guess)" % (np.maximum(its, 0).mean(), np.log(1e-10) / np.log(0.44)))
