#!/usr/bin/env python3
"""Development prototype (CPU): progressive snapshot phase, chunk by chunk (4 points x ncol columns), with (a) per-column
bases and (b) per-column bases augmented by r POD vectors of ALL snapshot columns taken so far (shared pole directions).
Prints the guess errors per chunk and the iterations they imply (0.44 per iteration)."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_mg import annulus

pb = annulus.build("small", tau=2e-4)
T = pb["terms"]; d = pb["d"]
Y, n, tau = 1e15, 1.0, 2e-4
terms = [T["M"].tocsc(), T["K"].tocsc(), T["C"].tocsc(), T["Q"].tocsc()]
coefs = lambda z: [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
xg, wg = np.polynomial.legendre.leggauss(32)
Z = np.concatenate([(xg * (G[(i + 1) % 4] - G[i]) / 2 + (G[i] + G[(i + 1) % 4]) / 2) for i in range(4)])
ncol = 8
V = np.random.default_rng(7).standard_normal((d, ncol)) + 0j
S = 40
idx = np.unique(((np.arange(S) + 0.5) * len(Z) / S).astype(int))
bits = int(np.ceil(np.log2(len(idx))))
order = [idx[i] for i in np.argsort([int(format(i, f"0{bits}b")[::-1], 2) for i in range(len(idx))], kind="stable")]
t0 = time.time()
X = {}
for j in order:
    A = sum(c * t for c, t in zip(coefs(Z[j]), terms)).tocsc()
    X[j] = spla.splu(A).solve(V)
print("exact solves", time.time() - t0, flush=True)


def run(r_pod):
    Q = [np.zeros((d, 0), dtype=complex) for _ in range(ncol)]
    snaps = []
    its_total = 0.0
    for c0 in range(0, len(order), 4):
        chunk = order[c0:c0 + 4]
        pod = np.zeros((d, 0), dtype=complex)
        if r_pod and snaps:
            Sm = np.hstack(snaps)                              # all columns of all snapshots so far
            U, s, _ = np.linalg.svd(Sm, full_matrices=False)
            pod = U[:, :min(r_pod, (s > 1e-10 * s[0]).sum())]
        errs = []
        for j in chunk:
            A = sum(c * t for c, t in zip(coefs(Z[j]), terms)).tocsr()
            for c in range(ncol):
                B = np.hstack([Q[c], pod]) if pod.shape[1] else Q[c]
                if B.shape[1]:
                    B, _ = np.linalg.qr(B)
                    y = np.linalg.solve(B.conj().T @ (A @ B), B.conj().T @ V[:, c])
                    errs.append(np.linalg.norm(B @ y - X[j][:, c]) / np.linalg.norm(X[j][:, c]))
                else:
                    errs.append(1.0)
        errs = np.minimum(np.array(errs), 1.0)
        its = np.maximum(np.log(1e-10 / errs) / np.log(0.44), 0)
        its_total += its.max()                                  # lock-step: the slowest column of the chunk
        print("  chunk", c0 // 4 + 1, "guess err max %.1e median %.1e -> lock-step its %.0f" % (errs.max(), np.median(errs), its.max()), flush=True)
        for j in chunk:
            snaps.append(X[j])
            for c in range(ncol):
                q = X[j][:, c].copy()
                for _ in range(2):
                    q -= Q[c] @ (Q[c].conj().T @ q)
                Q[c] = np.hstack([Q[c], (q / np.linalg.norm(q))[:, None]])
    return its_total


for r in (0, 8, 16):
    print("POD vectors:", r)
    print(" total lock-step iterations of the snapshot phase: %.0f" % run(r), flush=True)
