#!/usr/bin/env python3
"""Development prototype (CPU): left-preconditioned GMRES(m) whose Krylov basis is STORED in complex64 while all
arithmetic stays in double ("compressed-basis GMRES").  Question: how many extra iterations does the rounding of the
stored basis cost at tol 1e-10?  Not part of the product or the oracle."""
import sys
import numpy as np
import scipy.sparse as sp
sys.argv = sys.argv[:1] + sys.argv[1:]
from proto_mg import MG, annulus


def lgmres(A, b, M, tol=1e-10, restart=40, maxit=300, store=np.complex128, reorth=False):
    n = len(b)
    x = np.zeros(n, dtype=complex)
    bn = np.linalg.norm(M(b))
    its = 0
    hist = []
    while its < maxit:
        r = M(b - A @ x)
        beta = np.linalg.norm(r)
        hist.append(beta / bn)
        if beta / bn < tol:
            break
        V = np.zeros((restart + 1, n), dtype=store)
        H = np.zeros((restart + 1, restart), dtype=complex)
        V[0] = (r / beta).astype(store)
        g = np.zeros(restart + 1, dtype=complex)
        g[0] = beta
        for j in range(restart):
            w = M(A @ V[j].astype(complex))
            for _ in range(2 if reorth else 1):
                Vj = V[:j + 1].astype(complex)
                h = Vj.conj() @ w
                w = w - Vj.T @ h
                H[:j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = (w / H[j + 1, j]).astype(store)
            its += 1
            y, *_ = np.linalg.lstsq(H[:j + 2, :j + 1], g[:j + 2], rcond=None)
            rn = np.linalg.norm(H[:j + 2, :j + 1] @ y - g[:j + 2])
            if rn / bn < 0.7 * tol or its >= maxit:
                break
        x = x + V[:j + 1].astype(complex).T @ y
    return x, its, hist


if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "20k"
    pb = annulus.build(preset, tau=2e-4)
    T = pb["terms"]
    d = pb["d"]
    Y, n, tau = 1e15, 1.0, 2e-4
    terms = [T["M"], T["K"], T["C"], T["Q"]]
    coefs = lambda z: [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
    wref = 2 * np.pi * 500
    S = -(T["K"].real + wref ** 2 * T["M"].real)
    mg = MG(sp.csr_matrix(S), terms, theta=0.02, max_coarse=128)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(d) + 0j
    for z in [2 * np.pi * (150 + 100j), 2 * np.pi * (400 + 150j), 2 * np.pi * (700 - 150j), 2 * np.pi * (1000 - 20j), 2 * np.pi * (575 + 150j)]:
        mg.setup(coefs(z), smoother="jac", nu=1, omega=0.8)
        A = mg.A[0]
        for store in (np.complex128, np.complex64):
            x, its, hist = lgmres(A, b, mg.vcycle, store=store)
            print(f"z/2pi={z / 2 / np.pi:.0f} store={store.__name__:10s} its={its} cycles={len(hist) - 1} hist={['%.1e' % v for v in hist]}")
