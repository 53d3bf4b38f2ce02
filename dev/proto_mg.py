#!/usr/bin/env python3
"""Development prototype (numpy/scipy, CPU): smoothed-aggregation multigrid + right-preconditioned GMRES for
L(z) x = b on the synthetic annulus.  Used only to choose algorithmic parameters (aggregation threshold,
smoother, cycle) before writing the HIP solver; not part of the product or the oracle."""
import importlib.util
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("annulus", os.path.join(HERE, "..", "wavesandeigenvalues.jl_amd", "helmholtz", "annulus.py"))
annulus = importlib.util.module_from_spec(spec)
spec.loader.exec_module(annulus)


def strength(S, theta):
    S = S.tocsr()
    D = np.abs(S.diagonal())
    C = S.tocoo()
    keep = (C.row != C.col) & (np.abs(C.data) >= theta * np.sqrt(D[C.row] * D[C.col]))
    return sp.csr_matrix((np.ones(keep.sum()), (C.row[keep], C.col[keep])), shape=S.shape)


def aggregate(G):
    """standard greedy aggregation; returns agg id per node (-1 never)."""
    n = G.shape[0]
    indptr, indices = G.indptr, G.indices
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    # pass 1
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = indices[indptr[i]:indptr[i + 1]]
        if len(nb) and np.all(agg[nb] < 0):
            agg[i] = na
            agg[nb] = na
            na += 1
    # pass 2: attach to a neighbouring aggregate
    agg2 = agg.copy()
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = indices[indptr[i]:indptr[i + 1]]
        a = agg[nb]
        a = a[a >= 0]
        if len(a):
            agg2[i] = a[0]
    agg = agg2
    # pass 3: leftovers
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = indices[indptr[i]:indptr[i + 1]]
        agg[i] = na
        for j in nb:
            if agg[j] < 0:
                agg[j] = na
        na += 1
    return agg, na


def sa_level(S, theta, omega_p=4.0 / 3.0, smooth=True):
    G = strength(S, theta)
    agg, na = aggregate(G)
    n = S.shape[0]
    Pt = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
    if not smooth:
        return Pt
    # filtered matrix: drop weak off-diagonals, lump them to the diagonal
    Sc = S.tocoo()
    Gd = G.todok() if False else None
    Gs = G + sp.identity(n, format="csr")
    F = S.multiply(Gs).tocsr()
    lump = np.asarray(S.sum(axis=1)).ravel() - np.asarray(F.sum(axis=1)).ravel()
    F = F + sp.diags(lump)
    Dinv = 1.0 / F.diagonal()
    DA = sp.diags(Dinv) @ F
    # spectral radius estimate
    x = np.random.default_rng(0).standard_normal(n)
    for _ in range(15):
        x = DA @ x
        x /= np.linalg.norm(x)
    rho = np.linalg.norm(DA @ x)
    P = Pt - (omega_p / rho) * (DA @ Pt)
    return sp.csr_matrix(P)


class MG:
    def __init__(self, S, terms, theta=0.08, max_coarse=600, max_levels=8, smooth=True):
        """terms: list of fine matrices A_k (complex csr).  Galerkin-project every term."""
        self.P = []
        self.terms = [[t.tocsr() for t in terms]]
        Sl = S.tocsr()
        while Sl.shape[0] > max_coarse and len(self.P) < max_levels:
            P = sa_level(Sl, theta, smooth=smooth)
            if P.shape[1] >= 0.8 * P.shape[0]:
                break
            self.P.append(P)
            R = P.T.tocsr()
            self.terms.append([sp.csr_matrix(R @ t @ P) for t in self.terms[-1]])
            Sl = sp.csr_matrix(R @ Sl @ P)
            print("  level", len(self.P), "n", P.shape[1], "nnz/row", self.terms[-1][0].nnz / P.shape[1])
        self.sizes = [t[0].shape[0] for t in self.terms]

    def setup(self, coefs, smoother="l1", nu=1, omega=0.7):
        self.A = [sum(c * t for c, t in zip(coefs, ts)).tocsr() for ts in self.terms]
        self.nu = nu
        self.dinv = []
        for A in self.A[:-1]:
            if smoother == "l1":
                dl1 = np.asarray(abs(A).sum(axis=1)).ravel()
                self.dinv.append(1.0 / dl1 * (A.diagonal() / np.abs(A.diagonal())).conj() * 0 + 1.0 / (dl1 * A.diagonal() / np.abs(A.diagonal())))
            else:
                self.dinv.append(omega / A.diagonal())
        self.lu = spla.splu(sp.csc_matrix(self.A[-1]))

    def vcycle(self, b, lvl=0):
        if lvl == len(self.A) - 1:
            return self.lu.solve(b)
        A, dinv = self.A[lvl], self.dinv[lvl]
        x = dinv * b
        for _ in range(self.nu - 1):
            x = x + dinv * (b - A @ x)
        r = b - A @ x
        xc = self.vcycle(self.P[lvl].T @ r, lvl + 1)
        x = x + self.P[lvl] @ xc
        for _ in range(self.nu):
            x = x + dinv * (b - A @ x)
        return x


def fgmres(A, b, M, tol=1e-10, restart=50, maxit=300):
    n = len(b)
    x = np.zeros(n, dtype=complex)
    bn = np.linalg.norm(b)
    its = 0
    while its < maxit:
        r = b - A @ x
        beta = np.linalg.norm(r)
        if beta / bn < tol:
            break
        V = np.zeros((restart + 1, n), dtype=complex)
        Z = np.zeros((restart, n), dtype=complex)
        H = np.zeros((restart + 1, restart), dtype=complex)
        V[0] = r / beta
        g = np.zeros(restart + 1, dtype=complex)
        g[0] = beta
        for j in range(restart):
            Z[j] = M(V[j])
            w = A @ Z[j]
            for _ in range(2):
                h = V[:j + 1].conj() @ w
                w = w - V[:j + 1].T @ h
                H[:j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = w / H[j + 1, j]
            its += 1
            y, res, _, _ = np.linalg.lstsq(H[:j + 2, :j + 1], g[:j + 2], rcond=None)
            rn = np.linalg.norm(H[:j + 2, :j + 1] @ y - g[:j + 2])
            if rn / bn < tol or its >= maxit:
                break
        x = x + Z[:j + 1].T @ y
    return x, its, np.linalg.norm(b - A @ x) / bn


if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "20k"
    theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08
    pb = annulus.build(preset)
    T = pb["terms"]
    d = pb["d"]
    Y, n, tau = 1e15, 1.0, 1e-3
    terms = [T["M"], T["K"], T["C"], T["Q"]]

    def coefs(z):
        return [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
    t0 = time.time()
    wref = 2 * np.pi * 100
    S = -(T["K"].real + wref ** 2 * T["M"].real)
    mg = MG(sp.csr_matrix(S), terms, theta=theta, max_coarse=int(os.environ.get("MAXC", 600)), smooth=os.environ.get("SMOOTH", "1") == "1")
    print("setup", time.time() - t0, mg.sizes)
    rng = np.random.default_rng(0)
    b = np.zeros(d, dtype=complex); b[3] = 1.0
    for z in [2 * np.pi * (150 + 5j), 2 * np.pi * (400 + 5j), 2 * np.pi * (700 - 5j), 2 * np.pi * (1000 - 2j), 2 * np.pi * (575 + 50j)]:
        for sm, nu in (("l1", 1), ("jac", 1), ("jac", 2)):
            mg.setup(coefs(z), smoother=sm, nu=nu, omega=float(os.environ.get("OMEGA", 0.6)))
            A = mg.A[0]
            t0 = time.time()
            x, its, res = fgmres(A, b, mg.vcycle, tol=1e-10, restart=60, maxit=400)
            print(f"z/2pi={z / 2 / np.pi:.1f} smoother={sm} nu={nu} its={its} res={res:.2e} t={time.time() - t0:.1f}s")
