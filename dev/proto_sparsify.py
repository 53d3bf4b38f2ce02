#!/usr/bin/env python3
"""Prototype (CPU): non-Galerkin level-1 operators -- drop small entries of the Galerkin planes (lumped onto the diagonal)
and count preconditioned GMRES iterations.  Decides whether a sparsified coarse operator is worth building on the device."""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import proto_mg as PM


def sparsify(planes, theta, lump=True):
    """keep (i,j) if any of the first two planes (M, K) has |a_ij| >= theta*sqrt(|a_ii a_jj|); dropped values go to the diagonal"""
    n = planes[0].shape[0]
    keep = sp.identity(n, format="csr", dtype=bool)
    for A in planes[:2]:
        C = A.tocoo()
        D = np.abs(A.diagonal())
        k = np.abs(C.data) >= theta * np.sqrt(D[C.row] * D[C.col])
        keep = keep + sp.csr_matrix((np.ones(k.sum(), dtype=bool), (C.row[k], C.col[k])), shape=A.shape)
    keep = keep.astype(bool).astype(float)
    out = []
    for A in planes:
        F = A.multiply(keep).tocsr()
        if lump:
            F = F + sp.diags(np.asarray(A.sum(axis=1)).ravel() - np.asarray(F.sum(axis=1)).ravel())
        F.eliminate_zeros()
        out.append(sp.csr_matrix(F))
    return out


if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "20k"
    pb = PM.annulus.build(preset, tau=2e-4)
    T = pb["terms"]
    d = pb["d"]
    Y, n, tau = 1e15, 1.0, 2e-4
    terms = [T["M"], T["K"], T["C"], T["Q"]]
    coefs = lambda z: [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
    S = -(T["K"].real + (2 * np.pi * 100) ** 2 * T["M"].real)
    mg = PM.MG(sp.csr_matrix(S), terms, theta=0.08, max_coarse=600)
    gal = [list(ts) for ts in mg.terms]
    rng = np.random.default_rng(0)
    b = rng.standard_normal(d) + 1j * rng.standard_normal(d)
    zs = [2 * np.pi * (150 + 150j), 2 * np.pi * (400 - 150j), 2 * np.pi * (700 + 150j), 2 * np.pi * (1000 - 60j), 2 * np.pi * (575 + 150j), 2 * np.pi * (150 + 20j)]
    for theta in (0.0, 0.02, 0.05, 0.1, 0.2):
        for lump in ((True,) if theta == 0 else (True, False)):
            mg.terms = [list(ts) for ts in gal]
            if theta > 0:
                for lvl in range(1, len(mg.terms) - 1):
                    mg.terms[lvl] = sparsify(gal[lvl], theta, lump)
            nnz = [ts[0].nnz / ts[0].shape[0] for ts in mg.terms]
            its = []
            for z in zs:
                mg.setup(coefs(z), smoother="jac", nu=1, omega=0.8)
                x, it, res = PM.fgmres(mg.A[0], b, mg.vcycle, tol=1e-10, restart=40, maxit=200)
                its.append(it)
            print(f"theta={theta} lump={lump} nnz/row per level {[round(v, 1) for v in nnz]} its {its} sum {sum(its)}", flush=True)
