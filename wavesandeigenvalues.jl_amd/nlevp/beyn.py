"""Beyn's contour-integral solver on the device family (reference: src/NLEVP/beyn.jl).

The quadrature loop -- ``length(Γ)·N`` times {assemble L(z), sparse LU, l solves} in the reference
(beyn.jl:62-74,112-138) -- is ONE call into libwaehip (``wae_beyn_moments``): all quadrature points and probe
columns are solved in lock-step batches by multigrid-GMRES and the moments are accumulated in HBM.  The small
dense tail (block Hankel SVD + eigen, beyn.jl:76-107) stays on the host, as in the reference (LAPACK).
"""
from __future__ import annotations

import numpy as np


def wn(z, G):
    """beyn.jl:185-209 winding number"""
    def isleft(a, b, c):
        return (b.real - a.real) * (c.imag - a.imag) - (c.real - a.real) * (b.imag - a.imag)
    z = complex(z)
    w = 0
    n = len(G)
    for i in range(n):
        a, b = complex(G[i]), complex(G[(i + 1) % n])
        if a.imag <= z.imag:
            if b.imag > z.imag and isleft(a, b, z) > 0:
                w += 1
        elif b.imag <= z.imag and isleft(a, b, z) < 0:
            w -= 1
    return w


def inpoly(z, G):
    """beyn.jl:178"""
    return wn(z, G) != 0


def gauss_points(G, N):
    """Nodes and effective weights of `gauss` (beyn.jl:112-138): N Gauss-Legendre points per polygon edge,
    weights already multiplied by (b-a)/2.  (FastGaussQuadrature.gausslegendre -> numpy leggauss.)"""
    X, W = np.polynomial.legendre.leggauss(N)
    zs, ws = [], []
    n = len(G)
    for i in range(n):
        a, b = complex(G[i]), complex(G[(i + 1) % n])
        zs.append(X * (b - a) / 2 + (a + b) / 2)
        ws.append(W * (b - a) / 2)
    return np.concatenate(zs), np.concatenate(ws)


def initialize_V(d, l):
    """beyn.jl:41-57 / :379-392 (random=false)"""
    V = np.zeros((d, l), dtype=np.complex128)
    for i in range(min(d, l)):
        V[i, i] = 1.0
    return V


def coefficient_table(L, zs):
    """row j = the T coefficients of L(z_j); leaves L.params[eigval] at the last point like repeated L(z) calls."""
    saved_active, saved_mode = L.active, L.mode
    L.active, L.mode = [L.eigval], "all"
    try:
        return np.array([L.coefficients(z) for z in zs], dtype=np.complex128).reshape(len(zs), len(L.terms))
    finally:
        L.active, L.mode = saved_active, saved_mode


def compute_moment_matrices(L, G, V=None, l=5, K=1, N=16, points=None, out_dev=0):
    """beyn.jl:233-268.  ``points=(z, w)`` overrides the contour (used to shard the quadrature over GPUs)."""
    d = L.size()
    if V is None:
        V = initialize_V(d, l)
    zs, ws = gauss_points(G, N) if points is None else points
    fam = L.ensure_solver()
    ct = coefficient_table(L, zs) if len(zs) else np.zeros((0, len(L.terms)), dtype=np.complex128)
    return fam.beyn_moments(zs, ws, ct, V, K=K, tol=L.solver_tol, maxit=L.solver_maxit, out_dev=out_dev)


def moments2eigs(A_list, tol_sigma=0.0, return_sigma=False):
    """beyn.jl:289-323 (and :76-102)"""
    if isinstance(A_list, np.ndarray):
        A_list = [A_list]
    d, dl = A_list[0].shape[:2]
    l = len(A_list) * dl
    K = A_list[0].shape[2] // 2
    B0 = np.zeros((d * K, l * K), dtype=np.complex128)
    B1 = np.zeros((d * K, l * K), dtype=np.complex128)
    for i in range(K):
        for j in range(K):
            for ll, A in enumerate(A_list):
                c0 = ll * dl + l * j
                B0[d * i:d * (i + 1), c0:c0 + dl] = A[:, :, i + j]
                B1[d * i:d * (i + 1), c0:c0 + dl] = A[:, :, i + j + 1]
    if B0.shape[0] > 8 * B0.shape[1]:
        # tall-skinny: thin QR first, SVD of the small triangular factor (same factorisation up to rounding,
        # a few times cheaper than LAPACK's SVD of the d x l matrix)
        Q, R = np.linalg.qr(B0)
        Ur, S, Wh = np.linalg.svd(R)
        U = Q @ Ur
    else:
        U, S, Wh = np.linalg.svd(B0, full_matrices=False)
    W = Wh.conj().T
    if tol_sigma > 0:
        mask = S > tol_sigma
        U, S, W = U[:, mask], S[mask], W[:, mask]
    Om, P = np.linalg.eig(U.conj().T @ B1 @ W @ np.diag(1.0 / S))
    P = U[:d, :] @ P
    return (Om, P, S) if return_sigma else (Om, P)


def pos_test(Om, P, G):
    """beyn.jl:333-337"""
    mask = np.array([inpoly(z, G) for z in Om], dtype=bool)
    return Om[mask], P[:, mask]


def beyn(L, G, l=5, K=1, N=16, tol=0.0, pos_test_=True, output=False, random=False, return_sigma=False):
    """Ω, P = beyn(L, Γ; l, K, N, tol, pos_test, output, random)   (beyn.jl:34-110)"""
    d = L.size()
    K = max(K, l // d + int(l % d != 0))
    if random:
        rng = np.random.default_rng()
        V = rng.random((d, l)) + 1j * rng.random((d, l))
    else:
        V = initialize_V(d, l)
    A = compute_moment_matrices(L, G, V, K=K, N=N)
    Om, P, S = moments2eigs(A, tol_sigma=tol, return_sigma=True)
    if output:
        print("############\nsingular values:\n", S)
    if pos_test_:
        Om, P = pos_test(Om, P, G)
    return (Om, P, S) if return_sigma else (Om, P)
