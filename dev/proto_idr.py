#!/usr/bin/env python3
"""Development prototype (CPU): short-recurrence Krylov methods (BiCGStab, IDR(4)) against full GMRES on the
left-preconditioned annulus systems: number of preconditioned operator applications to reach 1e-10."""
import sys
import numpy as np
import scipy.sparse as sp
from proto_mg import MG, annulus

preset = sys.argv[1] if len(sys.argv) > 1 else "20k"
pb = annulus.build(preset, tau=2e-4)
T = pb["terms"]; d = pb["d"]
Y, n, tau = 1e15, 1.0, 2e-4
terms = [T["M"], T["K"], T["C"], T["Q"]]
coefs = lambda z: [z * z, 1.0, z * Y, n * np.exp(-1j * z * tau)]
S = -(T["K"].real + (2 * np.pi * 500) ** 2 * T["M"].real)
mg = MG(sp.csr_matrix(S), terms, theta=0.02, max_coarse=128)
rng = np.random.default_rng(0)
b = rng.standard_normal(d) + 0j


def gmres_count(op, rhs, tol=1e-10, m=60):
    bn = np.linalg.norm(rhs)
    V = [rhs / bn]; H = np.zeros((m + 1, m), dtype=complex)
    for j in range(m):
        w = op(V[j])
        for i in range(j + 1):
            H[i, j] = np.vdot(V[i], w); w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w); V.append(w / H[j + 1, j])
        e1 = np.zeros(j + 2, dtype=complex); e1[0] = bn
        y, *_ = np.linalg.lstsq(H[:j + 2, :j + 1], e1, rcond=None)
        if np.linalg.norm(H[:j + 2, :j + 1] @ y - e1) / bn < tol:
            return j + 1
    return m


def bicgstab_count(op, rhs, tol=1e-10, maxit=200):
    x = np.zeros_like(rhs); r = rhs.copy(); rt = r.copy(); bn = np.linalg.norm(rhs)
    rho = alpha = omega = 1.0; v = p = np.zeros_like(rhs); nmv = 0
    for it in range(maxit):
        rho1 = np.vdot(rt, r); beta = (rho1 / rho) * (alpha / omega); rho = rho1
        p = r + beta * (p - omega * v)
        v = op(p); nmv += 1
        alpha = rho / np.vdot(rt, v)
        s = r - alpha * v
        if np.linalg.norm(s) / bn < tol:
            return nmv
        t = op(s); nmv += 1
        omega = np.vdot(t, s) / np.vdot(t, t)
        x = x + alpha * p + omega * s
        r = s - omega * t
        if np.linalg.norm(r) / bn < tol:
            return nmv
    return nmv


def idrs_count(op, rhs, s=4, tol=1e-10, maxit=300):
    """IDR(s) (Sonneveld & van Gijzen 2008, biortho variant of van Gijzen & Sonneveld 2011)"""
    nrm = np.linalg.norm
    n_ = len(rhs); bn = nrm(rhs)
    P = np.random.default_rng(1).standard_normal((n_, s)) + 1j * np.random.default_rng(2).standard_normal((n_, s))
    P, _ = np.linalg.qr(P)
    x = np.zeros_like(rhs); r = rhs.copy(); nmv = 0
    G = np.zeros((n_, s), dtype=complex); U = np.zeros((n_, s), dtype=complex); M = np.eye(s, dtype=complex); om = 1.0
    while nmv < maxit:
        f = P.conj().T @ r
        for k in range(s):
            c = np.linalg.solve(M[k:, k:], f[k:])
            v = r - G[:, k:] @ c
            U[:, k] = U[:, k:] @ c + om * v
            G[:, k] = op(U[:, k]); nmv += 1
            for i in range(k):
                alpha = np.vdot(P[:, i], G[:, k]) / M[i, i]
                G[:, k] -= alpha * G[:, i]; U[:, k] -= alpha * U[:, i]
            M[k:, k] = P[:, k:].conj().T @ G[:, k]
            beta = f[k] / M[k, k]
            r = r - beta * G[:, k]; x = x + beta * U[:, k]
            if nrm(r) / bn < tol:
                return nmv
            if k + 1 < s:
                f[k + 1:] = f[k + 1:] - beta * M[k + 1:, k]
        t = op(r); nmv += 1
        om = np.vdot(t, r) / np.vdot(t, t)
        rho = abs(np.vdot(t, r)) / (nrm(t) * nrm(r))
        if rho < 0.7:
            om *= 0.7 / rho
        x = x + om * r; r = r - om * t
        if nrm(r) / bn < tol:
            return nmv
    return nmv


for z in [2 * np.pi * (150 + 100j), 2 * np.pi * (400 + 150j), 2 * np.pi * (700 - 150j), 2 * np.pi * (1000 - 20j), 2 * np.pi * (575 - 150j)]:
    mg.setup(coefs(z), smoother="jac", nu=1, omega=0.8)
    A = mg.A[0]
    op = lambda v: mg.vcycle(A @ v)
    rhs = mg.vcycle(b)
    print("z/2pi=%s  GMRES %d   BiCGStab %d   IDR(4) %d   IDR(8) %d" % (np.round(z / 2 / np.pi), gmres_count(op, rhs), bicgstab_count(op, rhs), idrs_count(op, rhs, 4), idrs_count(op, rhs, 8)), flush=True)
