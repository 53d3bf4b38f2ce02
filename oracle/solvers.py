"""Oracle restatement of the NLEVP solver layer (test infrastructure, see oracle/__init__.py).

Follows the reference files
  src/NLEVP/beyn.jl                (beyn, gauss, wn/inpoly, compute_moment_matrices, moments2eigs, pos_test,
                                    count_poles_and_zeros)
  src/NLEVP/Householder.jl         (householder_update, householder, poly_roots)
  src/NLEVP/iterative_solvers.jl   (status flags, mslp, inveriter, lancaster, rf2s, traceiter)
  src/NLEVP/perturbation.jl        (partition iterator, perturb, perturb_disk (in memory), perturb_norm)
  src/NLEVP/LinOpFam.jl:546-618    (perturb!, perturb_fast!, perturb_norm! wrappers)
Third-party numerics the reference delegates to and what stands in for them here:
  UMFPACK sparse `\\`/`lu` (Julia SuiteSparse stdlib) -> scipy.sparse.linalg.splu (SuperLU)
  Arpack.eigs(A,M,sigma=0) (Arpack.jl 0.4.0)          -> scipy.sparse.linalg.eigs(A,M=M,sigma=0) (ARPACK mode 3)
  FastGaussQuadrature.gausslegendre 0.4.7              -> numpy.polynomial.legendre.leggauss
  LAPACK svd/eigen (OpenBLAS 0.3.9)                    -> numpy.linalg.svd / eig
"""
from __future__ import annotations

import copy
from math import factorial

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .nlevp import LinearOperatorFamily, Solution, Term, pade, poly_roots, polyval, pow1

# iterative_solvers.jl:4-14
itsol_converged = 0
itsol_maxiter = 1
itsol_slow_convergence = 2
itsol_impossible = -1
itsol_singular_exception = -2
itsol_arpack_exception = -3
itsol_isnan = -4
itsol_unknown = -5
itsol_arpack_9999 = -9999


def _solve(A, B):
    """sparse or dense `A \\ B`."""
    if sp.issparse(A):
        return spla.splu(sp.csc_matrix(A)).solve(np.asarray(B, dtype=complex))
    return np.linalg.solve(A, B)


def _lu(A):
    if sp.issparse(A):
        lu = spla.splu(sp.csc_matrix(A))
        return lu.solve, (lambda b: lu.solve(b, trans="H"))
    import scipy.linalg as sla
    f = sla.lu_factor(A)
    return (lambda b: sla.lu_solve(f, b)), (lambda b: sla.lu_solve(f, b, trans=2))


# ----------------------------------------------------------------------------------------------
# beyn.jl
# ----------------------------------------------------------------------------------------------
def wn(z, G):
    """beyn.jl:185-209 winding number of polygon G around z."""
    def isleft(a, b, c):
        return (b.real - a.real) * (c.imag - a.imag) - (c.real - a.real) * (b.imag - a.imag)
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
    return wn(complex(z), G) != 0


def contour_points(G, N):
    """beyn.jl:112-138: Gauss-Legendre nodes/weights per polygon edge.  Returns z_j and the
    effective weights w_j*(b-a)/2 so that  int = sum_j weff_j f(z_j)."""
    X, W = np.polynomial.legendre.leggauss(N)
    zs, ws = [], []
    n = len(G)
    for i in range(n):
        a, b = complex(G[i]), complex(G[(i + 1) % n])
        zs.append(X * (b - a) / 2 + (a + b) / 2)
        ws.append(W * (b - a) / 2)
    return np.concatenate(zs), np.concatenate(ws)


def initial_V(d, l):
    """beyn.jl:41-57 (random=false)"""
    V = np.zeros((d, l), dtype=complex)
    for i in range(min(d, l)):
        V[i, i] = 1.0
    return V


def compute_moment_matrices(L, G, V=None, l=5, K=1, N=16):
    """beyn.jl:62-74 / :251-268:  A[:,:,p] = sum_j weff_j z_j^p L(z_j)^{-1} V,  p=0..2K-1."""
    d = L.size()
    if V is None:
        V = initial_V(d, l)
    l = V.shape[1]
    A = np.zeros((d, l, 2 * K), dtype=complex)
    zs, ws = contour_points(G, N)
    for z, w in zip(zs, ws):
        X = _solve(L(z), V) * w
        for p in range(2 * K):
            A[:, :, p] += z ** p * X
    return A


def moments2eigs(A_list, tol_sigma=0.0, return_sigma=False):
    """beyn.jl:76-102 / :289-323: block Hankel B0,B1 -> SVD -> small eigenproblem."""
    if isinstance(A_list, np.ndarray):
        A_list = [A_list]
    d, dl = A_list[0].shape[:2]
    l = len(A_list) * dl
    K = A_list[0].shape[2] // 2
    B0 = np.zeros((d * K, l * K), dtype=complex)
    B1 = np.zeros((d * K, l * K), dtype=complex)
    for i in range(K):
        for j in range(K):
            for ll, A in enumerate(A_list):
                c0 = ll * dl + l * j
                B0[d * i:d * (i + 1), c0:c0 + dl] = A[:, :, i + j]
                B1[d * i:d * (i + 1), c0:c0 + dl] = A[:, :, i + j + 1]
    U, S, Wh = np.linalg.svd(B0, full_matrices=False)
    W = Wh.conj().T
    if tol_sigma > 0:
        mask = S > tol_sigma
        U, S, W = U[:, mask], S[mask], W[:, mask]
    Om, P = np.linalg.eig(U.conj().T @ B1 @ W @ np.diag(1.0 / S))
    P = U[:d, :] @ P
    if return_sigma:
        return Om, P, S
    return Om, P


def pos_test(Om, P, G):
    """beyn.jl:333-337"""
    mask = np.array([inpoly(z, G) for z in Om], dtype=bool)
    return Om[mask], P[:, mask]


def beyn(L, G, l=5, K=1, N=16, tol=0.0, do_pos_test=True, return_sigma=False):
    """beyn.jl:34-110"""
    d = L.size()
    K = max(K, l // d + int(l % d != 0))
    A = compute_moment_matrices(L, G, initial_V(d, l), K=K, N=N)
    Om, P, S = moments2eigs(A, tol_sigma=tol, return_sigma=True)
    if do_pos_test:
        Om, P = pos_test(Om, P, G)
    if return_sigma:
        return Om, P, S
    return Om, P


def count_poles_and_zeros(L, G, N=16):
    """beyn.jl:355-368 (trace of L^{-1} L'; small problems only)"""
    zs, ws = contour_points(G, N)
    s = 0j
    for z, w in zip(zs, ws):
        A = L(z)
        L1 = L(z, 1)
        A = A.toarray() if sp.issparse(A) else A
        L1 = L1.toarray() if sp.issparse(L1) else L1
        s += np.trace(np.linalg.solve(A, L1)) * w
    return s / 2 / np.pi / 1j


# ----------------------------------------------------------------------------------------------
# perturbation.jl
# ----------------------------------------------------------------------------------------------
def partitions(n):
    """perturbation.jl:2-80 (Kelleher's accel_asc): ascending compositions of n, same order."""
    a = [0] * (n + 1)
    k = 1
    y = n - 1
    while k != 0:
        x = a[k - 1] + 1
        k -= 1
        while 2 * x <= y:
            a[k] = x
            y -= x
            k += 1
        l = k + 1
        while x <= y:
            a[k] = x
            a[l] = y
            yield a[:k + 2]
            x += 1
            y -= 1
        a[k] = x + y
        y = x + y - 1
        yield a[:k + 1]


def part2mult(p):
    """perturbation.jl:95-104"""
    z = sum(p)
    mu = [0] * z
    if list(p) != [0]:
        for i in p:
            mu[i - 1] += 1
    return mu


def multinomcoeff(mu):
    """perturbation.jl:111-113"""
    r = float(factorial(sum(mu)))
    for m in mu:
        r /= float(factorial(m))
    return r


def weigh(mu):
    """perturbation.jl:115-121"""
    return sum((g + 1) * m for g, m in enumerate(mu))


def multi_indices_at_order(k):
    """perturbation.jl:186-244 (in-memory variant): dict (m,n) -> list of multiplicity vectors,
    in the order the reference writes them to its `k/m_n` files."""
    Mu = {}
    for n in range(1, k + 1):
        Mu.setdefault((0, n), []).append([])
    for m in range(1, k + 1):
        for p in partitions(m):
            if p == [k]:
                continue
            mu = part2mult(p)
            for n in range(0, k - m + 1):
                Mu.setdefault((sum(mu), n), []).append(mu)
    return Mu


def _factor_singular(L00):
    """perturbation.jl:327-332: lu(L(0,0)) of the (nearly) singular operator."""
    return _lu(L00)[0]


def perturb(L, N, v0, v0Adj):
    """perturbation.jl:319-367 (partitions generated on the fly)."""
    v0 = np.array(v0, dtype=complex)
    v0Adj = np.array(v0Adj, dtype=complex)
    v0 = v0 / np.sqrt(np.vdot(v0, v0))
    L10 = L(1, 0)
    v0Adj = v0Adj / np.vdot(v0Adj, L10 @ v0)             # v0Adj /= v0Adj'*L(1,0)*v0  (divides by the scalar)
    lam = np.zeros(N + 1, dtype=complex)
    v = [None] * (N + 1)
    v[0] = v0
    solve = _factor_singular(L(0, 0))
    for k in range(1, N + 1):
        r = np.zeros_like(v0)
        for n in range(1, k + 1):
            r += L(0, n) @ v[k - n]
        for m in range(1, k + 1):
            for p in partitions(m):
                if p == [k]:
                    continue
                mu = part2mult(p)
                coeff = 1.0 + 0j
                for g, mu_g in enumerate(mu):
                    coeff *= lam[g + 1] ** mu_g
                for n in range(0, k - m + 1):
                    r += (L(sum(mu), n) @ v[k - n - m]) * multinomcoeff(mu) * coeff
        lam[k] = -np.vdot(v0Adj, r) / np.vdot(v0Adj, L10 @ v0)
        v[k] = solve(-(r + lam[k] * (L10 @ v0)))
        v[k] = v[k] - np.vdot(v0, v[k]) * v0
    return lam, v


def perturb_disk(L, N, v0, v0Adj, Y=None):
    """perturbation.jl:374-444 (multi-indices held in memory instead of read from `k/m_n` files);
    with Y given: perturb_norm, perturbation.jl:487-560."""
    v0 = np.array(v0, dtype=complex)
    v0Adj = np.array(v0Adj, dtype=complex)
    L10 = L(1, 0)
    if Y is None:
        ip = lambda a, b: np.vdot(a, b)
        v0 = v0 / np.sqrt(ip(v0, v0))
        v0Adj = v0Adj / np.vdot(v0Adj, L10 @ v0)
        wl = v0Adj
    else:
        ip = lambda a, b: np.vdot(a, Y @ b)
        v0 = v0 / np.sqrt(ip(v0, v0))
        v0Adj = _solve(Y, v0Adj)
        v0Adj = v0Adj / np.vdot(v0Adj, Y @ (L10 @ v0))
        wl = Y.conj().T @ v0Adj                       # v0Adj'*Y*x == (Y'v0Adj)'x
    lam = np.zeros(N + 1, dtype=complex)
    v = [None] * (N + 1)
    v[0] = v0
    solve = _factor_singular(L(0, 0))
    for k in range(1, N + 1):
        Mu = multi_indices_at_order(k)
        r = np.zeros_like(v0)
        for m in range(0, k + 1):
            for n in range(0, k - m + 1):
                if (m == 0 and n == 0) or (k == 1 and m == 1):
                    continue
                w = np.zeros_like(v0)
                for mu in Mu.get((m, n), []):
                    coeff = 1.0 + 0j
                    for g, mu_g in enumerate(mu):
                        coeff *= lam[g + 1] ** mu_g
                    w += v[k - n - weigh(mu)] * multinomcoeff(mu) * coeff
                r += L(m, n) @ w
        lam[k] = -np.vdot(wl, r) / np.vdot(wl, L10 @ v0)
        v[k] = solve(-(r + lam[k] * (L10 @ v0)))
        v[k] = v[k] - ip(v0, v[k]) * v0
        c = 0j
        for l in range(1, k):
            c -= 0.5 * ip(v[l], v[k - l])
        v[k] = v[k] + c * v[0]
    return lam, v


def _perturb_wrapper(kernel, sol, L, param, N, mode="compact"):
    """LinOpFam.jl:546-618"""
    active, params, cur_mode = L.active, L.params, L.mode
    L.params = sol.params
    L.active = [sol.eigval, param]
    L.mode = mode
    key = f"{param}/Taylor"
    try:
        lam, v = kernel(L, N, sol.v, sol.v_adj)
    finally:
        L.active, L.mode, L.params = active, cur_mode, params
    lam[0] = sol.params[sol.eigval]
    sol.eigval_pert[key], sol.v_pert[key] = lam, v


def perturb_(sol, L, param, N, mode="compact"):
    """perturb!  LinOpFam.jl:546-560"""
    _perturb_wrapper(perturb, sol, L, param, N, mode)


def perturb_fast_(sol, L, param, N, mode="compact"):
    """perturb_fast!  LinOpFam.jl:575-589"""
    _perturb_wrapper(perturb_disk, sol, L, param, N, mode)


def perturb_norm_(sol, L, param, N, mode="compact"):
    """perturb_norm!  LinOpFam.jl:604-618"""
    Y = -L.terms[-1].coeff
    _perturb_wrapper(lambda L_, N_, v, va: perturb_disk(L_, N_, v, va, Y=Y), sol, L, param, N, mode)


# ----------------------------------------------------------------------------------------------
# Householder.jl / iterative_solvers.jl
# ----------------------------------------------------------------------------------------------
def householder_update(f):
    """Householder.jl:21-35"""
    order = len(f) - 1
    if order == 1:
        return -f[0] / f[1]
    if order == 2:
        return -f[0] * f[1] / (f[1] ** 2 - 0.5 * f[0] * f[2])
    if order == 3:
        return -(6 * f[0] * f[1] ** 2 - 3 * f[0] ** 2 * f[2]) / (6 * f[1] ** 3 - 6 * f[0] * f[1] * f[2] + f[0] ** 2 * f[3])
    if order == 4:
        return -(4 * f[0] * (6 * f[1] ** 3 - 6 * f[0] * f[1] * f[2] + f[0] ** 2 * f[3])) / (
            24 * f[1] ** 4 - 36 * f[0] * f[1] ** 2 * f[2] + 6 * f[0] ** 2 * f[2] ** 2 + 8 * f[0] ** 2 * f[1] * f[3] - f[0] ** 3 * f[4])
    return (5 * f[0] * (24 * f[1] ** 4 - 36 * f[0] * f[1] ** 2 * f[2] + 6 * f[0] ** 2 * f[2] ** 2 + 8 * f[0] ** 2 * f[1] * f[3] - f[0] ** 3 * f[4])) / (
        -120 * f[1] ** 5 + 240 * f[0] * f[1] ** 3 * f[2] - 60 * f[0] ** 2 * f[1] ** 2 * f[3]
        + 10 * f[0] ** 2 * f[1] * (-9 * f[2] ** 2 + f[0] * f[4]) + f[0] ** 3 * (20 * f[2] * f[3] - f[0] * f[5]))


def _eigs_si(A, M, nev, v0):
    """Arpack.eigs(A,M,nev=nev,sigma=0,v0=v0) (Householder.jl:100): eigenvalues of A x = lam M x nearest 0."""
    d = A.shape[0]
    if d < 3:
        Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
        Md = M.toarray() if sp.issparse(M) else np.asarray(M)
        import scipy.linalg as sla
        lam, V = sla.eig(Ad, Md)
        idx = np.argsort(np.abs(lam))[:nev]
        return lam[idx], V[:, idx]
    lam, V = spla.eigs(sp.csc_matrix(A), k=nev, M=sp.csc_matrix(M), sigma=0, v0=np.asarray(v0, dtype=complex),
                       tol=0, maxiter=1000)
    return lam, V


def _newton_on_aux(L, z, order, nev, v0, v0_adj, update):
    """Shared body of householder / mslp (Householder.jl:96-127, iterative_solvers.jl:129-188):
    solve the auxiliary linear EVPs, expand lam(omega) by perturbation theory, return candidates."""
    L.params[L.eigval] = z
    L.params[L.auxval] = 0
    A = L(z)
    M = -L.terms[-1].coeff
    lam, v = _eigs_si(A, M, nev, v0)
    lam_adj, v_adj = _eigs_si(A.conj().T, M.conj().T, nev, v0_adj)
    idx = np.argsort(np.abs(lam)); lam, v = lam[idx], v[:, idx]
    idx = np.argsort(np.abs(lam_adj)); lam_adj, v_adj = lam_adj[idx], v_adj[:, idx]
    cand = []
    L.active = [L.auxval, L.eigval]
    for i in range(nev):
        L.params[L.auxval] = lam[i]
        sol = Solution(L.params, v[:, i], v_adj[:, i], L.auxval)
        perturb_(sol, L, L.eigval, order, mode="householder")
        cand.append(update(sol.eigval_pert[f"{L.eigval}/Taylor"]))
    L.active = [L.eigval]
    return lam, v, v_adj, cand


def householder(L, z, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, nev=1, v0=None, v0_adj=None):
    """Householder.jl:70-192.  Flags: 1 converged, 0 slow, -1 maxiter (Householder.jl:139-172)."""
    z = complex(z)
    z0 = complex(np.inf)
    lam = np.inf
    n = 0
    active, mode = L.active, L.mode
    d = L.size()
    v0 = np.ones(d, dtype=complex) if v0 is None else np.asarray(v0, dtype=complex)
    v0_adj = np.conj(v0) if v0_adj is None else np.asarray(v0_adj, dtype=complex)
    flag = 1
    M = -L.terms[-1].coeff
    history = []
    try:
        while abs(z - z0) > tol and n < maxiter:
            history.append(z)
            z0 = z
            lams, v, v_adj, dzs = _newton_on_aux(
                L, z, order, nev, v0, v0_adj,
                lambda c: householder_update([factorial(i) * ci for i, ci in enumerate(c)]))
            i = int(np.argsort(np.abs(dzs))[0])
            lam = lams[i]
            L.params[L.auxval] = lam
            z = z + relax * dzs[i]
            v0 = (1 - relax) * v0 + relax * v[:, i]
            v0_adj = (1 - relax) * v0_adj + relax * v_adj[:, i]
            n += 1
    except (spla.ArpackError, spla.ArpackNoConvergence):
        flag = -4
    except (RuntimeError, np.linalg.LinAlgError):
        flag = -6
        L.params[L.eigval] = z
    if flag == 1:
        L.params[L.eigval] = z
        history.append(z)
        if n >= maxiter:
            flag = -1
        elif abs(lam) <= lam_tol:
            flag = 1
        elif abs(z - z0) <= tol:
            flag = 0
        elif np.isnan(z):
            flag = -5
        else:
            flag = -3
    L.active, L.mode = active, mode
    v0 = v0 / np.sqrt(np.vdot(v0, M @ v0))
    v0_adj = v0_adj / np.conj(np.vdot(v0_adj, L(L.params[L.eigval], 1) @ v0))
    sol = Solution(L.params, v0, v0_adj, L.eigval)
    sol.history = history
    return sol, n, flag


def mslp(L, z, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, nev=1, v0=None, v0_adj=None,
         num_order=1, scale=1.0):
    """iterative_solvers.jl:93-252"""
    z = complex(z) * scale
    tol = tol * scale
    z0 = complex(np.inf)
    lam = np.inf
    lam0 = np.inf
    n = 0
    active, mode = L.active, L.mode
    d = L.size()
    v0 = np.ones(d, dtype=complex) if v0 is None else np.asarray(v0, dtype=complex)
    v0_adj = np.conj(v0) if v0_adj is None else np.asarray(v0_adj, dtype=complex)
    flag = itsol_converged
    if L.terms[-1].operator != "__aux__":
        I = sp.identity(d, dtype=complex, format="csc") if sp.issparse(L.terms[0].coeff) else np.eye(d, dtype=complex)
        L.push(Term(-I, (pow1,), (("__aux__",),), "__aux__", "__aux__"))
        L.auxval = "__aux__"
    M = -L.terms[-1].coeff
    history = []
    try:
        while abs(z - z0) > tol and n < maxiter:
            history.append(z)
            pades = []

            def upd(coeffs):
                num, den = pade(coeffs, num_order, order - num_order)
                pades.append((num, den))
                roots = poly_roots(num)
                return roots[np.argsort(np.abs(roots))[0]]
            lams, v, v_adj, dzs = _newton_on_aux(L, z, order, nev, v0, v0_adj, upd)
            if not np.isinf(z0):
                back = [lam0 - polyval(num, z0 - z) / polyval(den, z0 - z) for num, den in pades]
                i = int(np.argsort(np.abs(back))[0])
            else:
                i = int(np.argsort(np.abs(dzs))[0])
            lam = lams[i]
            L.params[L.auxval] = lam
            z0 = z
            lam0 = lam
            z = z + relax * dzs[i]
            v0 = (1 - relax) * v0 + relax * v[:, i]
            v0_adj = (1 - relax) * v0_adj + relax * v_adj[:, i]
            n += 1
    except (spla.ArpackError, spla.ArpackNoConvergence):
        flag = itsol_arpack_exception
    except (RuntimeError, np.linalg.LinAlgError):
        flag = itsol_singular_exception
        L.params[L.eigval] = z
    if flag == itsol_converged:
        L.params[L.eigval] = z
        history.append(z)
        if n >= maxiter:
            flag = itsol_maxiter
        elif abs(lam) <= lam_tol:
            flag = itsol_converged
        elif abs(z - z0) <= tol:
            flag = itsol_slow_convergence
        elif np.isnan(z):
            flag = itsol_isnan
        else:
            flag = itsol_impossible
    L.active, L.mode = active, mode
    v0 = v0 / np.sqrt(np.vdot(v0, M @ v0))
    v0_adj = v0_adj / np.conj(np.vdot(v0_adj, L(L.params[L.eigval], 1) @ v0))
    sol = Solution(L.params, v0, v0_adj, L.eigval)
    sol.history = history
    return sol, n, flag


def _finish(n, maxiter, z, z0, tol, flag):
    """iterative_solvers.jl:326-342 convergence checks shared by the Newton variants."""
    if flag != itsol_converged:
        return flag
    if n >= maxiter:
        return itsol_maxiter
    if abs(z - z0) <= tol:
        return itsol_converged
    if np.isnan(z):
        return itsol_isnan
    return itsol_impossible


def inveriter(L, z, maxiter=10, tol=0.0, x0=None, v=None):
    """iterative_solvers.jl:285-347"""
    d = L.size()
    x0 = np.ones(d, dtype=complex) if x0 is None else np.asarray(x0, dtype=complex)
    v = np.ones(d, dtype=complex) if v is None else np.asarray(v, dtype=complex)
    x0 = x0 / np.vdot(v, x0)
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            u = _solve(L(z, 0), L(z, 1) @ x0)
            z = z0 - np.vdot(v, x0) / np.vdot(v, u)
            x0 = u / np.vdot(v, u)
            n += 1
    except (RuntimeError, np.linalg.LinAlgError):
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, x0, [], L.eigval, L.auxval), n, flag


def lancaster(L, z, maxiter=10, tol=0.0, x0=None, y0=None):
    """iterative_solvers.jl:378-434"""
    d = L.size()
    x0 = np.ones(d, dtype=complex) if x0 is None else np.asarray(x0, dtype=complex)
    y0 = np.ones(d, dtype=complex) if y0 is None else np.asarray(y0, dtype=complex)
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            A = L(z)
            s, sH = _lu(A)
            xi = s(x0)
            eta = sH(y0)
            z = z0 - np.vdot(eta, L(z, 0) @ xi) / np.vdot(eta, L(z, 1) @ xi)
            n += 1
    except (RuntimeError, np.linalg.LinAlgError):
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, np.zeros(d, dtype=complex), [], L.eigval), n, flag


def rf2s(L, z, maxiter=10, tol=0.0, x0=None, y0=None):
    """iterative_solvers.jl:548-614"""
    d = L.size()
    if x0 is None:
        x0 = np.zeros(d, dtype=complex); x0[0] = 1
    if y0 is None:
        y0 = np.zeros(d, dtype=complex); y0[0] = 1
    x0 = np.asarray(x0, dtype=complex); y0 = np.asarray(y0, dtype=complex)
    x0 = x0 / np.sqrt(np.vdot(x0, x0)); y0 = y0 / np.sqrt(np.vdot(y0, y0))
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            s, sH = _lu(L(z))
            L1 = L(z, 1)
            x0 = s(L1 @ x0)
            y0 = sH(L1.conj().T @ y0)
            x0 = x0 / np.sqrt(np.vdot(x0, x0)); y0 = y0 / np.sqrt(np.vdot(y0, y0))
            idx = 0
            z00 = complex(np.inf)
            while abs(z - z00) > tol and idx < 10:
                z00 = z
                z = z - np.vdot(y0, L(z) @ x0) / np.vdot(y0, L(z, 1) @ x0)
                idx += 1
            n += 1
    except (RuntimeError, np.linalg.LinAlgError):
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, x0, y0, L.eigval), n, flag


def traceiter(L, z, maxiter=10, tol=0.0, relax=1.0):
    """iterative_solvers.jl:463-517 (d solves per step)"""
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            A = L(z); L1 = L(z, 1)
            A = A.toarray() if sp.issparse(A) else A
            L1 = L1.toarray() if sp.issparse(L1) else L1
            dz = -1.0 / np.trace(np.linalg.solve(A, L1))
            z = z0 + relax * dz
            n += 1
    except (RuntimeError, np.linalg.LinAlgError):
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, [], [], L.eigval), n, flag


# ----------------------------------------------------------------------------------------------
# reduced-basis helpers  (beyn.jl:429-595)
# ----------------------------------------------------------------------------------------------
def _orth_append(Q, x):
    """next column of the incremental QR (beyn.jl:604-626 builds Householder reflectors; the span of the first k
    columns is the same for Gram-Schmidt, the column phases differ -- nothing downstream depends on them)."""
    x = np.array(x, dtype=complex).ravel()
    for _ in range(2):
        if Q.shape[1]:
            x = x - Q @ (Q.conj().T @ x)
    return np.hstack([Q, (x / np.linalg.norm(x))[:, None]])


def generate_subspace(L, Y, tol, Z, include_Y=True):
    """beyn.jl:429-560: greedy orthonormal basis Q such that the Galerkin solution of L(z) x = y in span(Q) has
    residual <= tol for every sample point z in Z and every column y of Y.  Returns Q, resnorm."""
    Y = np.asarray(Y, dtype=complex)
    d, k = Y.shape
    Z = list(Z)
    Q = np.zeros((d, 0), dtype=complex)
    A0 = L(Z[0])
    for kk in range(k):
        Q = _orth_append(Q, Y[:, kk] if include_Y else _solve(A0, Y[:, kk]))
    resnorm = np.zeros(len(Z) * k)
    for idx, z in enumerate(Z):
        if Q.shape[1] == d:
            break
        Lz = L(z)
        QLQ = Q.conj().T @ (Lz @ Q)
        QY = Q.conj().T @ Y
        for kk in range(k):
            X = Q @ np.linalg.solve(QLQ, QY[:, kk])
            res = np.linalg.norm(Lz @ X - Y[:, kk])
            if res > tol:
                Q = _orth_append(Q, _solve(Lz, Y[:, kk]))
                QLQ = Q.conj().T @ (Lz @ Q)
                QY = Q.conj().T @ Y
                X = Q @ np.linalg.solve(QLQ, QY[:, kk])
                res = np.linalg.norm(Lz @ X - Y[:, kk])
            resnorm[kk + idx * k] = res
    return Q, resnorm


def generate_subspace_contour(L, Y, tol, G, N, include_Y=True):
    """beyn.jl:562-577: sample points = N Gauss-Legendre nodes on every edge of the polygon G"""
    xg, _ = np.polynomial.legendre.leggauss(N)
    G = list(G)
    Z = np.concatenate([xg * (G[(i + 1) % len(G)] - G[i]) / 2 + (G[i] + G[(i + 1) % len(G)]) / 2 for i in range(len(G))])
    return generate_subspace(L, Y, tol, Z, include_Y=include_Y)


def project(L, Q):
    """beyn.jl:579-595: the family P(z) = Q' L(z) Q, term by term"""
    P = LinearOperatorFamily([L.eigval], [L.params[L.eigval]])
    P.params = copy.deepcopy(L.params)
    P.eigval, P.auxval, P.mode, P.active = L.eigval, L.auxval, L.mode, list(L.active)
    for t in L.terms:
        M = Q.conj().T @ (t.coeff @ Q)
        P.push(Term(np.asarray(M), t.func, t.params, t.symbol, t.operator))
    return P
