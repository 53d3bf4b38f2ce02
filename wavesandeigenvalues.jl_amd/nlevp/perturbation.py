"""Adjoint high-order perturbation theory on the device family (reference: src/NLEVP/perturbation.jl and the
wrappers perturb!/perturb_fast!/perturb_norm! at src/NLEVP/LinOpFam.jl:546-618).

Same recurrence, regrouped.  With L_{m,n} = Σ_t f_t^{(m,n)} A_t the reference's
    r_k = Σ_{(m,n)} L_{m,n} · w_{m,n},   w_{m,n} = Σ_μ multinom(μ) Π_g λ_g^{μ_g} v_{k-n-wt(μ)}      (perturbation.jl:394-415)
becomes   r_k = Σ_t A_t · (V_k g_t),  g_t = Σ_{(m,n)} f_t^{(m,n)} c_{m,n}:   T tall-skinny products and ONE fused
multi-term SpMV with a per-term input column (``wae_spmv_sum_multi``) per order, instead of O(p(k)·k) axpys and
O(k²) assemble+SpMV; the multi-indices are generated in memory (no ``compressed_perturbation_data`` files,
deps/build.jl).  The solve with the singular L(0,0) (perturbation.jl:385-388,423) is a multigrid-GMRES solve
of the consistent system followed by the reference's re-orthogonalisation against v0.
"""
from __future__ import annotations

from math import factorial

import numpy as np


def partitions(n):
    """perturbation.jl:2-80 (Kelleher's accelerated ascending compositions), same output order."""
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
    mu = [0] * sum(p)
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
    """perturbation.jl:186-244, in memory: (m,n) -> list of multiplicity vectors."""
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


def _recurrence(L, N, v0, v0Adj, normalize, Y=None, skip_last_solve=False):
    """perturbation.jl:319-367 (normalize=False), :374-444 (normalize=True), :487-560 (Y given): ONE device call
    (``wae_perturb``).  The host only evaluates the (N+1)^2 x T scalar coefficient table f_t^{(m,n)} = L.coefficients(m,n);
    partitions, the tall-skinny products V_k g_t, the fused multi-input SpMV, the N solves on the fixed hierarchy and
    all normalisations run inside the library on HBM-resident vectors."""
    fam = L.ensure_solver()
    T = len(L.terms)
    table = np.zeros((N + 1, N + 1, T), dtype=np.complex128)
    for m in range(N + 1):
        for n in range(N + 1 - m):
            table[m, n] = L.coefficients(m, n)
    mode = (2 if Y is not None else (1 if normalize else 0)) + (16 if skip_last_solve else 0)
    lam, V = fam.perturb(table, N, v0, v0Adj, norm_mode=mode, coeffsY=None if Y is None else Y.coeffs,
                         tol=L.solver_tol, maxit=L.solver_maxit, quiet=skip_last_solve)   # (inside a Newton step: the caller judges)
    return lam, [V[:, i].copy() for i in range(N + 1)]


def _recurrence_host(L, N, v0, v0Adj, normalize, Y=None, skip_last_solve=False):
    """The same recurrence orchestrated from the host over wae_spmv_sum_multi + wae_solve (kept as a cross-check of
    the device implementation in the GPU tests; not used by the solvers)."""
    fam = L.ensure_solver()
    T = len(L.terms)
    v0 = np.array(v0, dtype=np.complex128)
    v0Adj = np.array(v0Adj, dtype=np.complex128)
    L10 = L(1, 0)
    if Y is None:
        ip = np.vdot
        v0 = v0 / np.sqrt(ip(v0, v0))
        u10 = L10 @ v0
        v0Adj = v0Adj / np.vdot(v0Adj, u10)
        wl = v0Adj
    else:
        ip = lambda a, b: np.vdot(a, Y @ b)
        v0 = v0 / np.sqrt(ip(v0, v0))
        u10 = L10 @ v0
        v0Adj = Y.solve(v0Adj)
        v0Adj = v0Adj / np.vdot(v0Adj, Y @ u10)
        wl = Y.H @ v0Adj
    denom = np.vdot(wl, u10)
    lam = np.zeros(N + 1, dtype=np.complex128)
    V = np.zeros((len(v0), N + 1), dtype=np.complex128, order="F")
    V[:, 0] = v0
    L00 = L(0, 0)
    # coefficient table f_t^{(m,n)}
    F = {}
    for m in range(N + 1):
        for n in range(N + 1 - m):
            F[(m, n)] = L.coefficients(m, n)
    ones = np.ones(T, dtype=np.complex128)
    for k in range(1, N + 1):
        Mu = multi_indices_at_order(k)
        Gk = np.zeros((k, T), dtype=np.complex128)       # g_t[i]: weight of v_i in the input column of term t
        for m in range(0, k + 1):
            for n in range(0, k - m + 1):
                if (m == 0 and n == 0) or (k == 1 and m == 1):
                    continue
                f = F[(m, n)]
                if not np.any(f):
                    continue
                c = np.zeros(k, dtype=np.complex128)
                for mu in Mu.get((m, n), []):
                    coeff = multinomcoeff(mu)
                    for g, mu_g in enumerate(mu):
                        if mu_g:
                            coeff = coeff * lam[g + 1] ** mu_g
                    c[k - n - weigh(mu)] += coeff
                Gk += np.outer(c, f)
        U = V[:, :k] @ Gk                                  # d x T input columns
        r = fam.spmv_multi(ones, U)
        lam[k] = -np.vdot(wl, r) / denom
        if skip_last_solve and k == N:
            break            # only the eigenvalue coefficients are wanted (Newton solvers): v_N feeds nothing
        vk = L00.solve(-(r + lam[k] * u10))
        vk = vk - ip(v0, vk) * v0
        if normalize:
            c = 0j
            for l in range(1, k):
                c -= 0.5 * ip(V[:, l], V[:, k - l])
            vk = vk + c * v0
        V[:, k] = vk
    return lam, [V[:, i].copy() for i in range(N + 1)]


def perturb(L, N, v0, v0Adj):
    """perturbation.jl:319-367"""
    return _recurrence(L, N, v0, v0Adj, normalize=False)


def perturb_disk(L, N, v0, v0Adj):
    """perturbation.jl:374-444"""
    return _recurrence(L, N, v0, v0Adj, normalize=True)


def perturb_norm(L, N, v0, v0Adj):
    """perturbation.jl:487-560 (Y = -L.terms[end].coeff)"""
    Y = L.term_operator(len(L.terms) - 1, -1.0)
    return _recurrence(L, N, v0, v0Adj, normalize=True, Y=Y)


def _wrapper(kernel, sol, L, param, N, mode):
    """LinOpFam.jl:546-560 (and :575-589, :604-618)"""
    active, params, cur_mode = L.active, L.params, L.mode
    L.params = sol.params
    L.active = [sol.eigval, param]
    L.mode = mode
    key = f"{param}/Taylor"
    try:
        if mode == "householder":      # called from householder/mslp: eigenvalue series only (Householder.jl:115-116)
            lam, v = _recurrence(L, N, sol.v, sol.v_adj, normalize=False, skip_last_solve=True)
        else:
            lam, v = kernel(L, N, sol.v, sol.v_adj)
    finally:
        L.active, L.mode, L.params = active, cur_mode, params
    lam[0] = sol.params[sol.eigval]
    sol.eigval_pert[key], sol.v_pert[key] = lam, v


def eigval_series_slots(L, eigval, param, N, v_slot, v_col, w_slot, w_col):
    """The eigenvalue series of `perturb!(sol, L, param, N; mode = :householder)` (LinOpFam.jl:546-560 with perturbation.jl:319-367)
    for the eigenpair held in slot columns of the family (wae_perturb_slots): L.params carries the expansion point, `eigval` names the
    eigenvalue parameter of the pair (the auxiliary eigenvalue of Householder.jl:115-116), `param` the perturbed one.  Returns the
    Taylor coefficients lam[0..N] (lam[0] = L.params[eigval]); no vector leaves the device."""
    fam = L.ensure_solver()
    T = len(L.terms)
    active, cur_mode = L.active, L.mode
    L.active = [eigval, param]
    L.mode = "householder"
    try:
        table = np.zeros((N + 1, N + 1, T), dtype=np.complex128)
        for m in range(N + 1):
            for n in range(N + 1 - m):
                table[m, n] = L.coefficients(m, n)
    finally:
        L.active, L.mode = active, cur_mode
    lam, _ = fam.perturb_slots(table, N, v_slot, v_col, w_slot, w_col, norm_mode=16, tol=L.solver_tol, maxit=L.solver_maxit, quiet=True)
    lam[0] = L.params[eigval]
    return lam


def perturb_(sol, L, param, N, mode="compact"):
    """perturb!(sol,L,param,N;mode)"""
    _wrapper(perturb, sol, L, param, N, mode)


def perturb_fast_(sol, L, param, N, mode="compact"):
    """perturb_fast!(sol,L,param,N;mode) -- needs no multi-index files here"""
    _wrapper(perturb_disk, sol, L, param, N, mode)


def perturb_norm_(sol, L, param, N, mode="compact"):
    """perturb_norm!(sol,L,param,N;mode)"""
    _wrapper(perturb_norm, sol, L, param, N, mode)
