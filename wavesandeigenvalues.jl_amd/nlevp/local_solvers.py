"""Local (Newton-type) eigenvalue solvers on the device family.

Reference: src/NLEVP/Householder.jl (householder, householder_update), src/NLEVP/iterative_solvers.jl
(status flags, mslp, inveriter, lancaster, rf2s, traceiter).  Same signatures, same flag conventions; every
``L(z)\\b`` / ``lu`` / ``Arpack.eigs`` of the reference is a call into libwaehip (multigrid-GMRES solves and
shift-invert Arnoldi on the GPU); only scalar updates and the tiny Hessenberg eigenproblem run on the host.
"""
from __future__ import annotations

from math import factorial

import numpy as np
import scipy.sparse as sp

from .._lib import OP_C, OP_N, WaeError
from .algebra import pow1
from .linopfam import Operator, Solution, Term, pade, poly_roots, polyval
from .perturbation import eigval_series_slots, perturb_

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


def decode_error_flag(flag):
    """iterative_solvers.jl:22-44 (the reference's version does not parse; this is its evident intent)"""
    return {
        itsol_converged: "Solution converged, everythink OK!",
        itsol_maxiter: "Warning: Maximum number of iterations has been reached!",
        itsol_slow_convergence: "Warning: Slow progress!",
        itsol_impossible: "Error: This error should be impossible. Please, contact the package developers!",
        itsol_singular_exception: "Error: Singular Exception!",
        itsol_arpack_exception: "Error: Arpack exception!",
        itsol_arpack_9999: "Error: Arpack -9999 error!",
        itsol_unknown: "Error: Unknown error ocurred!",
    }.get(flag, "Unknown flag code.")


class EigsError(RuntimeError):
    """stands where Arpack.ARPACKException stands in the reference (Householder.jl:140)"""


def eigs(A, M, nev=1, v0=None, ncv=None, tol=1e-12, maxiter=300, sigma=0.0, return_gap=False):
    """Arpack.eigs(A, M, nev=nev, sigma=0, v0=v0)  (Householder.jl:100-101, iterative_solvers.jl:132-133).

    A, M are Operator views of one family (pass ``A.H, M.H`` for the adjoint problem).  The Arnoldi factorisation
    of A^{-1}M runs on the device (``wae_arnoldi_shiftinvert``, one multigrid-GMRES solve per step); the Ritz
    values of the small Hessenberg matrix and the restarts are done here.  Short factorisations (6 steps) are
    restarted with the wanted Ritz vectors until the Ritz residual is below ``tol``: close to an eigenvalue of
    the NLEVP the wanted pair is separated by many orders of magnitude and one or two short runs suffice
    (ARPACK would spend ncv = 20 solves regardless).  Returns (lam[nev], V[d,nev]).

    ``sigma``: the factorisation is built for (A - sigma*M)^{-1} M, lam = sigma + 1/theta.  The reference always
    uses sigma = 0 and relies on UMFPACK factorising the (numerically singular) A once the Newton iteration
    has converged (Householder.jl:145 catches the SingularException); an iterative inner solver needs a
    regular operator, so the Newton-type callers pass a shift far smaller than the spectral gap as soon as
    |lam| itself falls below it -- the wanted eigenvalue is still the one nearest 0.
    """
    fam = A.fam
    d = A.shape[0]
    own = A.owner
    if own is not None:
        own.ensure_solver()
    stol = own.solver_tol if own is not None else 1e-12
    smax = own.solver_maxit if own is not None else 400
    if ncv is None:
        ncv = max(20, 2 * nev + 1)          # ARPACK's default
    step = int(min(d, ncv, max(6, 2 * nev + 2)))
    v = np.ones(d, dtype=np.complex128) if v0 is None else np.asarray(v0, dtype=np.complex128)
    cM = M.coeffs
    cA = A.coeffs - sigma * cM              # (A - sigma M)^H is formed by the library when A.op == OP_C
    sig_out = np.conj(sigma) if A.op == OP_C else sigma
    last = None
    gap = np.inf
    total = 0
    failed_runs = 0
    while total < maxiter:
        if nev == 1:
            # one wanted pair: the device stops the factorisation as soon as its Ritz residual is below tol
            Hb, Vb = fam.arnoldi_batch(cA[None, :], cM, step, v[:, None], op=A.op, tol=stol, maxit=smax, ritz_tol=tol, quiet=True)
            H, V = Hb[0], Vb[0]
        else:
            H, V = fam.arnoldi(cA, cM, step, v, op=A.op, tol=stol, maxit=smax, quiet=True)
        total += step
        # inner solves that did not even reach 1e-4 (far outside the range the multigrid hierarchy was built for, or
        # beyond what the mesh resolves) cannot produce Ritz pairs: give up like ARPACK does instead of restarting
        # up to `maxiter` steps of failing solves
        if fam.last_info["n_unconverged"] > 0 and fam.last_info["relres_max"] > 1e-4:
            failed_runs += 1
            if failed_runs >= 2:
                raise EigsError(f"inner solves stalled at relative residual {fam.last_info['relres_max']:.1e}")
        m = step
        while m > 1 and not H[:, m - 1].any():   # steps not taken (early exit on the device): zero columns
            m -= 1
        taken = m
        for j in range(m):                   # invariant subspace: H[j+1,j] == 0
            if H[j + 1, j] == 0:
                m = j + 1
                break
        theta, Yr = np.linalg.eig(H[:m, :m])
        order = np.argsort(-np.abs(theta))
        theta, Yr = theta[order], Yr[:, order]
        k = min(nev, m)
        res = np.abs(H[m, m - 1]) * np.abs(Yr[m - 1, :k])
        X = V[:, :m] @ Yr[:, :k]
        X = X / np.linalg.norm(X, axis=0)
        last = (sig_out + 1.0 / theta[:k], X)
        if m > k:
            gap = abs(1.0 / theta[k])          # crude estimate of the next eigenvalue's modulus
        if np.all(res <= tol * np.abs(theta[:k])) or m < taken or m >= d:
            return last + (gap,) if return_gap else last
        v = X @ np.ones(k)                    # restart with the wanted Ritz vectors
    if last is None:
        raise EigsError("no Ritz pair")
    return last + (gap,) if return_gap else last


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


def _aux_step(L, z, order, nev, v0, v0_adj, update, state):
    """one pass of the loop body shared by householder and mslp (Householder.jl:96-120).
    ``state`` carries |lam| and the gap estimate of the previous pass to choose the regularising shift."""
    L.params[L.eigval] = z
    L.params[L.auxval] = 0
    A = L(z)
    M = L.term_operator(len(L.terms) - 1, -1.0)          # M = -L.terms[end].coeff
    sigma = 0.0
    gap_prev, lam_prev = state.get("gap", np.inf), state.get("lam", np.inf)
    if np.isfinite(gap_prev) and lam_prev < 1e-4 * gap_prev:
        sigma = 1e-5 * gap_prev
    lam, v, gap = eigs(A, M, nev=nev, v0=v0, sigma=sigma, return_gap=True)
    lam_adj, v_adj = eigs(A.H, M.H, nev=nev, v0=v0_adj, sigma=sigma)
    state["gap"] = gap if np.isfinite(gap) else gap_prev
    state["lam"] = float(np.min(np.abs(lam)))
    idx = np.argsort(np.abs(lam)); lam, v = lam[idx], v[:, idx]
    idx = np.argsort(np.abs(lam_adj)); lam_adj, v_adj = lam_adj[idx], v_adj[:, idx]
    cand = []
    L.active = [L.auxval, L.eigval]
    try:
        # fewer than nev pairs come back when the Krylov space is numerically invariant (z on an eigenvalue: one
        # shift-invert step already spans it); the candidates are then the pairs that exist
        for i in range(min(nev, len(lam), len(lam_adj))):
            L.params[L.auxval] = lam[i]
            sol = Solution(L.params, v[:, i], v_adj[:, i], L.auxval)
            perturb_(sol, L, L.eigval, order, mode="householder")
            cand.append(update(sol.eigval_pert[f"{L.eigval}/Taylor"]))
    finally:
        L.active = [L.eigval]
    return lam, v, v_adj, cand


def _normalise(L, v0, v0_adj):
    """Householder.jl:189-190"""
    M = L.term_operator(len(L.terms) - 1, -1.0)
    v0 = v0 / np.sqrt(np.vdot(v0, M @ v0))
    saved = L.active, L.mode
    L.active, L.mode = [L.eigval], "all"
    try:
        v0_adj = v0_adj / np.conj(np.vdot(v0_adj, L(L.params[L.eigval], 1) @ v0))
    finally:
        L.active, L.mode = saved
    return v0, v0_adj


def _aux_step_slots(L, fam, z, order, update, state):
    """`_aux_step` for nev = 1 with the eigenvector pair resident in HBM: v0 / v0_adj are column 0 of the slots _SV / _SW of the family,
    the Ritz vectors of the two Arnoldi processes go to column 0 of _SXR / _SXL and the perturbation step reads them there.  Returns
    (lam, candidate update)."""
    L.params[L.eigval] = z
    L.params[L.auxval] = 0
    T = len(L.terms)
    cA = np.array([L.coefficients(z)])
    cM = np.zeros(T, dtype=np.complex128)
    cM[T - 1] = -1.0                                      # M = -L.terms[end].coeff
    sigma = 0.0
    gap_prev, lam_prev = state.get("gap", np.inf), state.get("lam", np.inf)
    if np.isfinite(gap_prev) and lam_prev < 1e-4 * gap_prev:
        sigma = 1e-5 * gap_prev
    right = eigs_many_slots(fam, cA, cM, _SV, [0], OP_N, [sigma], _SXR, stol=L.solver_tol, smax=L.solver_maxit)[0]
    left = eigs_many_slots(fam, cA, cM, _SW, [0], OP_C, [sigma], _SXL, stol=L.solver_tol, smax=L.solver_maxit)[0]
    for r in (right, left):
        if isinstance(r, Exception):
            raise r
    lam, gap = right
    state["gap"] = gap if np.isfinite(gap) else gap_prev
    state["lam"] = float(abs(lam))
    L.params[L.auxval] = lam
    try:
        cand = update(eigval_series_slots(L, L.auxval, L.eigval, order, _SXR, 0, _SXL, 0))
    finally:
        L.active = [L.eigval]
    return lam, cand


def _slots_begin(L, fam, v0, v0_adj):
    """the eigenvector pair of a single-start iteration into column 0 of the slots (v0_adj None: conj(v0), Householder.jl:84-86)"""
    d = L.size()
    fam.slot_write(_SV, np.ones((d, 1), dtype=np.complex128) if v0 is None else np.asarray(v0, dtype=np.complex128).reshape(d, 1))
    if v0_adj is None:
        fam.slot_write(_SW, None, ncols_total=1)
        fam.slot_axpby(_SW, [0], _SV, [0], alpha=1.0, beta=0.0, conj_src=True)
    else:
        fam.slot_write(_SW, np.asarray(v0_adj, dtype=np.complex128).reshape(d, 1))
    fam.slot_write(_SXR, None, ncols_total=1)
    fam.slot_write(_SXL, None, ncols_total=1)


def _slots_finish(L, fam):
    """`_normalise` (Householder.jl:189-190) on the slots, then the pair back to the host: (v0, v0_adj)"""
    T = len(L.terms)
    cM = np.zeros(T, dtype=np.complex128)
    cM[T - 1] = -1.0
    nv = fam.slot_forms(cM, _SV, [0], _SV, [0])
    fam.slot_axpby(_SV, [0], _SV, [0], alpha=1.0 / np.sqrt(nv), beta=0.0)
    saved = L.active, L.mode
    L.active, L.mode = [L.eigval], "all"
    try:
        cD = np.array([L.coefficients(L.params[L.eigval], 1)])
    finally:
        L.active, L.mode = saved
    dw = fam.slot_forms(cD, _SW, [0], _SV, [0])
    fam.slot_axpby(_SW, [0], _SW, [0], alpha=1.0 / np.conj(dw), beta=0.0)
    return fam.slot_read(_SV, 0, 1)[:, 0], fam.slot_read(_SW, 0, 1)[:, 0]


def householder(L, z, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, nev=1, v0=None, v0_adj=None, output=False, resident=True):
    """sol, n, flag = householder(L, z; maxiter, tol, relax, lam_tol, order, nev, v0, v0_adj, output)
    (Householder.jl:70-192; flags 1 converged / 0 slow / -1 maxiter / -4 eigs / -6 singular / -5 NaN)
    nev = 1 (the default) runs the device-resident iteration of `householder_many` for the one start value: the eigenvector pair stays
    in HBM between the Arnoldi processes, the perturbation step and the update (0.49 -> 0.26 s per call at 1M DoF); resident=False or
    nev > 1: the vectors pass through host memory between the library calls, as below."""
    if resident and nev == 1:
        d = L.size()
        return householder_many(L, [z], maxiter=maxiter, tol=tol, relax=relax, lam_tol=lam_tol, order=order,
                                v0s=None if v0 is None else np.asarray(v0, dtype=np.complex128).reshape(d, 1),
                                v0s_adj=None if v0_adj is None else np.asarray(v0_adj, dtype=np.complex128).reshape(d, 1), output=output,
                                _single=True)[0]
    z = complex(z)
    z0 = complex(np.inf)
    lam = np.inf
    n = 0
    active, mode = L.active, L.mode
    d = L.size()
    v0 = np.ones(d, dtype=np.complex128) if v0 is None else np.asarray(v0, dtype=np.complex128)
    v0_adj = np.conj(v0) if v0_adj is None else np.asarray(v0_adj, dtype=np.complex128)
    flag = 1
    history = []
    state = {}
    try:
        while abs(z - z0) > tol and n < maxiter:
            if output:
                print(n, "\t\t", abs(lam), "\t", abs(z - z0), "\t", z)
            history.append(z)
            z0 = z
            lams, v, v_adj, dzs = _aux_step(L, z, order, nev, v0, v0_adj,
                                            lambda c: householder_update([factorial(i) * ci for i, ci in enumerate(c)]), state)
            i = int(np.argsort(np.abs(dzs))[0])
            lam = lams[i]
            L.params[L.auxval] = lam
            z = z + relax * dzs[i]
            v0 = (1 - relax) * v0 + relax * v[:, i]
            v0_adj = (1 - relax) * v0_adj + relax * v_adj[:, i]
            n += 1
    except EigsError:
        flag = -4
    except WaeError as e:
        flag = -6 if e.code == -2 else -2
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
    v0, v0_adj = _normalise(L, v0, v0_adj)
    sol = Solution(L.params, v0, v0_adj, L.eigval)
    sol.history = history
    return sol, n, flag


def mslp(L, z, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, nev=1, v0=None, v0_adj=None, num_order=1,
         scale=1.0, output=False, resident=True):
    """sol, n, flag = mslp(L, z; ...)   (iterative_solvers.jl:93-252)
    nev = 1 (the default): the eigenvector pair stays in HBM between the Arnoldi processes, the perturbation step and the update
    (`_aux_step_slots`); resident=False or nev > 1: through host memory (`_aux_step`)."""
    z = complex(z) * scale
    tol = tol * scale
    z0 = complex(np.inf)
    lam = np.inf
    lam0 = np.inf
    n = 0
    active, mode = L.active, L.mode
    d = L.size()
    on_dev = bool(resident) and nev == 1
    if not on_dev:
        v0 = np.ones(d, dtype=np.complex128) if v0 is None else np.asarray(v0, dtype=np.complex128)
        v0_adj = np.conj(v0) if v0_adj is None else np.asarray(v0_adj, dtype=np.complex128)
    flag = itsol_converged
    if L.terms[-1].operator != "__aux__":          # iterative_solvers.jl:119-123
        L.push(Term(-sp.identity(d, dtype=np.complex128, format="csr"), (pow1,), (("__aux__",),), "__aux__", "__aux__"))
        L.auxval = "__aux__"
    history = []
    state = {}
    fam = None
    if on_dev:
        fam = L.ensure_solver()                    # (after the push: the family on the device has the aux term)
        _slots_begin(L, fam, v0, v0_adj)
    try:
        while abs(z - z0) > tol and n < maxiter:
            if output:
                print(n, "\t\t", abs(z - z0) / scale, "\t", z / scale)
            history.append(z)
            pades = []

            def upd(coeffs):
                num, den = pade(coeffs, num_order, order - num_order)
                pades.append((num, den))
                roots = poly_roots(num)
                return roots[np.argsort(np.abs(roots))[0]]
            if on_dev:
                lam1, dz1 = _aux_step_slots(L, fam, z, order, upd, state)
                lams, dzs = [lam1], [dz1]
            else:
                lams, v, v_adj, dzs = _aux_step(L, z, order, nev, v0, v0_adj, upd, state)
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
            if on_dev:
                fam.slot_axpby(_SV, [0], _SXR, [0], alpha=relax, beta=1.0 - relax)
                fam.slot_axpby(_SW, [0], _SXL, [0], alpha=relax, beta=1.0 - relax)
            else:
                v0 = (1 - relax) * v0 + relax * v[:, i]
                v0_adj = (1 - relax) * v0_adj + relax * v_adj[:, i]
            n += 1
    except EigsError:
        flag = itsol_arpack_exception
    except WaeError as e:
        flag = itsol_singular_exception if e.code == -2 else itsol_unknown
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
    v0, v0_adj = _slots_finish(L, fam) if on_dev else _normalise(L, v0, v0_adj)
    sol = Solution(L.params, v0, v0_adj, L.eigval)
    sol.history = history
    return sol, n, flag


def padesolve(L, z, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, nev=1, v0=None, v0_adj=None, output=False,
              num_order=1):
    """sol, n, flag = padesolve(L, z; ...)   (Householder.jl:205-355): the MSLP/Padé iteration with `householder`'s
    flag conventions (1 converged, 0 slow, -1 maxiter, ...), without the automatic aux term and without `scale`."""
    if L.terms[-1].operator != "__aux__":
        raise ValueError("padesolve needs a family whose last term is the __aux__ term (Householder.jl:225)")
    sol, n, flag = mslp(L, z, maxiter=maxiter, tol=tol, relax=relax, lam_tol=lam_tol, order=order, nev=nev, v0=v0,
                        v0_adj=v0_adj, num_order=num_order, scale=1.0, output=output)
    return sol, n, {itsol_converged: 1, itsol_slow_convergence: 0, itsol_maxiter: -1, itsol_isnan: -5, itsol_impossible: -3,
                    itsol_arpack_exception: -4, itsol_singular_exception: -6, itsol_unknown: -2,
                    itsol_arpack_9999: -9999}.get(flag, -2)


def _finish(n, maxiter, z, z0, tol, flag):
    """iterative_solvers.jl:326-342"""
    if flag != itsol_converged:
        return flag
    if n >= maxiter:
        return itsol_maxiter
    if abs(z - z0) <= tol:
        return itsol_converged
    if np.isnan(z):
        return itsol_isnan
    return itsol_impossible


def inveriter(L, z, maxiter=10, tol=0.0, relax=1.0, x0=None, v=None, output=False):
    """iterative_solvers.jl:285-347"""
    d = L.size()
    x0 = np.ones(d, dtype=np.complex128) if x0 is None else np.asarray(x0, dtype=np.complex128)
    v = np.ones(d, dtype=np.complex128) if v is None else np.asarray(v, dtype=np.complex128)
    x0 = x0 / np.vdot(v, x0)
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            if output:
                print(n, "\t\t", abs(z - z0), "\t", z)
            z0 = z
            u = L(z, 0).solve(L(z, 1) @ x0, guess=x0, quiet=True)        # the solution is x0/(ω*-z) to first order
            z = z0 - np.vdot(v, x0) / np.vdot(v, u)
            x0 = u / np.vdot(v, u)
            n += 1
    except WaeError:
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, x0, [], L.eigval, L.auxval), n, flag


def lancaster(L, z, maxiter=10, tol=0.0, relax=1.0, x0=None, y0=None, output=False):
    """iterative_solvers.jl:378-434"""
    d = L.size()
    x0 = np.ones(d, dtype=np.complex128) if x0 is None else np.asarray(x0, dtype=np.complex128)
    y0 = np.ones(d, dtype=np.complex128) if y0 is None else np.asarray(y0, dtype=np.complex128)
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            A = L(z)
            xi = A.solve(x0, quiet=True)          # (numerically singular by design close to convergence: direction matters)
            eta = A.H.solve(y0, quiet=True)
            z = z0 - np.vdot(eta, L(z, 0) @ xi) / np.vdot(eta, L(z, 1) @ xi)
            n += 1
    except WaeError:
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, np.zeros(d, dtype=np.complex128), [], L.eigval), n, flag


def rf2s(L, z, maxiter=10, tol=0.0, relax=1.0, x0=None, y0=None, output=False):
    """iterative_solvers.jl:548-614"""
    d = L.size()
    if x0 is None:
        x0 = np.zeros(d, dtype=np.complex128); x0[0] = 1
    if y0 is None:
        y0 = np.zeros(d, dtype=np.complex128); y0[0] = 1
    x0 = np.asarray(x0, dtype=np.complex128); y0 = np.asarray(y0, dtype=np.complex128)
    x0 = x0 / np.sqrt(np.vdot(x0, x0)); y0 = y0 / np.sqrt(np.vdot(y0, y0))
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            A = L(z)
            L1 = L(z, 1)
            x0 = A.solve(L1 @ x0, guess=x0, quiet=True)
            y0 = A.H.solve(L1.H @ y0, guess=y0, quiet=True)
            x0 = x0 / np.sqrt(np.vdot(x0, x0)); y0 = y0 / np.sqrt(np.vdot(y0, y0))
            idx = 0
            z00 = complex(np.inf)
            while abs(z - z00) > tol and idx < 10:
                z00 = z
                z = z - np.vdot(y0, L(z) @ x0) / np.vdot(y0, L(z, 1) @ x0)
                idx += 1
            n += 1
    except WaeError:
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, x0, y0, L.eigval), n, flag


def traceiter(L, z, maxiter=10, tol=0.0, relax=1.0, output=False):
    """iterative_solvers.jl:463-517: d solves per step (all d unit vectors as one batched device solve);
    small problems only, as in the reference."""
    d = L.size()
    z = complex(z)
    z0 = complex(np.inf)
    n = 0
    flag = itsol_converged
    try:
        while abs(z - z0) > tol and n < maxiter:
            z0 = z
            L1 = L(z, 1) @ np.eye(d, dtype=np.complex128)
            X = L(z).solve(L1)
            dz = -1.0 / np.trace(X)
            z = z0 + relax * dz
            n += 1
    except WaeError:
        flag = itsol_unknown
    flag = _finish(n, maxiter, z, z0, tol, flag)
    return Solution(L.params, [], [], L.eigval), n, flag


def count_poles_and_zeros(L, G, N=16, output=False):
    """beyn.jl:355-368"""
    from .beyn import gauss_points
    d = L.size()
    zs, ws = gauss_points(G, N)
    s = 0j
    eye = np.eye(d, dtype=np.complex128)
    for z, w in zip(zs, ws):
        X = L(z).solve(L(z, 1) @ eye)
        s += np.trace(X) * w
    return s / 2 / np.pi / 1j


# ------------------------------------------------------------------------------------------------------
# many start values at once: the Newton-type refinement of all the estimates a Beyn solve returned, in lock-step.
# A single-column solve is latency-bound on the device (0.65 ms per Krylov iteration whether the batch holds one column
# or eight), so refining 8 estimates together costs about as much as refining one.  Same iteration as `householder`
# (Householder.jl:70-192) per start value; only the order of the device work changes.
# ------------------------------------------------------------------------------------------------------
def eigs_many(fam, cA, cM, v0s, op, sigmas, nev=1, tol=1e-12, maxiter=300, stol=1e-12, smax=400, stats=None):
    """`eigs` for nsys operator pairs (A_s - sigma_s M, M) in lock-step.  cA: (nsys, T) coefficient rows, cM: (T,).
    Returns per system (lam[nev], V[d, nev], gap) or an EigsError instance."""
    cA = np.asarray(cA, dtype=np.complex128)
    nsys, d = cA.shape[0], fam.d
    step = int(min(d, max(6, 2 * nev + 2)))
    cAs = cA - np.asarray(sigmas, dtype=np.complex128)[:, None] * cM[None, :]
    sig_out = np.conj(sigmas) if op == OP_C else np.asarray(sigmas)
    own_v0 = True
    if isinstance(v0s, (list, tuple)):                    # columns given one by one (views): one copy into column-major storage
        V0 = np.empty((d, nsys), dtype=np.complex128, order="F")
        for q, col in enumerate(v0s):
            V0[:, q] = col
    else:                                                 # a column-major d x nsys array is handed to the library as it is (127 MB at 1M DoF
        V0 = np.asarray(v0s, dtype=np.complex128).reshape(d, nsys)      # and 8 start values: no copy); copied only if a restart writes to it
        own_v0 = not (V0 is v0s or V0.base is not None)
        if not V0.flags.f_contiguous:
            V0, own_v0 = np.asfortranarray(V0), True
    out = [None] * nsys
    pending = list(range(nsys))
    total = 0
    while pending and total < maxiter:
        H, V = fam.arnoldi_batch(cAs[pending], cM, step, V0 if len(pending) == nsys else V0[:, pending], op=op, tol=stol, maxit=smax,
                                 ritz_tol=tol if nev == 1 else 0.0, quiet=True)
        total += step
        if stats is not None:
            stats["inner_column_iterations"] = stats.get("inner_column_iterations", 0) + int(fam.last_info.get("iters_total", 0))
        failed = fam.last_info["n_unconverged"] > 0 and fam.last_info["relres_max"] > 1e-4
        still = []
        for q, s in enumerate(pending):
            Hs, Vs = H[q], V[q]
            m = step
            while m > 1 and not Hs[:, m - 1].any():
                m -= 1
            taken = m
            for j in range(m):
                if Hs[j + 1, j] == 0:
                    m = j + 1
                    break
            theta, Yr = np.linalg.eig(Hs[:m, :m])
            order = np.argsort(-np.abs(theta))
            theta, Yr = theta[order], Yr[:, order]
            k = min(nev, m)
            res = np.abs(Hs[m, m - 1]) * np.abs(Yr[m - 1, :k])
            X = Vs[:, :m] @ Yr[:, :k]
            for jj in range(k):                           # (BLAS dot + in-place scaling: np.linalg.norm and a division cost three passes
                X[:, jj] *= 1.0 / np.sqrt(np.vdot(X[:, jj], X[:, jj]).real)      # with temporaries over 16 MB per vector)
            gap = abs(1.0 / theta[k]) if m > k else np.inf
            out[s] = (sig_out[s] + 1.0 / theta[:k], X, gap)
            if not (np.all(res <= tol * np.abs(theta[:k])) or m < taken or m >= d):
                if failed:
                    out[s] = EigsError("inner solves stalled")
                else:
                    if not own_v0:
                        V0, own_v0 = V0.copy(order="F"), True
                    V0[:, s] = X @ np.ones(k)
                    still.append(s)
        pending = still
    return out


def _conjugate_span_coefficients(V):
    """C (ns x ns) with  W = conj(V C)  the start vectors of the left (adjoint) Arnoldi processes when the caller gives none, or None for
    W = conj(V).  `householder` starts from conj(v0) (Householder.jl:84-86), the left eigenvector of a complex-symmetric L(z) -- as long
    as v0^T v0 != 0.  For a (nearly) degenerate pair that fails: a spinning mode e^{im phi} of an annulus has v^T v = 0 and conj(v) is the
    OTHER mode of the pair, so the left process starts orthogonal to what it looks for (four Arnoldi steps instead of two at 1M DoF).
    With several start vectors at hand the left vectors are taken from their conjugate span instead, bi-orthogonal to them in the
    bilinear form: W = conj(Vn G^-1), Vn = V with unit columns, G = Vn^T Vn -- conj(v0) up to scale for an isolated mode, the partner's
    conjugate for a spinning pair.  Only a start: the converged left eigenvectors do not depend on it.  None when G is numerically
    singular (dependent start vectors, or a spinning mode without its partner).  (Two readings of V, no copy of it.)"""
    ns = V.shape[1]
    if ns < 2:
        return None
    nrm = np.sqrt(np.einsum("ij,ij->j", V.real, V.real) + np.einsum("ij,ij->j", V.imag, V.imag))
    nrm[nrm == 0] = 1.0
    G = (V.T @ V) / np.outer(nrm, nrm)
    if not np.all(np.isfinite(G)) or np.linalg.svd(G, compute_uv=False)[-1] < 1e-6:      # (unit columns: |G_ij| <= 1, the bound is absolute)
        return None
    return np.linalg.inv(G) / nrm[:, None]


def _conjugate_span_start(V):
    """conj(V C), C = _conjugate_span_coefficients(V) (host arrays: householder_many_host)"""
    C = _conjugate_span_coefficients(V)
    return np.conj(V) if C is None else np.conj(V @ C)


def householder_many_host(L, zs, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, v0s=None, v0s_adj=None, output=False, stats=None):
    """`householder_many` with every vector passing through host memory between the device calls (wae_arnoldi_shiftinvert_batch,
    wae_perturb, wae_spmv_sum): the form of rounds 2-3, kept as the cross-check of the device-resident one below
    (tests/test_gpu_parity.py) and for `householder_many(..., resident=False)`."""
    import time as _time
    st_ = {"right_arnoldi_seconds": 0.0, "left_arnoldi_seconds": 0.0, "perturbation_seconds": 0.0, "newton_rounds": 0,
           "inner_column_iterations": 0}
    zs = [complex(z) for z in zs]
    ns = len(zs)
    if ns == 0:                                           # an empty batch of start values: nothing to refine
        if stats is not None:
            stats.update(st_)
        return []
    d = L.size()
    fam = L.ensure_solver()
    active, mode = L.active, L.mode
    # (column-major: every start value's vectors are contiguous -- the updates below touch 16 MB columns at 1M DoF)
    V = np.ones((d, ns), dtype=np.complex128, order="F") if v0s is None else np.array(np.asarray(v0s, dtype=np.complex128).reshape(d, ns), order="F")
    W = np.array(_conjugate_span_start(V) if v0s_adj is None else np.asarray(v0s_adj, dtype=np.complex128).reshape(d, ns), order="F")
    z = list(zs)
    z0 = [complex(np.inf)] * ns
    lam = [np.inf] * ns
    n = [0] * ns
    flag = [1] * ns
    hist = [[] for _ in range(ns)]
    state = [dict() for _ in range(ns)]
    T = len(L.terms)
    cM = np.zeros(T, dtype=np.complex128)
    cM[T - 1] = -1.0                                      # M = -L.terms[end].coeff  (Householder.jl:92)
    upd = lambda c: householder_update([factorial(i) * ci for i, ci in enumerate(c)])   # noqa: E731
    while True:
        act = [s for s in range(ns) if flag[s] == 1 and abs(z[s] - z0[s]) > tol and n[s] < maxiter]
        if not act:
            break
        cA, sig = [], []
        for s in act:
            hist[s].append(z[s])
            z0[s] = z[s]
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = 0
            cA.append(L.coefficients(z[s]))
            gp, lp = state[s].get("gap", np.inf), state[s].get("lam", np.inf)
            sig.append(1e-5 * gp if (np.isfinite(gp) and lp < 1e-4 * gp) else 0.0)
        cA = np.array(cA)
        try:
            st_["newton_rounds"] += 1
            t_ = _time.perf_counter()
            allact = len(act) == ns
            right = eigs_many(fam, cA, cM, V if allact else [V[:, s] for s in act], OP_N, sig, stol=L.solver_tol, smax=L.solver_maxit, stats=st_)
            st_["right_arnoldi_seconds"] += _time.perf_counter() - t_
            t_ = _time.perf_counter()
            left = eigs_many(fam, cA, cM, W if allact else [W[:, s] for s in act], OP_C, sig, stol=L.solver_tol, smax=L.solver_maxit, stats=st_)
            st_["left_arnoldi_seconds"] += _time.perf_counter() - t_
        except WaeError as e:
            for s in act:
                flag[s] = -6 if e.code == -2 else -2
            break
        for q, s in enumerate(act):
            if isinstance(right[q], Exception) or isinstance(left[q], Exception):
                flag[s] = -4
                continue
            lam_r, v_r, gap = right[q]
            lam_l, v_l, _ = left[q]
            state[s]["gap"] = gap if np.isfinite(gap) else state[s].get("gap", np.inf)
            state[s]["lam"] = float(np.min(np.abs(lam_r)))
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = lam_r[0]
            L.active = [L.auxval, L.eigval]
            try:
                t_ = _time.perf_counter()
                sol = Solution(L.params, v_r[:, 0], v_l[:, 0], L.auxval)
                perturb_(sol, L, L.eigval, order, mode="householder")
                dz = upd(sol.eigval_pert[f"{L.eigval}/Taylor"])
                st_["perturbation_seconds"] += _time.perf_counter() - t_
            except WaeError as e:
                flag[s] = -6 if e.code == -2 else -2
                continue
            finally:
                L.active = [L.eigval]
            lam[s] = lam_r[0]
            if output:
                print(s, n[s], "\t", abs(lam[s]), "\t", abs(dz), "\t", z[s])
            z[s] = z[s] + relax * dz
            if relax == 1.0:
                V[:, s] = v_r[:, 0]
                W[:, s] = v_l[:, 0]
            else:
                V[:, s] = (1 - relax) * V[:, s] + relax * v_r[:, 0]
                W[:, s] = (1 - relax) * W[:, s] + relax * v_l[:, 0]
            n[s] += 1
    # Householder.jl:189-190 for all start values at once: v / sqrt(v' M v), v_adj / conj(v_adj' L'(z) v) -- two batched operator
    # products (the same term coefficients `_normalise` forms one start value at a time)
    t_ = _time.perf_counter()
    cMn = np.zeros(T, dtype=np.complex128)
    cMn[T - 1] = -1.0
    MV = fam.spmv(cMn, V)
    for s in range(ns):
        V[:, s] *= 1.0 / np.sqrt(np.vdot(V[:, s], MV[:, s]))          # (a complex in-place division is ten times slower)
    del MV
    cD = np.zeros((ns, T), dtype=np.complex128)
    saved_p = dict(L.params)
    L.active, L.mode = [L.eigval], "all"
    try:
        for s in range(ns):
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = lam[s] if np.isfinite(lam[s]) else 0
            cD[s] = L.coefficients(z[s], 1)
    finally:
        L.active, L.mode = active, mode
        L.params.update(saved_p)
    DV = fam.spmv(cD, V)
    for s in range(ns):
        W[:, s] *= 1.0 / np.conj(np.vdot(W[:, s], DV[:, s]))
    del DV
    st_["normalisation_seconds"] = _time.perf_counter() - t_
    out = []
    for s in range(ns):
        f = flag[s]
        L.params[L.eigval] = z[s]
        L.params[L.auxval] = lam[s] if np.isfinite(lam[s]) else 0
        if f == 1:
            hist[s].append(z[s])
            if n[s] >= maxiter:
                f = -1
            elif abs(lam[s]) <= lam_tol:
                f = 1
            elif abs(z[s] - z0[s]) <= tol:
                f = 0
            elif np.isnan(z[s]):
                f = -5
            else:
                f = -3
        L.active, L.mode = active, mode
        sol = Solution(L.params, V[:, s], W[:, s], L.eigval)        # (columns of the local column-major arrays: contiguous views, no copies)
        sol.history = hist[s]
        out.append((sol, n[s], f))
    L.active, L.mode = active, mode
    if stats is not None:
        stats.update(st_)
    return out


# slots of the family the lock-step Newton iteration keeps its vectors in (the upper half of DeviceFamily.NSLOTS: 0-3 stay the caller's)
_SV, _SW, _SXR, _SXL = 4, 5, 6, 7


def eigs_many_slots(fam, cA, cM, v0_slot, cols, op, sigmas, out_slot, tol=1e-12, maxiter=300, stol=1e-12, smax=400, stats=None):
    """`eigs_many` (nev = 1) on device-resident vectors: start vectors = columns `cols` of slot v0_slot, the normalised Ritz vector of
    system q is written to column cols[q] of out_slot; only the small Hessenberg matrices come back to the host.  Returns per system
    (lam, gap) or an EigsError instance."""
    cA = np.asarray(cA, dtype=np.complex128)
    nsys, d = cA.shape[0], fam.d
    step = int(min(d, 6))
    cAs = cA - np.asarray(sigmas, dtype=np.complex128)[:, None] * cM[None, :]
    sig_out = np.conj(sigmas) if op == OP_C else np.asarray(sigmas)
    out = [None] * nsys
    pending = list(range(nsys))
    total = 0
    src = v0_slot
    while pending and total < maxiter:
        H = fam.arnoldi_slots(cAs[pending], cM, step, src, [cols[s] for s in pending], op=op, tol=stol, maxit=smax, ritz_tol=tol, quiet=True)
        total += step
        if stats is not None:
            stats["inner_column_iterations"] = stats.get("inner_column_iterations", 0) + int(fam.last_info.get("iters_total", 0))
        failed = fam.last_info["n_unconverged"] > 0 and fam.last_info["relres_max"] > 1e-4
        Y = np.zeros((len(pending), step + 1), dtype=np.complex128)
        ny = 1
        still = []
        for q, s in enumerate(pending):
            Hs = H[q]
            m = step
            while m > 1 and not Hs[:, m - 1].any():
                m -= 1
            taken = m
            for j in range(m):
                if Hs[j + 1, j] == 0:
                    m = j + 1
                    break
            theta, Yr = np.linalg.eig(Hs[:m, :m])
            order = np.argsort(-np.abs(theta))
            theta, Yr = theta[order], Yr[:, order]
            res = np.abs(Hs[m, m - 1]) * np.abs(Yr[m - 1, 0])
            Y[q, :m] = Yr[:, 0]
            ny = max(ny, m)
            gap = abs(1.0 / theta[1]) if m > 1 else np.inf
            out[s] = (sig_out[s] + 1.0 / theta[0], gap)
            if not (res <= tol * np.abs(theta[0]) or m < taken or m >= d):
                if failed:
                    out[s] = EigsError("inner solves stalled")
                else:
                    still.append(s)                           # restart from the Ritz vector (now in out_slot)
        fam.ritz_to_slot(Y[:, :ny], out_slot, [cols[s] for s in pending], normalise=True)
        pending = still
        src = out_slot
    return out


def householder_many(L, zs, maxiter=10, tol=0.0, relax=1.0, lam_tol=np.inf, order=1, v0s=None, v0s_adj=None, output=False, stats=None, resident=True,
                     _single=False):
    """[(sol, n, flag), ...] = householder_many(L, zs; ...): `householder` (Householder.jl:70-192) for several start values, the device
    work (two shift-invert Arnoldi processes per Newton step and start value) batched over the start values, and every vector of the
    iteration resident in HBM: the estimates go to the device once (wae_slot_write), the Arnoldi processes start from slot columns and
    leave their Ritz vectors in slot columns (wae_arnoldi_shiftinvert_slots, wae_arnoldi_ritz_to_slot), the perturbation step reads them
    there (wae_perturb_slots), the relaxed update and the two normalisations of Householder.jl:173-176,189-190 are slot operations
    (wae_slot_axpby, wae_slot_forms), and the eigenvectors come back once at the end.  The host sees Hessenberg matrices and scalars.
    (Through host memory -- householder_many_host, resident=False -- a step of 8 start values at 1M DoF moved 2 GB over PCIe and the
    GPU idled 45 % of the time.)
    stats (optional dict): receives the seconds spent in the right / left Arnoldi processes and in the perturbation step, the
    number of lock-step Newton rounds and the inner (Krylov) column-iterations."""
    if not resident:
        return householder_many_host(L, zs, maxiter=maxiter, tol=tol, relax=relax, lam_tol=lam_tol, order=order, v0s=v0s, v0s_adj=v0s_adj,
                                     output=output, stats=stats)
    import time as _time
    st_ = {"right_arnoldi_seconds": 0.0, "left_arnoldi_seconds": 0.0, "perturbation_seconds": 0.0, "newton_rounds": 0,
           "inner_column_iterations": 0}
    zs = [complex(z) for z in zs]
    ns = len(zs)
    if ns == 0:                                           # an empty batch of start values: nothing to refine
        if stats is not None:
            stats.update(st_)
        return []
    d = L.size()
    fam = L.ensure_solver()
    active, mode = L.active, L.mode
    allc = list(range(ns))
    t_ = _time.perf_counter()
    if v0s is None:
        fam.slot_write(_SV, np.ones((d, ns), dtype=np.complex128, order="F"))
    else:
        fam.slot_write(_SV, np.asarray(v0s, dtype=np.complex128).reshape(d, ns))
    if v0s_adj is None:
        # the conjugate-span start W = conj(V C): the small matrix C from the caller's array, the combination on the device (column i of V,
        # conjugated, into every column of W with its weight)
        Cs = None if v0s is None else _conjugate_span_coefficients(np.asarray(v0s, dtype=np.complex128).reshape(d, ns))
        fam.slot_write(_SW, None, ncols_total=ns)
        if Cs is None:
            fam.slot_axpby(_SW, allc, _SV, allc, alpha=1.0, beta=0.0, conj_src=True)
        else:
            for i in range(ns):
                fam.slot_axpby(_SW, allc, _SV, [i] * ns, alpha=np.conj(Cs[i, :]), beta=0.0 if i == 0 else 1.0, conj_src=True)
    else:
        fam.slot_write(_SW, np.asarray(v0s_adj, dtype=np.complex128).reshape(d, ns))
    fam.slot_write(_SXR, None, ncols_total=ns)
    fam.slot_write(_SXL, None, ncols_total=ns)
    st_["upload_seconds"] = _time.perf_counter() - t_
    z = list(zs)
    z0 = [complex(np.inf)] * ns
    lam = [np.inf] * ns
    n = [0] * ns
    flag = [1] * ns
    hist = [[] for _ in range(ns)]
    state = [dict() for _ in range(ns)]
    T = len(L.terms)
    cM = np.zeros(T, dtype=np.complex128)
    cM[T - 1] = -1.0                                      # M = -L.terms[end].coeff  (Householder.jl:92)
    upd = lambda c: householder_update([factorial(i) * ci for i, ci in enumerate(c)])   # noqa: E731
    while True:
        act = [s for s in range(ns) if flag[s] == 1 and abs(z[s] - z0[s]) > tol and n[s] < maxiter]
        if not act:
            break
        cA, sig = [], []
        for s in act:
            if output and _single:                            # (the line `householder` prints per iteration, Householder.jl:95)
                print(n[s], "\t\t", abs(lam[s]), "\t", abs(z[s] - z0[s]), "\t", z[s])
            hist[s].append(z[s])
            z0[s] = z[s]
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = 0
            cA.append(L.coefficients(z[s]))
            gp, lp = state[s].get("gap", np.inf), state[s].get("lam", np.inf)
            sig.append(1e-5 * gp if (np.isfinite(gp) and lp < 1e-4 * gp) else 0.0)
        cA = np.array(cA)
        try:
            st_["newton_rounds"] += 1
            t_ = _time.perf_counter()
            right = eigs_many_slots(fam, cA, cM, _SV, act, OP_N, sig, _SXR, stol=L.solver_tol, smax=L.solver_maxit, stats=st_)
            st_["right_arnoldi_seconds"] += _time.perf_counter() - t_
            t_ = _time.perf_counter()
            left = eigs_many_slots(fam, cA, cM, _SW, act, OP_C, sig, _SXL, stol=L.solver_tol, smax=L.solver_maxit, stats=st_)
            st_["left_arnoldi_seconds"] += _time.perf_counter() - t_
        except WaeError as e:
            for s in act:
                flag[s] = -6 if e.code == -2 else -2
            break
        moved = []
        for q, s in enumerate(act):
            if isinstance(right[q], Exception) or isinstance(left[q], Exception):
                flag[s] = -4
                continue
            lam_r, gap = right[q]
            state[s]["gap"] = gap if np.isfinite(gap) else state[s].get("gap", np.inf)
            state[s]["lam"] = float(abs(lam_r))
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = lam_r
            try:
                t_ = _time.perf_counter()
                series = eigval_series_slots(L, L.auxval, L.eigval, order, _SXR, s, _SXL, s)
                dz = upd(series)
                st_["perturbation_seconds"] += _time.perf_counter() - t_
            except WaeError as e:
                flag[s] = -6 if e.code == -2 else -2
                continue
            finally:
                L.active = [L.eigval]
            lam[s] = lam_r
            if output and not _single:
                print(s, n[s], "\t", abs(lam[s]), "\t", abs(dz), "\t", z[s])
            z[s] = z[s] + relax * dz
            moved.append(s)
            n[s] += 1
        if moved:                                         # v0 = (1 - relax) v0 + relax v  (Householder.jl:173-176), both vectors, on the device
            fam.slot_axpby(_SV, moved, _SXR, moved, alpha=relax, beta=1.0 - relax)
            fam.slot_axpby(_SW, moved, _SXL, moved, alpha=relax, beta=1.0 - relax)
    # Householder.jl:189-190 for all start values at once: v / sqrt(v' M v), v_adj / conj(v_adj' L'(z) v) -- two batched forms
    t_ = _time.perf_counter()
    nv = fam.slot_forms(cM, _SV, allc, _SV, allc)
    fam.slot_axpby(_SV, allc, _SV, allc, alpha=1.0 / np.sqrt(nv), beta=0.0)
    cD = np.zeros((ns, T), dtype=np.complex128)
    saved_p = dict(L.params)
    L.active, L.mode = [L.eigval], "all"
    try:
        for s in range(ns):
            L.params[L.eigval] = z[s]
            L.params[L.auxval] = lam[s] if np.isfinite(lam[s]) else 0
            cD[s] = L.coefficients(z[s], 1)
    finally:
        L.active, L.mode = active, mode
        L.params.update(saved_p)
    dw = fam.slot_forms(cD, _SW, allc, _SV, allc)
    fam.slot_axpby(_SW, allc, _SW, allc, alpha=1.0 / np.conj(dw), beta=0.0)
    st_["normalisation_seconds"] = _time.perf_counter() - t_
    t_ = _time.perf_counter()
    V = fam.slot_read(_SV, 0, ns)
    W = fam.slot_read(_SW, 0, ns)
    st_["download_seconds"] = _time.perf_counter() - t_
    out = []
    for s in range(ns):
        f = flag[s]
        L.params[L.eigval] = z[s]
        L.params[L.auxval] = lam[s] if np.isfinite(lam[s]) else 0
        if f == 1:
            hist[s].append(z[s])
            if n[s] >= maxiter:
                f = -1
            elif abs(lam[s]) <= lam_tol:
                f = 1
            elif abs(z[s] - z0[s]) <= tol:
                f = 0
            elif np.isnan(z[s]):
                f = -5
            else:
                f = -3
        L.active, L.mode = active, mode
        sol = Solution(L.params, V[:, s], W[:, s], L.eigval)        # (columns of the local column-major arrays: contiguous views, no copies)
        sol.history = hist[s]
        out.append((sol, n[s], f))
    L.active, L.mode = active, mode
    if stats is not None:
        stats.update(st_)
    return out
