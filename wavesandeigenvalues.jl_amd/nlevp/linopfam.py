"""Term / LinearOperatorFamily / Solution with the reference's semantics (src/NLEVP/LinOpFam.jl), backed by a
device-resident family handle.

Difference to the reference by design: calling ``L(z)`` does NOT materialise a sparse matrix
(LinOpFam.jl:499-521 allocates T+1 of them per call); it returns a light ``Operator`` view = (device handle,
T coefficients) that supports what the reference's solvers do with the matrix: ``A @ x`` (fused multi-term
SpMV on the GPU), ``A.solve(b)`` (Julia ``A\\b``: multigrid-GMRES on the GPU), ``A.H`` (Julia ``A'``), ``shape``.
"""
from __future__ import annotations

import copy
import ctypes as C
from math import factorial

import numpy as np
import scipy.sparse as sp

from .. import _lib
from .._lib import OP_C, OP_N, SolveInfo, check, zptr

NaN = complex(float("nan"), float("nan"))


class Term:
    """LinOpFam.jl:16-35, 61-75"""

    def __init__(self, coeff, func, params, symbol="", operator=""):
        if isinstance(symbol, str) and operator == "" and symbol != "":
            # Term(coeff, func, params, operator) form (LinOpFam.jl:61): 4th positional argument is the operator name
            operator, symbol = symbol, ""
        self.coeff = coeff
        self.func = tuple(func)
        self.params = tuple(tuple(p) for p in params)
        self.symbol = symbol
        self.operator = operator
        varlist = []
        for par in self.params:
            for var in par:
                if var not in varlist:
                    varlist.append(var)
        self.varlist = varlist

    def scalar(self, d):
        """the scalar part of the term functor, LinOpFam.jl:466-477"""
        c = 1.0 + 0j
        for func, pars in zip(self.func, self.params):
            c *= func(*[d[p][0] for p in pars], *[d[p][1] for p in pars])
        return complex(c)


class Solution:
    """LinOpFam.jl:95-112; callable like the reference's (LinOpFam.jl:684-699)."""

    def __init__(self, params, v, v_adj, eigval, auxval=""):
        self.params = copy.deepcopy(params)
        self.v = v
        self.v_adj = v_adj
        self.eigval = eigval
        self.eigval_pert = {}
        self.v_pert = {}
        self.auxval = auxval

    def __call__(self, param, eps, L=0, M=0, vector=False):
        key = f"{param}/[{L}/{M}]"
        if key not in self.eigval_pert or (vector and key not in self.v_pert):
            pade_(self, param, L, M, vector=vector)
        a, b = self.eigval_pert[key]
        de = eps - self.params[param]
        eigval = polyval(a, de) / polyval(b, de)
        if not vector:
            return eigval
        A, B = self.v_pert[key]                       # (L+1, d) and (M+1, d) coefficient arrays
        return eigval, polyval(list(A), de) / polyval(list(B), de)


class UnconvergedWarning(RuntimeWarning):
    """an inner multigrid-GMRES solve stopped at its iteration limit or stagnated above the requested tolerance"""


class DeviceFamily:
    """Owner of a ``wae_family`` handle: all term matrices resident in HBM (include/waehip.h).

    ``strict`` (default True): the reference's ``\\`` is a direct LU that either solves or throws, so an inner solve that
    did not reach its tolerance must not pass silently.  Contour integrals (``beyn_moments*``) raise ``WaeError`` -- a
    stalled quadrature point corrupts every eigenpair; single solves, Arnoldi processes and the perturbation recurrence
    emit an ``UnconvergedWarning`` (the Newton-type solvers run them on numerically singular operators by design and
    judge the outcome themselves: they pass ``quiet=True``).  ``strict=False`` restores the silent behaviour."""

    def __init__(self, mats, device=0, symmetry_tol=0.0):
        """symmetry_tol: opts[0] of wae_family_create_opts -- 0: a term counts as symmetric (its transposed products run on the
        forward path) only if its mirror entries are equal bit for bit; t > 0: if they agree to t of the row scale (finite-element
        matrices, symmetric up to the order of their element sums)."""
        lib = _lib.lib()
        self.T = len(mats)
        self.d = mats[0].shape[0]
        csr = []
        for A in mats:
            A = sp.csr_matrix(A, dtype=np.complex128)
            A.sum_duplicates()
            A.sort_indices()
            csr.append(A)
        self._ptr_arrays = [A.indptr.astype(np.int32) for A in csr]
        self._idx_arrays = [A.indices.astype(np.int32) for A in csr]
        self._val_arrays = [np.ascontiguousarray(A.data, dtype=np.complex128) for A in csr]
        ptrs = (C.c_void_p * self.T)(*[a.ctypes.data for a in self._ptr_arrays])
        idxs = (C.c_void_p * self.T)(*[a.ctypes.data for a in self._idx_arrays])
        vals = (C.c_void_p * self.T)(*[a.ctypes.data for a in self._val_arrays])
        self.handle = C.c_void_p()
        opts = (C.c_double * 1)(float(symmetry_tol))
        check(lib.wae_family_create_opts(C.byref(self.handle), self.d, self.T, 4, 0, _lib.CSR, ptrs, idxs, vals, device, opts, 1))
        self.device = device
        self.solver_ready = False
        self.last_info = None
        self.last_code = 0
        self.strict = True

    def _report(self, code, info, what, fatal=False, quiet=False):
        """record the outcome of a library call that solved linear systems; see the class docstring"""
        self.last_info = info.as_dict() if hasattr(info, "as_dict") else dict(info)
        self.last_code = code
        n = self.last_info.get("n_unconverged", 0)
        if n > 0 and self.strict and not quiet:
            msg = (f"{what}: {n} inner solve(s) did not reach the tolerance (largest relative residual "
                   f"{self.last_info.get('relres_max', float('nan')):.2e}, "
                   f"{'stagnation' if code == _lib.WAE_WARN_STAGNATION else 'iteration limit'})")
            if fatal:
                raise _lib.WaeError(code, msg)
            import warnings
            warnings.warn(msg, UnconvergedWarning, stacklevel=3)
        return code

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            _lib.lib().wae_family_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- operator application ------------------------------------------------------------------------
    def spmv(self, coeffs, X, op=OP_N):
        c = np.ascontiguousarray(coeffs, dtype=np.complex128)
        X = np.asarray(X, dtype=np.complex128)
        one = X.ndim == 1
        Xf = np.asfortranarray(X.reshape(self.d, -1))
        Y = np.empty_like(Xf, order="F")
        if c.ndim == 2:          # one coefficient row per column
            assert c.shape == (Xf.shape[1], self.T)
            check(_lib.lib().wae_spmv_sum_cols(self.handle, zptr(c), c.shape[0], zptr(Xf), zptr(Y), Xf.shape[1], op))
        else:
            check(_lib.lib().wae_spmv_sum(self.handle, zptr(c), zptr(Xf), zptr(Y), Xf.shape[1], op))
        return Y[:, 0].copy() if one else Y

    def spmv_multi(self, coeffs, X):
        c = np.ascontiguousarray(coeffs, dtype=np.complex128)
        Xf = np.asfortranarray(np.asarray(X, dtype=np.complex128).reshape(self.d, self.T))
        Y = np.empty(self.d, dtype=np.complex128)
        check(_lib.lib().wae_spmv_sum_multi(self.handle, zptr(c), zptr(Xf), zptr(Y)))
        return Y

    def spmv_bytes(self, r=1, mask=None):
        m = None if mask is None else (C.c_uint8 * self.T)(*[1 if x else 0 for x in mask])
        return _lib.lib().wae_family_spmv_bytes(self.handle, m, r)

    # -- solver --------------------------------------------------------------------------------------
    def setup_solver(self, coeffs_ref, theta=0.02, max_coarse=128, jacobi_weight=0.8, sweeps=1, restart=30,
                     penalty_ratio=1e8, batch=64, shape_exclude=(), probe_columns=0, snapshots=0, jacobi_weight_post=0.0,
                     jacobi_weight_light=0.0):
        """shape_exclude: indices of terms kept out of the multigrid shape matrix (e.g. the seam parts of a Bloch family);
        probe_columns, snapshots: workspace hints for the contour integrals to come (include/waehip.h opts[8], [9]);
        jacobi_weight_post / _light: opts[10], [11] (0 = the library's defaults, 0.9 and 0.5)"""
        c = np.ascontiguousarray(coeffs_ref, dtype=np.complex128)
        mask = 0
        for k in shape_exclude:
            if k < 52:
                mask |= 1 << int(k)
        opts = np.array([theta, max_coarse, jacobi_weight, sweeps, restart, penalty_ratio, batch, float(mask), float(probe_columns),
                         float(snapshots), float(jacobi_weight_post), float(jacobi_weight_light)], dtype=np.float64)
        check(_lib.lib().wae_solver_setup(self.handle, zptr(c), opts.ctypes.data_as(C.POINTER(C.c_double)), len(opts)))
        self.solver_ready = True
        self.batch = batch

    def solve(self, coeffs, B, op=OP_N, tol=1e-12, maxit=300, strict=False, guess=None, quiet=False):
        c = np.ascontiguousarray(coeffs, dtype=np.complex128)
        B = np.asarray(B, dtype=np.complex128)
        one = B.ndim == 1
        Bf = np.asfortranarray(B.reshape(self.d, -1))
        r = Bf.shape[1]
        ncoef = 1 if c.ndim == 1 else c.shape[0]
        X = np.empty_like(Bf, order="F")
        info = SolveInfo()
        if guess is None:
            code = check(_lib.lib().wae_solve(self.handle, zptr(c), ncoef, zptr(Bf), zptr(X), r, op, tol, maxit, C.byref(info)),
                         warn_ok=not strict)
        else:
            Gf = np.asfortranarray(np.asarray(guess, dtype=np.complex128).reshape(self.d, -1))
            assert Gf.shape == Bf.shape
            code = check(_lib.lib().wae_solve_guess(self.handle, zptr(c), ncoef, zptr(Bf), zptr(Gf), zptr(X), r, op, tol, maxit,
                                                    C.byref(info)), warn_ok=not strict)
        self._report(code, info, "solve", quiet=quiet)
        return X[:, 0].copy() if one else X

    def beyn_moments(self, z, w, coeff_table, V, K=1, tol=1e-10, maxit=300, out_dev=0):
        z = np.ascontiguousarray(z, dtype=np.complex128)
        w = np.ascontiguousarray(w, dtype=np.complex128)
        ct = np.ascontiguousarray(coeff_table, dtype=np.complex128)
        Vf = np.asfortranarray(np.asarray(V, dtype=np.complex128))
        l = Vf.shape[1]
        info = SolveInfo()
        A = None
        aptr = None
        if not out_dev:
            A = np.zeros((self.d, l, 2 * K), dtype=np.complex128, order="F")
            aptr = zptr(A)
        code = check(_lib.lib().wae_beyn_moments(self.handle, len(z), zptr(z), zptr(w), zptr(ct), zptr(Vf), l, K, tol, maxit,
                                                 aptr, int(out_dev), C.byref(info)))
        self._report(code, info, "beyn_moments", fatal=True)
        return A

    def beyn_moments_rb(self, z, w, coeff_table, V, mode, nbasis, slot0=0, Q_dev=0, K=1, tol=1e-10, maxit=300, out_dev=0,
                        accumulate=False, l_total=0, col0=0):
        """wae_beyn_moments_rb: mode 0 solves the points and stores their solutions as snapshots, mode 1 starts every
        system from the Galerkin projection on the snapshot basis (include/waehip.h).  V=None (mode 2 only, with
        l_total = number of probe columns): reuse the probe matrix of the call that started the basis."""
        z = np.ascontiguousarray(z, dtype=np.complex128)
        w = np.ascontiguousarray(w, dtype=np.complex128)
        ct = np.ascontiguousarray(coeff_table, dtype=np.complex128).reshape(len(z), self.T)
        if V is None:                      # mode 2: the probe matrix of the call that started the basis, still on the device
            Vf, l = None, int(l_total)
            l_total = 0
        else:
            Vf = np.asfortranarray(np.asarray(V, dtype=np.complex128))
            l = Vf.shape[1]
        info = SolveInfo()
        A, aptr = None, None
        if not out_dev:
            A = np.zeros((self.d, l_total if l_total > 0 else l, 2 * K), dtype=np.complex128, order="F")
            aptr = zptr(A)
        code = check(_lib.lib().wae_beyn_moments_rb(self.handle, len(z), zptr(z), zptr(w), zptr(ct), None if Vf is None else zptr(Vf), l, K, tol, maxit,
                                                    int(mode), int(nbasis), int(slot0), int(Q_dev), aptr, int(out_dev),
                                                    1 if accumulate else 0, int(l_total), int(col0), C.byref(info)))
        self._report(code, info, "beyn_moments_rb", fatal=True)
        return A

    def rb_export(self):
        """wae_rb_export: (kact, Hk[nk, S, S, l], g[S, l]) of the snapshot basis in the handle"""
        S, l, nk = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(_lib.lib().wae_rb_export(self.handle, C.byref(S), C.byref(l), C.byref(nk), None, None, None))
        kact = np.zeros(max(nk.value, 1), dtype=np.int32)
        Hk = np.zeros((nk.value, S.value, S.value, l.value), dtype=np.complex128)
        g = np.zeros((S.value, l.value), dtype=np.complex128)
        check(_lib.lib().wae_rb_export(self.handle, C.byref(S), C.byref(l), C.byref(nk), kact.ctypes.data_as(C.POINTER(C.c_int32)),
                                       zptr(Hk), zptr(g)))
        return kact[:nk.value], Hk, g

    def rb_import(self, Q_dev, kact, Hk, g):
        """wae_rb_import: install a basis (vectors in Q_dev: S x d x l interleaved, orthonormal per column) for mode 2"""
        kact = np.ascontiguousarray(kact, dtype=np.int32)
        Hk = np.ascontiguousarray(Hk, dtype=np.complex128)
        g = np.ascontiguousarray(g, dtype=np.complex128)
        S, l = g.shape
        assert Hk.shape == (len(kact), S, S, l)
        check(_lib.lib().wae_rb_import(self.handle, S, l, int(Q_dev), len(kact), kact.ctypes.data_as(C.POINTER(C.c_int32)), zptr(Hk), zptr(g)))

    def eig_residuals(self, coeff_table, P=None, P_dev=0):
        """wae_eig_residuals: ||L(ω_j) v_j|| / Σ_k |c_jk| ||A_k v_j|| for every pair; P (host, d x n) or P_dev (device
        pointer of the column-major d x n matrix)."""
        ct = np.ascontiguousarray(coeff_table, dtype=np.complex128)
        n = ct.shape[0]
        out = np.zeros(n, dtype=np.float64)
        Pf = None if P is None else np.asfortranarray(np.asarray(P, dtype=np.complex128))
        check(_lib.lib().wae_eig_residuals(self.handle, n, zptr(ct), None if Pf is None else zptr(Pf), int(P_dev),
                                           out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def arnoldi(self, coeffsA, coeffsM, m, v0, op=OP_N, tol=1e-12, maxit=300, quiet=False):
        cA = np.ascontiguousarray(coeffsA, dtype=np.complex128)
        cM = np.ascontiguousarray(coeffsM, dtype=np.complex128)
        v0 = np.ascontiguousarray(v0, dtype=np.complex128)
        H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
        V = np.zeros((self.d, m + 1), dtype=np.complex128, order="F")
        info = SolveInfo()
        code = check(_lib.lib().wae_arnoldi_shiftinvert(self.handle, zptr(cA), zptr(cM), m, zptr(v0), op, tol, maxit, zptr(H), zptr(V),
                                                        C.byref(info)))
        self._report(code, info, "arnoldi", quiet=quiet)
        return H, V

    def arnoldi_batch(self, coeffsA, coeffsM, m, V0, op=OP_N, tol=1e-12, maxit=300, ritz_tol=0.0, quiet=False):
        """wae_arnoldi_shiftinvert_batch: nsys Arnoldi processes in lock-step.  coeffsA, coeffsM: (nsys, T); V0: (d, nsys).
        Returns H (nsys, m+1, m) and V (nsys, d, m+1)."""
        cA = np.ascontiguousarray(coeffsA, dtype=np.complex128).reshape(-1, self.T)
        nsys = cA.shape[0]
        cM = np.ascontiguousarray(np.broadcast_to(np.asarray(coeffsM, dtype=np.complex128).reshape(-1, self.T), (nsys, self.T)))
        V0 = np.asfortranarray(np.asarray(V0, dtype=np.complex128).reshape(self.d, nsys))
        H = np.zeros((nsys, m, m + 1), dtype=np.complex128)            # each block column-major (m+1) x m
        V = np.zeros((nsys, m + 1, self.d), dtype=np.complex128)        # each block column-major d x (m+1)
        info = SolveInfo()
        code = check(_lib.lib().wae_arnoldi_shiftinvert_batch(self.handle, nsys, zptr(cA), zptr(cM), m, zptr(V0), op, tol, maxit,
                                                              float(ritz_tol), zptr(H), zptr(V), C.byref(info)))
        self._report(code, info, "arnoldi_batch", quiet=quiet)
        return H.transpose(0, 2, 1), V.transpose(0, 2, 1)

    def perturb(self, coeff_table, N, v0, v0adj, norm_mode=1, coeffsY=None, tol=1e-12, maxit=400, quiet=False):
        """wae_perturb: the whole recurrence on the device. coeff_table[(m,n)] = T coefficients of L(m,n)."""
        ct = np.ascontiguousarray(coeff_table, dtype=np.complex128).reshape((N + 1) * (N + 1), self.T)
        v0 = np.ascontiguousarray(v0, dtype=np.complex128)
        va = np.ascontiguousarray(v0adj, dtype=np.complex128)
        lam = np.zeros(N + 1, dtype=np.complex128)
        V = np.zeros((self.d, N + 1), dtype=np.complex128, order="F")
        cy = None if coeffsY is None else np.ascontiguousarray(coeffsY, dtype=np.complex128)
        info = SolveInfo()
        code = check(_lib.lib().wae_perturb(self.handle, zptr(ct), N, zptr(v0), zptr(va), norm_mode, None if cy is None else zptr(cy),
                                            tol, maxit, zptr(lam), zptr(V), C.byref(info)))
        self._report(code, info, f"perturb (order {N})", quiet=quiet)
        return lam, V

    # -- device-resident multivectors ("slots", include/waehip.h): the vectors of the Newton-type solvers stay in HBM between the calls
    NSLOTS = 8

    @staticmethod
    def _icols(cols):
        a = np.ascontiguousarray(cols, dtype=np.int32).reshape(-1)
        return a, a.ctypes.data_as(C.POINTER(C.c_int32))

    def slot_write(self, slot, X=None, ncols_total=None, col0=0):
        """wae_slot_write: X (d x ncols) into columns col0.. of the slot, (re)created with ncols_total columns (default: X's) if its width differs"""
        if X is None:
            check(_lib.lib().wae_slot_write(self.handle, int(slot), int(ncols_total), int(col0), 0, None))
            return
        Xf = np.asarray(X, dtype=np.complex128)
        Xf = Xf.reshape(self.d, -1)
        if not Xf.flags.f_contiguous:
            Xf = np.asfortranarray(Xf)
        n = Xf.shape[1]
        check(_lib.lib().wae_slot_write(self.handle, int(slot), int(n if ncols_total is None else ncols_total), int(col0), n, zptr(Xf)))

    def slot_read(self, slot, col0, ncols):
        X = np.empty((self.d, int(ncols)), dtype=np.complex128, order="F")
        check(_lib.lib().wae_slot_read(self.handle, int(slot), int(col0), int(ncols), zptr(X)))
        return X

    def slot_axpby(self, dst_slot, dst_cols, src_slot, src_cols, alpha=1.0, beta=0.0, conj_src=False):
        """dst[:, dst_cols[i]] = alpha[i] src[:, src_cols[i]] + beta[i] dst[:, dst_cols[i]], one column after the other (conj_src: the
        conjugate of the source column)"""
        dc, dcp = self._icols(dst_cols)
        sc, scp = self._icols(src_cols)
        n = len(dc)
        assert len(sc) == n
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.complex128), (n,)))
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(beta, dtype=np.complex128), (n,)))
        check(_lib.lib().wae_slot_axpby(self.handle, n, int(dst_slot), dcp, int(src_slot), scp, zptr(a), zptr(b), 1 if conj_src else 0))

    def slot_forms(self, coeffs, a_slot, a_cols, b_slot, b_cols, op=OP_N):
        """out[i] = a_i^H op(sum_k coeffs[i, k] A_k) b_i for pairs of slot columns"""
        ac, acp = self._icols(a_cols)
        bc, bcp = self._icols(b_cols)
        n = len(ac)
        assert len(bc) == n
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(coeffs, dtype=np.complex128).reshape(-1, self.T), (n, self.T)))
        out = np.zeros(n, dtype=np.complex128)
        check(_lib.lib().wae_slot_forms(self.handle, n, zptr(c), int(op), int(a_slot), acp, int(b_slot), bcp, zptr(out)))
        return out

    def arnoldi_slots(self, coeffsA, coeffsM, m, v0_slot, v0_cols, op=OP_N, tol=1e-12, maxit=300, ritz_tol=0.0, quiet=False):
        """wae_arnoldi_shiftinvert_slots: as arnoldi_batch with the start vectors in slot columns; returns H (nsys, m+1, m) only --
        the basis stays on the device for ritz_to_slot"""
        cA = np.ascontiguousarray(coeffsA, dtype=np.complex128).reshape(-1, self.T)
        nsys = cA.shape[0]
        cM = np.ascontiguousarray(np.broadcast_to(np.asarray(coeffsM, dtype=np.complex128).reshape(-1, self.T), (nsys, self.T)))
        vc, vcp = self._icols(v0_cols)
        assert len(vc) == nsys
        H = np.zeros((nsys, m, m + 1), dtype=np.complex128)            # each block column-major (m+1) x m
        info = SolveInfo()
        code = check(_lib.lib().wae_arnoldi_shiftinvert_slots(self.handle, nsys, zptr(cA), zptr(cM), m, int(v0_slot), vcp, op, tol, maxit,
                                                              float(ritz_tol), zptr(H), C.byref(info)))
        self._report(code, info, "arnoldi_slots", quiet=quiet)
        return H.transpose(0, 2, 1)

    def ritz_to_slot(self, Y, dst_slot, dst_cols, normalise=True):
        """wae_arnoldi_ritz_to_slot: dst[:, dst_cols[s]] = sum_j Y[s, j] v_j^(s) of the basis of the last arnoldi_slots call"""
        Yc = np.ascontiguousarray(Y, dtype=np.complex128)
        nsys, ny = Yc.shape
        dc, dcp = self._icols(dst_cols)
        assert len(dc) == nsys
        check(_lib.lib().wae_arnoldi_ritz_to_slot(self.handle, nsys, ny, zptr(Yc), int(dst_slot), dcp, 1 if normalise else 0))

    def perturb_slots(self, coeff_table, N, v_slot, v_col, vadj_slot, vadj_col, norm_mode=1, coeffsY=None, tol=1e-12, maxit=400, quiet=False,
                      vectors=False):
        """wae_perturb_slots: wae_perturb on slot columns; vectors=False: the eigenvalue series only (no vector leaves the device)"""
        ct = np.ascontiguousarray(coeff_table, dtype=np.complex128).reshape((N + 1) * (N + 1), self.T)
        lam = np.zeros(N + 1, dtype=np.complex128)
        V = np.zeros((self.d, N + 1), dtype=np.complex128, order="F") if vectors else None
        cy = None if coeffsY is None else np.ascontiguousarray(coeffsY, dtype=np.complex128)
        info = SolveInfo()
        code = check(_lib.lib().wae_perturb_slots(self.handle, zptr(ct), N, int(v_slot), int(v_col), int(vadj_slot), int(vadj_col), norm_mode,
                                                  None if cy is None else zptr(cy), tol, maxit, zptr(lam), None if V is None else zptr(V),
                                                  C.byref(info)))
        self._report(code, info, f"perturb_slots (order {N})", quiet=quiet)
        return lam, V

    def debug_spmv(self, coeffs, X, mode=0, B=None, Y0=None, op=OP_N, jac_w=0.8, cmask=None, level=0, which=0, no_tiles=False):
        """wae_debug_spmv (test hook): one launch of the fused operator product in any of the solver's forms.  Returns Y, or
        (Y, B2) for mode 6.  ``level``/``which`` select a coarse-level operator or a restriction of the multigrid hierarchy."""
        lib = _lib.lib()
        nin, nout = C.c_int64(0), C.c_int64(0)
        check(lib.wae_debug_spmv(self.handle, which, level, 0, None, 0, None, None, None, None, 0, op, 0.0, None, 0, C.byref(nin), C.byref(nout)))
        c = np.ascontiguousarray(coeffs, dtype=np.complex128).reshape(-1, self.T)
        Xf = np.asfortranarray(np.asarray(X, dtype=np.complex128).reshape(nin.value, -1))
        r = Xf.shape[1]
        Y = np.zeros((nout.value, r), dtype=np.complex128, order="F") if Y0 is None else np.asfortranarray(np.array(Y0, dtype=np.complex128))
        Bf = None if B is None else np.asfortranarray(np.asarray(B, dtype=np.complex128).reshape(nout.value, r))
        B2 = np.zeros((nout.value, r), dtype=np.complex128, order="F") if mode == 6 else None
        if mode == 6 and Y0 is not None:
            B2[...] = Y0
        cm = None if cmask is None else (C.c_uint8 * ((r + 7) // 8))(*[1 if x else 0 for x in cmask])
        check(lib.wae_debug_spmv(self.handle, which, level, mode, zptr(c), c.shape[0], zptr(Xf), None if Bf is None else zptr(Bf), zptr(Y),
                                 None if B2 is None else zptr(B2), r, op, float(jac_w), cm, 1 if no_tiles else 0, C.byref(nin), C.byref(nout)))
        return (Y, B2) if mode == 6 else Y

    def level_sizes(self):
        """(n_in, n_out) of every sparse level operator / restriction of the hierarchy (test hook)"""
        out = []
        lib = _lib.lib()
        for which in (0, 1):
            lv = 0
            while True:
                nin, nout = C.c_int64(0), C.c_int64(0)
                if lib.wae_debug_spmv(self.handle, which, lv, 0, None, 0, None, None, None, None, 0, 0, 0.0, None, 0, C.byref(nin), C.byref(nout)) != 0:
                    break
                out.append((which, lv, nin.value, nout.value))
                lv += 1
        return out

    def bench_spmv_level(self, coeffs, which=0, level=1, r=64, reps=20):
        """(ms per launch, algorithmic bytes per launch) of a level operator / restriction of the hierarchy"""
        c = np.ascontiguousarray(coeffs, dtype=np.complex128)
        ms, nbytes = C.c_double(0), C.c_int64(0)
        check(_lib.lib().wae_bench_spmv_level(self.handle, zptr(c), which, level, r, reps, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def bench_spmv(self, coeffs, r=1, reps=20):
        c = np.ascontiguousarray(coeffs, dtype=np.complex128)
        ms = C.c_double(0)
        check(_lib.lib().wae_bench_spmv(self.handle, zptr(c), r, reps, C.byref(ms)))
        return ms.value


class Operator:
    """What ``L(z)`` returns here: sum_k c_k A_k on the device (no matrix is assembled)."""

    def __init__(self, fam, coeffs, op=OP_N, owner=None):
        self.fam = fam
        self.coeffs = np.asarray(coeffs, dtype=np.complex128)
        self.op = op
        self.owner = owner
        self.shape = (fam.d, fam.d)

    @property
    def H(self):
        """Julia ``A'`` (Householder.jl:101, iterative_solvers.jl:398,572)"""
        return Operator(self.fam, self.coeffs, OP_N if self.op == OP_C else OP_C, self.owner)

    def __matmul__(self, x):
        return self.fam.spmv(self.coeffs, x, self.op)

    dot = __matmul__

    def solve(self, b, tol=None, maxit=None, guess=None, quiet=False):
        """Julia ``A \\ b`` (beyn.jl:65; iterative_solvers.jl:307,397-398)"""
        own = self.owner
        if own is not None:
            own.ensure_solver()
        tol = tol if tol is not None else (own.solver_tol if own is not None else 1e-12)
        maxit = maxit if maxit is not None else (own.solver_maxit if own is not None else 300)
        return self.fam.solve(self.coeffs, b, self.op, tol, maxit, guess=guess, quiet=quiet)

    def __neg__(self):
        return Operator(self.fam, -self.coeffs, self.op, self.owner)

    def toarray(self):
        """dense copy (small problems / tests only)"""
        return self @ np.eye(self.shape[0], dtype=np.complex128)


class LinearOperatorFamily:
    """LinOpFam.jl:131-186 (type + constructors), :305-346 (push!), :482-529 (functor)."""

    def __init__(self, params=("λ",), values=None, device=0):
        if isinstance(params, str):                     # LinearOperatorFamily(fname)  LinOpFam.jl:196-225
            from .save import load_family
            self.__dict__.update(load_family(params, device=device).__dict__)
            return
        params = list(params)
        if values is None:
            values = [NaN for _ in params]
        self.terms = []
        self.eigval = params[0]
        self.auxval = params[-1] if len(params) > 1 else ""
        self.active = [self.eigval]
        self.params = {p: complex(v) for p, v in zip(params, values)}
        self.mode = "all"
        self.device_id = device
        self._fam = None
        # inner-solver controls (no reference counterpart: the reference solves directly with UMFPACK)
        self.solver_tol = 1e-12
        self.solver_maxit = 400
        self.solver_opts = {}
        self.solver_ref = None          # reference value of the eigenvalue parameter for the multigrid set-up
        self.solver_ref_coeffs = None   # or: explicit reference coefficients (one per term) for the set-up
        self.rb_snapshots = None        # Beyn: snapshot points for projected initial guesses (None = automatic, 0 = off)
        # when the library may treat a term as symmetric for `A'` products (include/waehip.h wae_family_create_opts opts[0]): 0 = only
        # if it is bitwise symmetric; producers of finite-element terms (helmholtz_family) set 1e-14
        self.symmetry_tol = 0.0

    # -- term management -------------------------------------------------------------------------------
    def push(self, T):
        """push!  LinOpFam.jl:305-346"""
        self._drop_device()
        for idx, term in enumerate(self.terms):
            if term.func == T.func and term.params == T.params:
                coeff = term.coeff + T.coeff
                nrm = abs(coeff).sum()
                if nrm == 0:
                    del self.terms[idx]
                else:
                    self.terms[idx] = Term(coeff, term.func, term.params, term.symbol, term.operator)
                return self
        for pars in T.params:
            for par in pars:
                if par not in self.params:
                    self.params[par] = NaN
        self.terms.append(T)
        return self

    def __add__(self, T):
        """LinOpFam.jl:353-357"""
        L = self.copy()
        L.push(T)
        return L

    def __sub__(self, T):
        """LinOpFam.jl:364-368"""
        L = self.copy()
        L.push(Term(-T.coeff, T.func, T.params, T.symbol, T.operator))
        return L

    def copy(self):
        fam, self._fam = self._fam, None
        L = copy.deepcopy(self)
        self._fam = fam
        return L

    def size(self):
        """LinOpFam.jl:385-393"""
        return self.terms[0].coeff.shape[0] if self.terms else 0

    # -- device handle ---------------------------------------------------------------------------------
    def _drop_device(self):
        if self._fam is not None:
            self._fam.close()
            self._fam = None

    def device(self):
        if self._fam is None:
            self._fam = DeviceFamily([t.coeff for t in self.terms], self.device_id, symmetry_tol=getattr(self, "symmetry_tol", 0.0))
        return self._fam

    def ensure_solver(self):
        fam = self.device()
        if not fam.solver_ready and self.solver_ref_coeffs is not None:
            fam.setup_solver(np.asarray(self.solver_ref_coeffs, dtype=np.complex128), **self.solver_opts)
        if not fam.solver_ready:
            zref = self.solver_ref
            if zref is None:
                zref = self.params.get(self.eigval, 0j)
                if not np.isfinite(zref):
                    zref = 0j
            saved = dict(self.params), list(self.active), self.mode
            self.active, self.mode = [self.eigval], "all"
            try:
                c = self.coefficients(zref)
            finally:
                self.params, self.active, self.mode = saved
            fam.setup_solver(c, **self.solver_opts)
        return fam

    # -- functor ---------------------------------------------------------------------------------------
    def coefficients(self, *args, oplist=(), in_or_ex=False):
        """The scalar half of the reference functor (LinOpFam.jl:482-526): c_k for every term, 0 where the
        reference skips the term (oplist / "__aux__" outside householder mode / derivative of a constant)."""
        nact = len(self.active)
        if self.mode == "all":
            for var, val in zip(self.active, args):
                self.params[var] = complex(val)
        if self.mode == "all" and len(args) == nact:
            derivs = [0] * nact
        else:
            derivs = [int(a) for a in args[len(args) - nact:]]
        deriv_dict = dict(zip(self.active, derivs))
        out = np.zeros(len(self.terms), dtype=np.complex128)
        for k, term in enumerate(self.terms):
            if ((not in_or_ex and term.operator in oplist) or (in_or_ex and term.operator not in oplist)
                    or (self.mode != "householder" and term.operator == "__aux__")):
                continue
            if any(dd > 0 and var not in term.varlist for var, dd in zip(self.active, derivs)):
                continue
            out[k] = term.scalar({var: (self.params[var], deriv_dict.get(var, 0)) for var in term.varlist})
        if self.mode in ("compact", "householder"):
            div = 1.0
            for a in args[len(args) - nact:]:
                div *= float(factorial(int(a)))
            out /= div
        return out

    def term_operator(self, k, scale=1.0):
        """scale * A_k as an Operator (e.g. M = -L.terms[end].coeff, Householder.jl:92)"""
        c = np.zeros(len(self.terms), dtype=np.complex128)
        c[k] = scale
        return Operator(self.device(), c, OP_N, self)

    def __call__(self, *args, oplist=(), in_or_ex=False):
        return Operator(self.device(), self.coefficients(*args, oplist=oplist, in_or_ex=in_or_ex), OP_N, self)


def polyval(p, z):
    """LinOpFam.jl:723-730"""
    f = p[-1]
    for i in range(len(p) - 2, -1, -1):
        f = f * z + p[i]
    return f


def pade(w, L, M):
    """LinOpFam.jl:622-642"""
    w = np.asarray(w, dtype=complex)
    b = np.array([1.0 + 0j])
    if M > 0:
        A = np.zeros((M, M), dtype=complex)
        for i in range(1, M + 1):
            for j in range(1, M + 1):
                if L + i - j >= 0:
                    A[i - 1, j - 1] = w[L + i - j]
        b = np.concatenate([[1.0 + 0j], np.linalg.solve(A, -w[L + 1:L + M + 1])])
    a = np.zeros(L + 1, dtype=complex)
    for l in range(L + 1):
        for m in range(min(l, M) + 1):
            a[l] += w[l - m] * b[m]
    return a, b


def pade_(sol, param, L, M, vector=False):
    """pade!(sol, param, L, M; vector)  (LinOpFam.jl:646-680): Padé coefficients of the eigenvalue series and,
    component-wise, of the eigenvector series."""
    key, tkey = f"{param}/[{L}/{M}]", f"{param}/Taylor"
    sol.eigval_pert[key] = pade(sol.eigval_pert[tkey], L, M)
    if vector:
        V = np.array(sol.v_pert[tkey][:L + M + 1])    # (L+M+1, d)
        d = V.shape[1]
        A = np.zeros((L + 1, d), dtype=complex)
        B = np.zeros((M + 1, d), dtype=complex)
        for i in range(d):
            A[:, i], B[:, i] = pade(V[:, i], L, M)
        sol.v_pert[key] = (A, B)


def estimate_pol(w):
    """estimate_pol (LinOpFam.jl:736-747): pole position / order estimates from consecutive Taylor coefficients."""
    w = np.asarray(w, dtype=complex)
    N = len(w)
    de = np.zeros(N - 2, dtype=complex)
    k = np.zeros(N - 2, dtype=complex)
    for j in range(1, N - 1):
        i = j                                           # 1-based j-1 in the reference
        denom = (i + 1) * w[j + 1] * w[j - 1] - i * w[j] ** 2
        de[i - 1] = w[j] * w[j - 1] / denom
        k[i - 1] = (i * i - 1) * w[j + 1] * w[j - 1] - (i * w[j]) ** 2
    return de, k


def conv_radius(a):
    """LinOpFam.jl:754-761"""
    a = np.asarray(a)
    return np.abs(a[:-1] / a[1:])


def poly_roots(p):
    """Householder.jl:195-203"""
    p = np.asarray(p, dtype=complex)
    N = len(p) - 1
    Cm = np.zeros((N, N), dtype=complex)
    for i in range(1, N):
        Cm[i, i - 1] = 1
    Cm[:, N - 1] = -p[:N] / p[N]
    return np.linalg.eigvals(Cm)
