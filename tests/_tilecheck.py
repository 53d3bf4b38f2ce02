"""Shared checker of the fused operator product against scipy (used by tests/test_gpu_tile_parity.py and its child-process
worker tests/tile_worker.py).  The oracle here is scipy's CSR product of the caller's own term matrices -- the reference's
`L(z)*x` (LinOpFam.jl:482-529: the sum of the scaled term matrices applied to a vector) -- evaluated term by term, so that one
set of four sparse products serves every coefficient row.

Tolerances: 1e-13 relative per column in the max-norm AND element-wise against the rounding bound of
a row, |y_i - yhat_i| <= 1e-13 * (sum_k |c_k| |A_k| |x|)_i: the max-norm of a column is set by the admittance rows (entries of
1e15), which says nothing about the interior rows; the element-wise bound does."""
import numpy as np

TOL = 1e-13
NAMES = ("M", "K", "C", "Q")


def annulus_coeffs(z, n=1.0, tau=2e-4, Y=1e15):
    """coefficients of (M, K, C, Q, aux) in L(w) = w^2 M + K + w Y C + n exp(-i w tau) Q  (Helmholtz.jl:422-487; aux = 0)"""
    z = np.atleast_1d(np.asarray(z, dtype=complex))
    return np.stack([z * z, np.ones_like(z), z * Y, n * np.exp(-1j * z * tau), np.zeros_like(z)], axis=1)


class TermProducts:
    """A_k X and |A_k| |X| for the four annulus terms, computed once per X (and per orientation)"""

    def __init__(self, terms, X, op="N"):
        self.op = op
        mats = [terms[k].tocsr() for k in NAMES]
        if op == "C":
            mats = [A.conj().T.tocsr() for A in mats]
        self.AX = [A @ X for A in mats]
        aX = np.abs(X)
        self.absAX = [abs(A) @ aX for A in mats]
        self.diag = [A.diagonal() for A in mats]

    def apply(self, ct):
        """ct: (1, 5) or (r, 5) term coefficients -> (Y, bound, diag) with column j using row min(j, len-1)"""
        r = self.AX[0].shape[1]
        ct = np.asarray(ct)[:, :4]
        if self.op == "C":
            ct = ct.conj()
        cj = ct if ct.shape[0] == r else np.repeat(ct[:1], r, axis=0)
        Y = sum(self.AX[k] * cj[None, :, k] for k in range(4))
        bound = sum(self.absAX[k] * np.abs(cj[None, :, k]) for k in range(4))
        dg = sum(self.diag[k][:, None] * cj[None, :, k] for k in range(4))
        return Y, bound, dg


class FamilyProducts:
    """The same for ANY family: A_k X and |A_k| |X| for every term of ``L`` (host matrices ``t.coeff``), coefficient rows as
    ``L.coefficients(z)`` gives them (one per term, 0 where the functor skips a term) -- e.g. the 11 terms of a Bloch unit cell
    (base / seam parts times exp(+-i b 2 pi / DOS), src/Helmholtz.jl:508-513)."""

    def __init__(self, L, X, op="N"):
        self.op = op
        mats = [t.coeff.tocsr() for t in L.terms]
        if op == "C":
            mats = [A.conj().T.tocsr() for A in mats]
        self.AX = [A @ X for A in mats]
        aX = np.abs(X)
        self.absAX = [abs(A) @ aX for A in mats]
        self.diag = [A.diagonal() for A in mats]

    def apply(self, ct):
        r = self.AX[0].shape[1]
        ct = np.asarray(ct)
        if self.op == "C":
            ct = ct.conj()
        cj = ct if ct.shape[0] == r else np.repeat(ct[:1], r, axis=0)
        nz = [k for k in range(ct.shape[1]) if np.any(cj[:, k] != 0)]
        Y = sum(self.AX[k] * cj[None, :, k] for k in nz)
        bound = sum(self.absAX[k] * np.abs(cj[None, :, k]) for k in nz)
        dg = sum(self.diag[k][:, None] * cj[None, :, k] for k in nz)
        return Y, bound, dg


def scipy_operator_product(L, coeffs, V):
    """sum_k c_k A_k V with scipy on the host matrices of the family: the product the residual checks of the full-size tests form
    WITHOUT the kernels under test"""
    return sum(c * (t.coeff @ V) for c, t in zip(coeffs, L.terms) if c != 0)


def assert_close(got, want, bound, what, cols=None, tol=TOL):
    cols = range(want.shape[1]) if cols is None else cols
    for j in cols:
        err = np.abs(got[:, j] - want[:, j])
        cmax = np.max(np.abs(want[:, j]))
        assert np.max(err) <= tol * cmax, f"{what}: column {j} max-norm error {np.max(err) / cmax:.2e}"
        ew = np.max(err / np.maximum(bound[:, j], 1e-300))
        assert ew <= tol, f"{what}: column {j} element-wise error {ew:.2e} of the row's rounding bound"


_RANDOM = {}


def _random_block(rng, d, r):
    """(B0, Y0) of a shape, drawn once per process: at 1M unknowns and 64 columns three normal draws are 2.5 s of every call below"""
    if (d, r) not in _RANDOM:
        _RANDOM.clear()                                              # (one shape at a time: a block is 1 GB at the benchmark size)
        _RANDOM[(d, r)] = (rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r)), rng.standard_normal((d, r)) + 7j)
    return _RANDOM[(d, r)]


def check_modes(fam, tp, ct, X, rng, what, modes=(0, 1, 2, 3, 4, 5, 6), cmask=None, op=0, jac_w=0.8):
    """every fused form of the operator product (include/waehip.h wae_debug_spmv) against the scipy term products `tp`"""
    want, bound, dg = tp.apply(ct)
    d, r = X.shape
    B0, Y0 = _random_block(rng, d, r)                                # Y0: what a masked chunk must keep
    B = B0 * np.maximum(np.abs(dg), 1.0)                             # right-hand sides of the size of the rows they meet
    act = np.ones(r, dtype=bool)
    if cmask is not None:
        act = np.repeat(np.asarray(cmask, dtype=bool), 8)[:r]
    on, off = np.nonzero(act)[0], np.nonzero(~act)[0]
    adg = np.abs(dg)
    for mode in modes:
        out = fam.debug_spmv(ct, X, mode=mode, B=None if mode in (0, 4, 6) else B, Y0=Y0, op=op, jac_w=jac_w, cmask=cmask)
        Y, B2 = out if mode == 6 else (out, None)
        if mode in (0, 6):
            ref, bnd = want, bound
        elif mode == 1:
            ref, bnd = B - want, np.abs(B) + bound
        elif mode == 2:
            ref, bnd = X + jac_w / dg * (B - want), np.abs(X) + jac_w / adg * (np.abs(B) + bound)
        elif mode == 3:
            ref, bnd = B + want, np.abs(B) + bound
        elif mode == 4:
            ref, bnd = want / dg, bound / adg
        else:
            ref, bnd = (B - want) / dg, (np.abs(B) + bound) / adg
        assert_close(Y, ref, bnd, f"{what} mode {mode}", cols=on)
        if mode == 6:
            assert_close(B2, jac_w / dg * want, jac_w / adg * bound, f"{what} mode 6 (second output)", cols=on)
        if len(off):
            assert np.array_equal(Y[:, off], Y0[:, off]), f"{what} mode {mode}: a masked chunk was written"
            if mode == 6:
                assert np.array_equal(B2[:, off], Y0[:, off]), f"{what} mode 6: second output of a masked chunk was written"
