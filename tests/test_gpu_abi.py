"""The drop-in boundary exercised the way the Julia host will use it: term matrices handed
over as Julia's SparseMatrixCSC arrays -- column-compressed, 1-based, UInt32 (what Helmholtz.discretize produces,
src/Helmholtz.jl:407-408,515) or Int64 (what L(z) itself carries) -- through ctypes, and through a plain C caller
compiled with gcc against include/waehip.h.  Checked against scipy (the oracle's arithmetic) on the same inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import fixtures as F
from oracle import solvers as OS
from wae_amd import _lib
from wae_amd._lib import SolveInfo, check, zptr

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _julia_csc(A, index_dtype, shuffle=False, duplicates=False):
    """(colptr, rowval, nzval) of SparseMatrixCSC(A) with 1-based indices of the given type.  shuffle: entries of every
    column in random order; duplicates: every third entry split into two entries that sum to it (both are legal inputs of
    `sparse(I,J,V)`-style producers and must be merged by the library)."""
    A = sp.csc_matrix(A, dtype=np.complex128)
    A.sum_duplicates()
    A.sort_indices()
    ptr, idx, val = [0], [], []
    for j in range(A.shape[1]):
        rows = A.indices[A.indptr[j]:A.indptr[j + 1]].tolist()
        vals = A.data[A.indptr[j]:A.indptr[j + 1]].tolist()
        if duplicates:
            extra_r, extra_v = [], []
            for k in range(0, len(rows), 3):
                half = vals[k] * 0.25
                vals[k] = vals[k] - half
                extra_r.append(rows[k]); extra_v.append(half)
            rows += extra_r; vals += extra_v
        if shuffle and len(rows) > 1:
            p = RNG.permutation(len(rows))
            rows = [rows[i] for i in p]; vals = [vals[i] for i in p]
        idx += rows; val += vals
        ptr.append(len(idx))
    return ((np.asarray(ptr, dtype=np.int64) + 1).astype(index_dtype), (np.asarray(idx, dtype=np.int64) + 1).astype(index_dtype),
            np.ascontiguousarray(val, dtype=np.complex128))


class RawFamily:
    """wae_family_create through ctypes with caller-chosen index width / base / orientation"""

    def __init__(self, arrays, d, index_bytes, base, orientation):
        self.keep = arrays
        T = len(arrays)
        ptrs = (C.c_void_p * T)(*[a[0].ctypes.data for a in arrays])
        idxs = (C.c_void_p * T)(*[a[1].ctypes.data for a in arrays])
        vals = (C.c_void_p * T)(*[a[2].ctypes.data for a in arrays])
        self.h = C.c_void_p()
        self.d, self.T = d, T
        check(_lib.lib().wae_family_create(C.byref(self.h), d, T, index_bytes, base, orientation, ptrs, idxs, vals, 0))

    def spmv(self, c, X, op):
        c = np.ascontiguousarray(c, dtype=np.complex128)
        Xf = np.asfortranarray(X, dtype=np.complex128)
        Y = np.empty_like(Xf, order="F")
        check(_lib.lib().wae_spmv_sum(self.h, zptr(c), zptr(Xf), zptr(Y), Xf.shape[1], op))
        return Y

    def setup(self, c):
        c = np.ascontiguousarray(c, dtype=np.complex128)
        check(_lib.lib().wae_solver_setup(self.h, zptr(c), None, 0))

    def solve(self, c, B, op, tol=1e-12):
        c = np.ascontiguousarray(c, dtype=np.complex128)
        Bf = np.asfortranarray(B, dtype=np.complex128)
        X = np.empty_like(Bf, order="F")
        info = SolveInfo()
        code = check(_lib.lib().wae_solve(self.h, zptr(c), 1, zptr(Bf), zptr(X), Bf.shape[1], op, tol, 300, C.byref(info)))
        return X, code, info.as_dict()

    def close(self):
        if self.h:
            _lib.lib().wae_family_destroy(self.h)
            self.h = None


@pytest.mark.parametrize("index_dtype,messy", [(np.uint32, False), (np.int64, False), (np.uint32, True), (np.int64, True)])
def test_julia_shaped_csc_one_based_inputs(index_dtype, messy):
    """M, K, C (symmetric) and the non-symmetric flame matrix Q of the Rijke fixture as 1-based CSC with UInt32 / Int64
    indices, optionally with unsorted and duplicate entries: SpMV for op N/T/C equals scipy on every term and on the sum,
    and a solve (N and C) equals the direct solution."""
    t = F.rijke_terms()
    d = t["M"].shape[0]
    mats = [t["M"], t["K"], t["C"], t["Q"], -t["M"]]
    arrays = [_julia_csc(A, index_dtype, shuffle=messy, duplicates=messy) for A in mats]
    fam = RawFamily(arrays, d, np.dtype(index_dtype).itemsize, 1, _lib.CSC)
    try:
        dd, TT, nnz = C.c_int64(0), C.c_int32(0), C.c_int64(0)
        check(_lib.lib().wae_family_info(fam.h, C.byref(dd), C.byref(TT), C.byref(nnz)))
        assert dd.value == d and TT.value == 5 and nnz.value == sum(sp.csc_matrix(A).nnz for A in mats)   # duplicates merged
        X = RNG.standard_normal((d, 3)) + 1j * RNG.standard_normal((d, 3))
        z = 2 * np.pi * (300 + 20j)
        coef = np.array([z * z, 1.0, z * 1e15, 0.7 * np.exp(-1j * z * 1e-3), 0.0])
        A = sum(ck * sp.csr_matrix(Ak, dtype=complex) for ck, Ak in zip(coef, mats)).tocsr()
        for k in range(4):                                   # every term on its own (Q: the transposed orientation matters)
            e = np.zeros(5, dtype=complex); e[k] = 1.0
            Ak = sp.csr_matrix(mats[k], dtype=complex)
            assert relerr(fam.spmv(e, X, _lib.OP_N), Ak @ X) < 1e-13
            assert relerr(fam.spmv(e, X, _lib.OP_T), Ak.T @ X) < 1e-13
            assert relerr(fam.spmv(e, X, _lib.OP_C), Ak.conj().T @ X) < 1e-13
        assert relerr(fam.spmv(coef, X, _lib.OP_N), A @ X) < 1e-13
        assert relerr(fam.spmv(coef, X, _lib.OP_T), A.T @ X) < 1e-13
        assert relerr(fam.spmv(coef, X, _lib.OP_C), A.conj().T @ X) < 1e-13
        zr = 2 * np.pi * 400.0
        fam.setup(np.array([zr * zr, 1.0, zr * 1e15, 0.7 * np.exp(-1j * zr * 1e-3), 0.0]))
        lu = spla.splu(A.tocsc())
        Xs, code, info = fam.solve(coef, X, _lib.OP_N)
        assert code == 0 and info["n_unconverged"] == 0 and relerr(Xs, lu.solve(X)) < 1e-8
        luh = spla.splu(A.conj().T.tocsc())
        Xh, code, info = fam.solve(coef, X, _lib.OP_C)
        assert code == 0 and relerr(Xh, luh.solve(X)) < 1e-8
    finally:
        fam.close()


def test_zero_based_csr_and_one_based_csc_handles_agree():
    """the layouts the tests / bench use (CSR, 0-based, Int32) and the one Julia hands over give the same operator"""
    t = F.rijke_terms()
    d = t["M"].shape[0]
    mats = [t["M"], t["K"], t["C"], t["Q"]]
    csr = []
    for A in mats:
        A = sp.csr_matrix(A, dtype=np.complex128); A.sort_indices()
        csr.append((A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)))
    f0 = RawFamily(csr, d, 4, 0, _lib.CSR)
    f1 = RawFamily([_julia_csc(A, np.uint32) for A in mats], d, 4, 1, _lib.CSC)
    f2 = RawFamily([_julia_csc(sp.csc_matrix(A).T, np.int64) for A in mats], d, 8, 1, _lib.CSR)    # CSR of A == CSC of A^T
    try:
        X = RNG.standard_normal((d, 9)) + 1j * RNG.standard_normal((d, 9))
        c = RNG.standard_normal(4) + 1j * RNG.standard_normal(4)
        for op in (_lib.OP_N, _lib.OP_T, _lib.OP_C):
            Y0 = f0.spmv(c, X, op)
            assert np.array_equal(Y0, f1.spmv(c, X, op)) and np.array_equal(Y0, f2.spmv(c, X, op))   # same kernels, same data: bitwise
    finally:
        f0.close(); f1.close(); f2.close()


def test_bad_inputs_return_codes_not_crashes():
    t = F.rijke_terms()
    d = t["M"].shape[0]
    ptr, idx, val = _julia_csc(t["M"], np.uint32)
    bad = idx.copy(); bad[5] = d + 7                          # row index out of range
    for arrays, base in (([(ptr, bad, val)], 1), ([(ptr, idx, val)], 0)):     # (1-based arrays declared 0-based: pointer check fails)
        with pytest.raises(_lib.WaeError) as e:
            RawFamily(arrays, d, 4, base, _lib.CSC)
        assert e.value.code == _lib.WAE_ERR_INVALID
    with pytest.raises(_lib.WaeError):
        RawFamily([(ptr, idx, val)], d, 2, 1, _lib.CSC)       # index width


def test_transposed_products_are_exact_unless_the_caller_allows_a_symmetry_tolerance():
    """`A'` is `A'` (VERDICT r03 weak 3 / ADVICE): wae_family_create applies a term un-transposed for op = T / C only if it is bitwise
    symmetric; wae_family_create_opts(opts[0] = t) accepts mirror entries that agree to t of the rows' OFF-DIAGONAL scale -- so one huge
    diagonal entry (a penalty / Dirichlet row) cannot make a genuinely non-symmetric term pass."""
    from wae_amd.nlevp.linopfam import DeviceFamily
    n = 600
    rng = np.random.default_rng(11)
    S = sp.random(n, n, density=0.02, random_state=3, format="csr")
    S = (S + S.T + sp.identity(n) * 4.0).tocsr()                       # exactly symmetric, O(1) entries
    E = sp.triu(S, k=1).tocsr()
    E.data = rng.standard_normal(E.nnz)
    X = rng.standard_normal((n, 8)) + 1j * rng.standard_normal((n, 8))
    c = np.array([1.0 + 0.5j])

    def err(Y, A):
        R = c[0] * (A.T @ X)
        return np.max(np.abs(Y - R)) / np.max(np.abs(R))

    A10 = (S + 1e-10 * (E - E.T)).tocsr()                               # symmetric to 1e-10 of the row scale
    f = DeviceFamily([A10])                                             # default: exact test -> exact transposed product
    assert err(f.spmv(c, X, op=_lib.OP_T), A10) < 1e-14
    f.close()
    f = DeviceFamily([A10], symmetry_tol=1e-8)                          # allowed: applied as stored, off by the asymmetry and no more
    Y = f.spmv(c, X, op=_lib.OP_T)
    assert 1e-12 < err(Y, A10) < 1e-8 and np.max(np.abs(Y - c[0] * (A10 @ X))) < 1e-13 * np.max(np.abs(Y))
    f.close()
    f = DeviceFamily([A10], symmetry_tol=1e-13)                         # a tolerance the asymmetry exceeds: exact again
    assert err(f.spmv(c, X, op=_lib.OP_T), A10) < 1e-14
    f.close()
    B = (S + 0.5 * (E - E.T)).tolil()                                   # grossly non-symmetric ...
    B[7, 7] = 1e15                                                      # ... with one penalty entry that dwarfs everything
    B = B.tocsr()
    f = DeviceFamily([B], symmetry_tol=1e-8)
    Xs = X.copy(); Xs[7, :] = 0.0                                       # (keep the 1e15 out of the comparison's scale)
    R = c[0] * (B.T @ Xs)
    assert np.max(np.abs(f.spmv(c, Xs, op=_lib.OP_T) - R)) < 1e-13 * np.max(np.abs(R))
    f.close()
    with pytest.raises(_lib.WaeError):
        DeviceFamily([S], symmetry_tol=1e-3)                            # not a rounding tolerance any more: refused


def test_zero_columns_are_a_no_op_like_the_reference():
    """L(z) * zeros(d, 0), L(z) \\ zeros(d, 0) and the residual test of no eigenpairs return WAE_OK and touch nothing (the round-3
    bench ended on "bad argument" from an empty batch of start values)."""
    from wae_amd.helmholtz.family import helmholtz_family
    from wae_amd.nlevp import householder_many
    L = helmholtz_family(F.rijke_terms(), n=0.5)
    L.solver_ref = 2 * np.pi * 400
    fam = L.ensure_solver()
    d = L.size()
    c = L.coefficients(2 * np.pi * (300 + 10j))
    E = np.zeros((d, 0), dtype=np.complex128)
    assert fam.spmv(c, E).shape == (d, 0)
    assert fam.spmv(np.zeros((0, len(L.terms)), dtype=np.complex128), E).shape == (d, 0)
    assert fam.solve(c, E).shape == (d, 0) and fam.last_info["n_unconverged"] == 0
    assert len(fam.eig_residuals(np.zeros((0, len(L.terms)), dtype=np.complex128), P=E)) == 0
    assert householder_many(L, []) == []
    # negative widths and missing buffers are still refused
    with pytest.raises(_lib.WaeError):
        check(_lib.lib().wae_spmv_sum(fam.handle, zptr(np.asarray(c)), None, None, -1, 0))
    with pytest.raises(_lib.WaeError):
        check(_lib.lib().wae_spmv_sum(fam.handle, zptr(np.asarray(c)), None, None, 2, 0))


def test_plain_c_caller_runs_the_beyn_moment_sequence(tmp_path):
    """tests/abi/abi_caller.c (gcc, C99) does create -> setup -> wae_beyn_moments -> destroy on the Rijke fixture handed over
    as UInt32 1-based CSC (config C1 shape: quadratic problem + flame, l = 4, N = 8 per edge); the moments it writes equal
    the oracle's direct-solver moments to the Beyn parity bar (1e-8)."""
    from test_abi import _build_c_caller
    exe = _build_c_caller(str(tmp_path / "abi_caller"), link=True)
    t = F.rijke_terms()
    d = t["M"].shape[0]
    n, tau, Yv = 0.5, 1e-3, 1e15
    mats = [t["M"], t["K"], t["C"], t["Q"], -t["M"]]
    Lo = F.rijke_family(n=n, tau=tau)
    Gam = np.array([150 + 50j, 150 - 50j, 1000 - 50j, 1000 + 50j]) * 2 * np.pi
    l, K, N = 4, 1, 8
    from wae_amd.nlevp.beyn import gauss_points
    zs, ws = gauss_points(Gam, N)
    coef = lambda z: np.array([z * z, 1.0, z * Yv, n * np.exp(-1j * z * tau), 0.0])      # noqa: E731
    ct = np.array([coef(z) for z in zs])
    V = OS.initial_V(d, l)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([d, len(mats), 4, 1, len(zs), l, K, 300], dtype=np.int64).tofile(f)
        np.array([1e-12]).tofile(f)
        for A in mats:
            ptr, idx, val = _julia_csc(A, np.uint32)
            np.array([len(idx)], dtype=np.int64).tofile(f)
            ptr.tofile(f); idx.tofile(f); val.tofile(f)
        coef(2 * np.pi * 400.0).astype(np.complex128).tofile(f)
        zs.astype(np.complex128).tofile(f); ws.astype(np.complex128).tofile(f)
        np.ascontiguousarray(ct, dtype=np.complex128).tofile(f)
        np.asfortranarray(V).ravel(order="F").tofile(f)
    res = subprocess.run([exe, "run", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    rc = raw[:32].view(np.int64)
    ii = raw[32:64].view(np.int64)
    assert list(rc) == [0, 0, 0, 0] and ii[2] == 0 and ii[3] >= 1            # all calls WAE_OK, no unconverged column
    A = raw[80:].view(np.complex128).reshape((d, l, 2 * K), order="F")
    Ao = OS.compute_moment_matrices(Lo, Gam, V, K=K, N=N)
    assert relerr(A, Ao) < 1e-8
