"""GPU tests of the device-resident multivectors ("slots", include/waehip.h wae_slot_* / wae_arnoldi_shiftinvert_slots /
wae_arnoldi_ritz_to_slot / wae_perturb_slots) behind the lock-step Newton-type refinement: every entry point against the host-memory
call it stands in for (wae_arnoldi_shiftinvert_batch, wae_perturb, wae_spmv_sum) or against numpy / scipy on the same inputs, and
`householder_many` resident against `householder_many_host` -- the same iteration (Householder.jl:70-192) with the vectors passing
through host memory between the calls.  Tolerances: the two forms run the same kernels on the same numbers, so Hessenberg matrices and
eigenvalue series agree to rounding (1e-12 relative); eigenvalues of the Newton iteration to 1e-10 relative."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(11)


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


@pytest.fixture(scope="module")
def annulus():
    from wae_amd.helmholtz.family import annulus_family
    L, pb = annulus_family("small", tau=2e-4)
    L.solver_tol = 1e-12
    L.solver_ref = 2 * np.pi * 500.0
    L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    fam = L.ensure_solver()
    T = pb["terms"]
    mats = [sp.csr_matrix(T[k]) for k in T] if isinstance(T, dict) else [sp.csr_matrix(t) for t in T]
    yield L, pb, fam, mats
    L._drop_device()


def test_slot_write_read_axpby(annulus):
    L, pb, fam, mats = annulus
    d = pb["d"]
    X = RNG.standard_normal((d, 5)) + 1j * RNG.standard_normal((d, 5))
    fam.slot_write(0, X)
    assert np.array_equal(fam.slot_read(0, 0, 5), X)                       # the row renumbering is undone exactly
    assert np.array_equal(fam.slot_read(0, 2, 2), X[:, 2:4])
    Y = RNG.standard_normal((d, 2)) + 1j * RNG.standard_normal((d, 2))
    fam.slot_write(0, Y, ncols_total=5, col0=3)                            # same width: the other columns stay
    ref = X.copy()
    ref[:, 3:5] = Y
    assert np.array_equal(fam.slot_read(0, 0, 5), ref)
    fam.slot_write(1, None, ncols_total=3)                                 # created empty: zeros
    assert not fam.slot_read(1, 0, 3).any()
    fam.slot_write(0, Y, ncols_total=4, col0=1)                            # another width: recreated, zero but for the written columns
    got = fam.slot_read(0, 0, 4)
    assert np.array_equal(got[:, 1:3], Y) and not got[:, 0].any() and not got[:, 3].any()
    # dst = alpha src + beta dst, per column; in place (a scaling); across slots
    fam.slot_write(0, X)
    fam.slot_write(1, X[:, ::-1].copy())
    al = np.array([0.5 - 1j, 2.0, -1j])
    be = np.array([1.0, 0.0, 0.25 + 0.5j])
    fam.slot_axpby(1, [0, 2, 4], 0, [1, 1, 3], alpha=al, beta=be)
    ref = X[:, ::-1].copy()
    for a, b, dc, sc in zip(al, be, [0, 2, 4], [1, 1, 3]):
        ref[:, dc] = a * X[:, sc] + b * ref[:, dc]
    assert relerr(fam.slot_read(1, 0, 5), ref) < 1e-15
    fam.slot_axpby(0, [2], 0, [2], alpha=3.0 - 2j, beta=0.0)
    assert relerr(fam.slot_read(0, 2, 1)[:, 0], (3.0 - 2j) * X[:, 2]) < 1e-15
    cur = fam.slot_read(1, 0, 5)                                           # conj_src: alpha conj(src) + beta dst
    fam.slot_axpby(1, [0, 1], 0, [3, 4], alpha=[1j, 2.0], beta=[0.0, 1.0], conj_src=True)
    got = fam.slot_read(1, 0, 2)
    src = fam.slot_read(0, 3, 2)
    assert relerr(got[:, 0], 1j * np.conj(src[:, 0])) < 1e-15 and relerr(got[:, 1], 2.0 * np.conj(src[:, 1]) + cur[:, 1]) < 1e-15
    # argument checks: column and slot ranges
    from wae_amd._lib import WaeError
    with pytest.raises(WaeError):
        fam.slot_read(0, 4, 2)
    with pytest.raises(WaeError):
        fam.slot_write(fam.NSLOTS, X)
    with pytest.raises(WaeError):
        fam.slot_read(3, 0, 1)                                             # never written


@pytest.mark.parametrize("op", [0, 1, 2])
def test_slot_forms_against_scipy(annulus, op):
    L, pb, fam, mats = annulus
    d = pb["d"]
    n = 5
    A = RNG.standard_normal((d, 6)) + 1j * RNG.standard_normal((d, 6))
    B = RNG.standard_normal((d, 7)) + 1j * RNG.standard_normal((d, 7))
    fam.slot_write(2, A)
    fam.slot_write(3, B)
    zs = 2 * np.pi * (np.array([310.0, 455.0, 520.0, 610.0, 700.0]) + 1j * np.linspace(-40, 40, n))
    C = np.array([L.coefficients(z) for z in zs])
    ac, bc = [5, 0, 3, 3, 1], [6, 2, 2, 0, 4]                              # scattered columns (copied side by side inside), repeats allowed
    got = fam.slot_forms(C, 2, ac, 3, bc, op=op)
    for i in range(n):
        M = sum(ck * Ak for ck, Ak in zip(C[i], mats)).tocsr()
        M = M if op == 0 else (M.T if op == 1 else M.conj().T)
        ref = np.vdot(A[:, ac[i]], M @ B[:, bc[i]])
        scale = np.linalg.norm(A[:, ac[i]]) * np.linalg.norm(M @ B[:, bc[i]])
        assert abs(got[i] - ref) <= 1e-12 * scale, (op, i, got[i], ref)
    # consecutive columns (read in place) and one coefficient row for all pairs
    got = fam.slot_forms(C[0], 2, [1, 2, 3], 3, [2, 3, 4], op=op)
    M = sum(ck * Ak for ck, Ak in zip(C[0], mats)).tocsr()
    M = M if op == 0 else (M.T if op == 1 else M.conj().T)
    for i in range(3):
        ref = np.vdot(A[:, 1 + i], M @ B[:, 2 + i])
        assert abs(got[i] - ref) <= 1e-12 * np.linalg.norm(A[:, 1 + i]) * np.linalg.norm(M @ B[:, 2 + i])


@pytest.mark.parametrize("op", [0, 2])
def test_arnoldi_from_slots_equals_arnoldi_through_host_memory(annulus, op):
    """Same start vectors, same operators: the Hessenberg matrices agree to rounding and a combination of the device-kept basis equals
    the combination of the basis the host call returned.  ritz_tol = 0 (every solve to the tolerance, no early exit, no replacement of
    poor start vectors: the two calls take the same m steps)."""
    L, pb, fam, mats = annulus
    d, T = pb["d"], fam.T
    nsys, m = 3, 3
    zs = 2 * np.pi * (np.array([333.0, 512.0, 777.0]) + 1j * np.array([10.0, -20.0, 5.0]))
    cA = np.array([L.coefficients(z) for z in zs])
    cM = np.zeros(T, dtype=complex)
    cM[-1] = -1.0
    V0 = RNG.standard_normal((d, 5)) + 1j * RNG.standard_normal((d, 5))
    H1, V1 = fam.arnoldi_batch(cA, cM, m, np.asfortranarray(V0[:, [4, 0, 2]]), op=op, tol=1e-12)
    fam.slot_write(0, V0)
    H2 = fam.arnoldi_slots(cA, cM, m, 0, [4, 0, 2], op=op, tol=1e-12)
    assert relerr(H2, H1) < 1e-9
    Y = RNG.standard_normal((nsys, m + 1)) + 1j * RNG.standard_normal((nsys, m + 1))
    fam.slot_write(1, np.zeros((d, 4), dtype=complex))                    # (an existing slot of that width keeps its columns: write zeros)
    fam.ritz_to_slot(Y, 1, [3, 1, 0], normalise=False)
    got = fam.slot_read(1, 0, 4)
    for s, c in enumerate([3, 1, 0]):
        assert relerr(got[:, c], V1[s] @ Y[s]) < 1e-8
    assert not got[:, 2].any()
    fam.ritz_to_slot(Y[:, :2], 1, [0, 1, 2], normalise=True)               # fewer coefficients than basis vectors; unit 2-norm
    got = fam.slot_read(1, 0, 3)
    for s in range(nsys):
        ref = V1[s][:, :2] @ Y[s, :2]
        assert relerr(got[:, s], ref / np.linalg.norm(ref)) < 1e-8
    from wae_amd._lib import WaeError
    with pytest.raises(WaeError):
        fam.ritz_to_slot(Y[:2], 1, [0, 1])                                 # not the shape of the basis on the device
    with pytest.raises(WaeError):
        fam.ritz_to_slot(np.ones((nsys, m + 2)), 1, [0, 1, 2])             # more coefficients than basis vectors


def test_perturb_from_slots_equals_perturb_through_host_memory(annulus):
    L, pb, fam, mats = annulus
    from wae_amd.nlevp import beyn
    d = pb["d"]
    G = np.array([300 - 100j, 600 - 100j, 600 + 100j, 300 + 100j]) * 2 * np.pi
    Om, P = beyn(L, G, l=6, K=1, N=32)[:2]
    res = fam.eig_residuals(np.array([L.coefficients(w_) for w_ in Om]), P=np.asfortranarray(P))
    k = int(np.argmin(res))
    assert res[k] <= 1e-5
    v, w = P[:, k], np.conj(P[:, k])
    N = 3
    T = fam.T
    saved = (L.active, L.mode, dict(L.params))
    L.params[L.eigval] = Om[k]
    L.params[L.auxval] = 0
    L.active, L.mode = [L.auxval, L.eigval], "householder"
    try:
        table = np.zeros((N + 1, N + 1, T), dtype=complex)
        for m in range(N + 1):
            for n in range(N + 1 - m):
                table[m, n] = L.coefficients(m, n)
    finally:
        L.active, L.mode = saved[0], saved[1]
        L.params.update(saved[2])
    fam.slot_write(0, np.column_stack([w, v, w]))
    for mode in (0, 1, 16):
        lam1, V1 = fam.perturb(table, N, v, w, norm_mode=mode, tol=1e-12, quiet=True)
        lam2, V2 = fam.perturb_slots(table, N, 0, 1, 0, 2, norm_mode=mode, tol=1e-12, quiet=True, vectors=True)
        assert relerr(lam2[1:], lam1[1:]) < 1e-9, mode
        kmax = N if mode != 16 else N - 1                                    # (+16: the vector of order N is not computed)
        assert relerr(V2[:, :kmax + 1], V1[:, :kmax + 1]) < 1e-8, mode
        lam3, V3 = fam.perturb_slots(table, N, 0, 1, 0, 0, norm_mode=mode, tol=1e-12, quiet=True)     # no vector leaves the device
        assert V3 is None and relerr(lam3[1:], lam1[1:]) < 1e-9


def test_householder_many_resident_equals_the_host_memory_form(annulus):
    """The lock-step Householder iteration with its vectors resident in HBM against the same iteration through host memory: eigenvalues,
    step counts and flags, right and left eigenvectors (normalised as Householder.jl:189-190: v' M v = 1, v_adj' L'(z) v = 1)."""
    L, pb, fam, mats = annulus
    from wae_amd.nlevp import beyn, householder_many
    from wae_amd.nlevp.local_solvers import householder_many_host
    d = pb["d"]
    G = np.array([300 - 100j, 900 - 100j, 900 + 100j, 300 + 100j]) * 2 * np.pi
    Om, P = beyn(L, G, l=10, K=1, N=48)[:2]
    res = fam.eig_residuals(np.array([L.coefficients(w_) for w_ in Om]), P=np.asfortranarray(P))
    ok = np.nonzero(res <= 1e-4)[0][:5]
    assert len(ok) >= 3, res
    z0 = list(Om[ok] * (1 + 2e-4))                                         # (pushed off the eigenvalues: two or three Newton steps)
    P0 = np.asfortranarray(P[:, ok])
    st_r, st_h = {}, {}
    for relax in (1.0, 0.8):
        tol = 1e-7 * 2 * np.pi * 500.0                                     # (relax = 0.8 converges linearly, a fifth of the error per step)
        res_r = householder_many(L, z0, maxiter=14, tol=tol, relax=relax, v0s=P0, stats=st_r)
        res_h = householder_many_host(L, z0, maxiter=14, tol=tol, relax=relax, v0s=P0, stats=st_h)
        assert "download_seconds" in st_r and "download_seconds" not in st_h
        T = fam.T
        cM = np.zeros(T, dtype=complex)
        cM[-1] = -1.0
        for (s1, n1, f1), (s2, n2, f2) in zip(res_r, res_h):
            w1, w2 = complex(s1.params[L.eigval]), complex(s2.params[L.eigval])
            assert abs(w1 - w2) <= 1e-10 * abs(w2), (relax, w1, w2, n1, n2, f1, f2)
            assert f1 == f2 and f1 in (0, 1) and abs(n1 - n2) <= 1, (relax, f1, f2, n1, n2)
            for a, b in ((s1.v, s2.v), (s1.v_adj, s2.v_adj)):
                ov = abs(np.vdot(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))
                assert ov > 1 - 1e-8, (relax, ov)
            # the normalisations themselves, from the returned vectors
            Mv = fam.spmv(cM, np.asfortranarray(s1.v.reshape(d, 1)))[:, 0]
            assert abs(np.vdot(s1.v, Mv) - 1) < 1e-9
            saved = (L.active, L.mode, dict(L.params))
            L.params[L.eigval] = w1
            L.params[L.auxval] = s1.params[L.auxval]
            L.active, L.mode = [L.eigval], "all"
            try:
                cD = L.coefficients(w1, 1)
            finally:
                L.active, L.mode = saved[0], saved[1]
                L.params.update(saved[2])
            Dv = fam.spmv(cD, np.asfortranarray(s1.v.reshape(d, 1)))[:, 0]
            assert abs(np.vdot(s1.v_adj, Dv) - 1) < 1e-8
    assert householder_many(L, [], v0s=np.zeros((d, 0))) == []


def test_single_start_solvers_resident_equal_the_host_memory_forms(annulus):
    """`householder` and `mslp` with nev = 1 keep their eigenvector pair in HBM by default (slots 4-7, column 0); resident=False is the
    host-memory iteration of rounds 1-3: same eigenvalue, same step count (+-1), same flag, same normalised vectors up to a phase."""
    L, pb, fam, mats = annulus
    from wae_amd.nlevp import beyn, householder, mslp
    G = np.array([300 - 100j, 600 - 100j, 600 + 100j, 300 + 100j]) * 2 * np.pi
    Om, P = beyn(L, G, l=6, K=1, N=32)[:2]
    res = fam.eig_residuals(np.array([L.coefficients(w_) for w_ in Om]), P=np.asfortranarray(P))
    k = int(np.argmin(res))
    z0, v0 = Om[k] * (1 + 3e-4), P[:, k]
    tol = 1e-8 * 2 * np.pi * 500.0
    for solver, kw in ((householder, {}), (householder, {"order": 2}), (mslp, {}), (mslp, {"order": 2, "num_order": 1})):
        for start in (None, v0):
            a = solver(L, z0, maxiter=12, tol=tol, v0=start, output=False, **kw)
            b = solver(L, z0, maxiter=12, tol=tol, v0=start, output=False, resident=False, **kw)
            wa, wb = complex(a[0].params[L.eigval]), complex(b[0].params[L.eigval])
            assert abs(wa - wb) <= 1e-9 * abs(wb), (solver.__name__, kw, wa, wb)
            assert a[2] == b[2] and abs(a[1] - b[1]) <= 1, (solver.__name__, kw, a[1:], b[1:])
            for x, y in ((a[0].v, b[0].v), (a[0].v_adj, b[0].v_adj)):
                ov = abs(np.vdot(x, y)) / (np.linalg.norm(x) * np.linalg.norm(y))
                assert ov > 1 - 1e-8, (solver.__name__, kw, ov)
            assert abs(np.linalg.norm(a[0].v) / np.linalg.norm(b[0].v) - 1) < 1e-8          # (the same normalisation)
