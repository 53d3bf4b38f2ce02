"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes -> libwaehip.so), against the CPU
oracle on the same inputs, and against the committed golden values of the reference's tutorials.

Tolerances: fused SpMV-sum <= 1e-13 relative (max-norm per column); linear solves to the inner
tolerance (reported true residual <= tol, solution <= 1e-8 relative vs the direct oracle solve);
Householder/mslp eigenvalues <= 1e-10 relative; Beyn moments/eigenvalues <= 1e-8 relative; Taylor
coefficients <= 1e-8 relative up to order 20.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fixtures as F
from oracle import solvers as OS
from wae_amd import _lib
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import (LinearOperatorFamily, Term, beyn, compute_moment_matrices, householder, inveriter,
                           moments2eigs, mslp, perturb_fast_, pow1, pow2)

pytestmark = pytest.mark.gpu
G = F.golden()
c = lambda p: complex(p[0], p[1])
RNG = np.random.default_rng(0)


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def rijke():
    Lo = F.rijke_family(n=0.7, tau=1.3e-3)
    Lp = helmholtz_family(F.rijke_terms(), n=0.7, tau=1.3e-3)
    yield Lo, Lp
    Lp._drop_device()


def test_native_library_is_loaded():
    assert b"gfx950" in _lib.lib().wae_version()
    assert _lib.device_count() >= 1


@pytest.mark.parametrize("r", [1, 3, 8, 19])
def test_spmv_sum_parity_rijke(rijke, r):
    Lo, Lp = rijke
    d = Lo.size()
    X = RNG.standard_normal((d, r)) + 1j * RNG.standard_normal((d, r))
    z = 1500.0 + 40j
    for args in [(z,), (z, 1), (z, 2)]:
        Ao = Lo(*args)
        Ap = Lp(*args)
        Yo = Ao @ X
        assert relerr(Ap @ X, Yo) < 1e-13
        assert relerr(Ap.H @ X, Ao.conj().T @ X) < 1e-13
    x = X[:, 0]
    assert relerr(Lp(z) @ x, Lo(z) @ x) < 1e-13          # 1-D input


def test_spmv_sum_compact_and_householder_modes(rijke):
    Lo, Lp = rijke
    d = Lo.size()
    x = RNG.standard_normal(d) + 1j * RNG.standard_normal(d)
    for L in (Lo, Lp):
        L.params["ω"] = 1000.0 + 5j
        L.active = ["ω", "τ"]; L.mode = "compact"
    try:
        for m in range(3):
            for n in range(3):
                yo = Lo(m, n) @ x
                assert relerr(Lp(m, n) @ x, yo) < 1e-13 or np.max(np.abs(yo)) == 0
        for L in (Lo, Lp):
            L.active = ["λ", "ω"]; L.mode = "householder"; L.params["λ"] = 0.3 + 0.2j
        for m in range(2):
            for n in range(3):
                yo = Lo(m, n) @ x
                assert relerr(Lp(m, n) @ x, yo) < 1e-13 or np.max(np.abs(yo)) == 0
    finally:
        for L in (Lo, Lp):
            L.active = ["ω"]; L.mode = "all"


def test_spmv_multi_parity(rijke):
    Lo, Lp = rijke
    d, T = Lo.size(), len(Lo.terms)
    X = RNG.standard_normal((d, T)) + 1j * RNG.standard_normal((d, T))
    cs = RNG.standard_normal(T) + 1j * RNG.standard_normal(T)
    want = sum(cs[k] * (Lo.terms[k].coeff @ X[:, k]) for k in range(T))
    got = Lp.device().spmv_multi(cs, X)
    assert relerr(got, want) < 1e-13


def test_spmv_annulus_small_all_batch_widths():
    pb = annulus.build("small")
    Lp = helmholtz_family(pb["terms"])
    T = pb["terms"]
    z = 2 * np.pi * (500 + 20j)
    A = (z * z * T["M"] + T["K"] + z * 1e15 * T["C"] + np.exp(-1j * z * 1e-3) * T["Q"]).tocsr()
    for r in (1, 2, 4, 8, 16, 33):
        X = RNG.standard_normal((pb["d"], r)) + 1j * RNG.standard_normal((pb["d"], r))
        assert relerr(Lp(z) @ X, A @ X) < 1e-13
        assert relerr(Lp(z).H @ X, A.conj().T @ X) < 1e-13
    assert Lp.device().spmv_bytes(r=1, mask=[1, 1, 1, 1, 0]) == sum(
        T[k].nnz * 20 + (pb["d"] + 1) * 4 for k in "MKCQ") + 2 * pb["d"] * 16
    Lp._drop_device()


def test_dense_and_ragged_terms():
    """qep1 (dense 3x3 terms) and a family with empty rows / an empty term."""
    g = G["G8"]
    T = LinearOperatorFamily()
    T.push(Term(np.array(g["A2"], dtype=complex), (pow2,), (("λ",),), "λ^2", "A2"))
    T.push(Term(np.array(g["A1"], dtype=complex), (pow1,), (("λ",),), "λ", "A1"))
    T.push(Term(np.array(g["A0"], dtype=complex), (), (), "", "A0"))
    z = 0.3 - 0.8j
    want = z * z * np.array(g["A2"]) + z * np.array(g["A1"]) + np.array(g["A0"])
    assert relerr(T(z).toarray(), want) < 1e-14
    assert relerr(T(z, 1).toarray(), 2 * z * np.array(g["A2"]) + np.array(g["A1"])) < 1e-14
    Gam = [2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j]
    Om, P, Sig = beyn(T, Gam, l=6, return_sigma=True)
    assert np.sum(Sig < 1e-10 * Sig[0]) == g["n_small_sigma"]
    for w in [c(e) for e in g["eigs_inside"]]:
        assert np.min(np.abs(Om - w)) < 1e-9
    T._drop_device()
    # ragged: rows without entries, one structurally empty term
    d = 50
    A = sp.random(d, d, density=0.05, random_state=1, format="lil", dtype=float)
    A[7, :] = 0; A[:, 3] = 0
    A = sp.csr_matrix(A) + 0j
    E = sp.csr_matrix((d, d), dtype=complex)
    Lr = LinearOperatorFamily()
    Lr.push(Term(A, (pow1,), (("λ",),), "λ", "A"))
    Lr.push(Term(E, (pow2,), (("λ",),), "λ^2", "E"))
    Lr.push(Term(sp.identity(d, dtype=complex, format="csr") * (1 + 2j), (), (), "", "I"))
    x = RNG.standard_normal(d) + 0j
    assert relerr(Lr(2.0 + 1j) @ x, (2.0 + 1j) * (A @ x) + (1 + 2j) * x) < 1e-14
    Lr._drop_device()


def test_solve_parity_rijke(rijke):
    Lo, Lp = rijke
    d = Lo.size()
    Lp.solver_ref = 2 * np.pi * 400
    B = RNG.standard_normal((d, 5)) + 1j * RNG.standard_normal((d, 5))
    B[:, 3] = 0.0                                            # a zero right-hand side must give zero
    for z in (2 * np.pi * (150 + 5j), 2 * np.pi * (640 - 5j), 1710.0 + 9.0j):
        Xo = OS._solve(Lo(z), B)
        Ap = Lp(z)
        X = Ap.solve(B, tol=1e-12)
        info = Lp.device().last_info
        assert info["n_unconverged"] == 0 and info["relres_max"] <= 1e-12
        assert np.all(X[:, 3] == 0)
        assert relerr(X, Xo) < 1e-8
        Xh = Ap.H.solve(B, tol=1e-12)
        assert relerr(Xh, OS._solve(Lo(z).conj().T.tocsc(), B)) < 1e-8


def test_solve_many_systems_in_lockstep(rijke):
    """ncoef = r: every column has its own coefficient set (different z)."""
    Lo, Lp = rijke
    d = Lo.size()
    fam = Lp.ensure_solver()
    zs = 2 * np.pi * np.array([200 + 5j, 300 - 4j, 555 + 1j, 900 + 30j, 150 - 5j, 999 + 5j, 431 + 2j])
    B = RNG.standard_normal((d, len(zs))) + 1j * RNG.standard_normal((d, len(zs)))
    ct = np.array([Lp.coefficients(z) for z in zs])
    X = fam.solve(ct, B, tol=1e-12, maxit=400)
    assert fam.last_info["n_unconverged"] == 0
    for j, z in enumerate(zs):
        assert relerr(X[:, j], OS._solve(Lo(z), B[:, j])) < 1e-8


def test_beyn_moments_parity_C1():
    """BASELINE.json configs[0]: Rijke P1, quadratic K+ωYC+ω²M (n=0), Beyn l=5, N=16 per edge."""
    Lo = F.rijke_family(n=0.0)
    Lp = helmholtz_family(F.rijke_terms(), n=0.0)
    Gam = np.array([150 + 5j, 150 - 5j, 1000 - 5j, 1000 + 5j]) * 2 * np.pi
    Lp.solver_ref = 2 * np.pi * 500
    Ao = OS.compute_moment_matrices(Lo, Gam, OS.initial_V(Lo.size(), 5), K=1, N=16)
    Ap = compute_moment_matrices(Lp, Gam, l=5, K=1, N=16)
    assert Ap.shape == Ao.shape
    assert relerr(Ap, Ao) < 1e-8
    assert Lp.device().last_info["n_unconverged"] == 0
    Om_p, _, Sp = moments2eigs(Ap, return_sigma=True)
    Om_o, _, So = OS.moments2eigs(Ao, return_sigma=True)
    assert np.allclose(Sp[:2], So[:2], rtol=1e-8)
    # K = 2 (four moments) and empty contour list
    Ao2 = OS.compute_moment_matrices(Lo, Gam, OS.initial_V(Lo.size(), 3), K=2, N=4)
    Ap2 = compute_moment_matrices(Lp, Gam, l=3, K=2, N=4)
    assert relerr(Ap2, Ao2) < 1e-8
    A0 = compute_moment_matrices(Lp, Gam, l=3, K=1, points=(np.zeros(0, complex), np.zeros(0, complex)))
    assert np.all(A0 == 0)
    Lp._drop_device()


def test_G1_householder_and_G2_perturbation():
    Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    sol, n, flag = householder(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    w = c(G["G1"]["omega"])
    assert abs(sol.params["ω"] - w) < 1e-10 * abs(w)
    # the reference records 7 / 6 / 5 iterations for this call (three notebook runs) and a converged flag; the device path: 7
    assert flag in (0, 1) and 5 <= n <= 9
    first = next(i for i, zk in enumerate(sol.history) if abs(zk - w) < 1e-9 * abs(w))
    assert first <= G["G1"]["iterations"] - 2
    for mine, ref in zip(sol.history, G["G1"]["iterates"]):
        assert abs(mine - c(ref)) < 1e-6 * abs(c(ref))
    perturb_fast_(sol, Lp, "τ", 20)
    lam = sol.eigval_pert["τ/Taylor"]
    for k, ref in enumerate(G["G2"]["taylor"]):
        assert abs(lam[k] - c(ref)) < 1e-8 * abs(c(ref)), (k, lam[k], c(ref))
    assert abs(sol("τ", 0.001 + 1e-5, 20) - c(G["G3"]["taylor20_estimate"])) < 1e-7
    Lp._drop_device()


def test_G5_mslp_active_flame():
    Lp = helmholtz_family(F.rijke_terms(), n=1.0, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    sol, n, flag = mslp(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    w = c(G["G5"]["omega"])
    assert abs(sol.params["ω"] - w) < 1e-10 * abs(w)
    # the reference stops after 8 iterations at |Δω| <= 1e-11 (absolute, ~1e-14 relative): whether the last
    # iterates dip below that floor is rounding noise, so pin the convergence history instead of the final count
    assert flag in (0, 1) and n <= G["G5"]["iterations"] + 2          # (the reference: 8 iterations; the device path: 8)
    first = next(i for i, zk in enumerate(sol.history) if abs(zk - w) < 1e-9 * abs(w))
    assert first <= G["G5"]["iterations"] - 1
    # padesolve = the same iteration with householder's flags; order 2 exercises the in-loop perturbation solve
    from wae_amd.nlevp import padesolve
    Lp.params["τ"] = 0.001
    sol3, n3, flag3 = padesolve(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-9, order=2, num_order=1)
    assert flag3 in (0, 1) and abs(sol3.params["ω"] - w) < 1e-8 * abs(w) and n3 <= 8
    sol4, n4, flag4 = householder(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-9, order=3)
    assert flag4 in (0, 1) and abs(sol4.params["ω"] - w) < 1e-8 * abs(w) and n4 <= 6
    Lp._drop_device()


def test_G4_G6_perturbation_order_30_on_the_device():
    """perturb_fast!(sol, L, :τ, 30) (perturbation.jl:374-444 via LinOpFam.jl:575-589) on the device, from the G5 base state
    (n = 1, mslp to 1e-11), against the reference's recorded outputs: the full 30-entry convergence-radius table
    (docs/src/tutorial_04_perturbation_theory.md:241-271), the 30th-order estimate at τ + 5e-4 (:210), the 20th-order
    estimate and the six printed Taylor coefficients (:128,142).  Tolerances: the table is a ratio of consecutive
    coefficients, so it inherits the inner tolerance (1e-12) times the conditioning of the recurrence -- 1e-7 relative is
    asserted (the CPU oracle with a direct solver reaches 1e-8); estimates 1e-7 absolute like the oracle pin."""
    from wae_amd.nlevp import conv_radius
    Lp = helmholtz_family(F.rijke_terms(), n=1.0, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    Lp.solver_tol = 1e-13
    sol, n, flag = mslp(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    assert abs(sol.params["ω"] - c(G["G5"]["omega"])) < 1e-10 * abs(c(G["G5"]["omega"]))
    perturb_fast_(sol, Lp, "τ", 30)
    info = Lp.device().last_info
    assert info["n_unconverged"] == 0
    lam30 = sol.eigval_pert["τ/Taylor"]
    assert len(lam30) == 31 and len(sol.v_pert["τ/Taylor"]) == 31
    r = conv_radius(lam30)
    ref = np.array(G["G4"]["conv_radius"])
    assert len(r) == len(ref) == 30
    assert np.max(np.abs(r - ref) / ref) < 1e-7, np.max(np.abs(r - ref) / ref)
    est30 = sol("τ", 0.0015, 30) / 2 / np.pi
    assert abs(est30 - c(G["G4"]["taylor30_estimate_over_2pi_at_tau_plus_5e-4"])) < 1e-7
    sol.eigval_pert["τ/Taylor"] = lam30[:21]
    for k, refk in enumerate(G["G6"]["taylor_6digits"]):
        assert abs(lam30[k] - c(refk)) < 2e-5 * abs(c(refk))
    assert abs(sol("τ", 0.0015, 20) - c(G["G6"]["taylor20_estimate"])) < 1e-7
    # the re-solve at the perturbed delay from the 20th-order estimate (tutorial :158-171)
    Lp.params["τ"] = 0.0015
    sol2, n2, _ = mslp(Lp, sol("τ", 0.0015, 20), maxiter=20, tol=1e-11)
    assert abs(sol2.params["ω"] - c(G["G6"]["omega_exact"])) < 1e-9 * abs(c(G["G6"]["omega_exact"])) * 10
    Lp._drop_device()


def test_newton_variants_match_oracle():
    """inveriter / rf2s / lancaster: same start values on the oracle and on the device path."""
    from wae_amd.nlevp import lancaster, rf2s
    Lo = F.rijke_family(n=0.01, tau=0.001)
    Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    w = c(G["G1"]["omega"])
    so, no_, fo = OS.inveriter(Lo, 1710 + 9j, maxiter=20, tol=1e-9)
    sp_, np_, fp = inveriter(Lp, 1710 + 9j, maxiter=20, tol=1e-9)
    # the last Newton steps solve with a numerically singular L(z): the iterative inner solver reaches the same
    # eigenvalue but may need a few more outer steps than the direct solver
    assert fo == fp == 0 and no_ <= np_ <= no_ + 6
    assert abs(sp_.params["ω"] - so.params["ω"]) < 1e-8 * abs(w) and abs(sp_.params["ω"] - w) < 1e-8 * abs(w)
    x = sp_.v
    so, no_, fo = OS.rf2s(Lo, 1710 + 9j, maxiter=20, tol=1e-9, x0=x, y0=np.conj(x))
    sp2, np2, fp2 = rf2s(Lp, 1710 + 9j, maxiter=20, tol=1e-9, x0=x, y0=np.conj(x))
    assert fo == fp2 == 0 and abs(sp2.params["ω"] - so.params["ω"]) < 1e-8 * abs(w)
    so, no_, fo = OS.lancaster(Lo, 1710 + 9j, maxiter=10, tol=1e-9)
    sp3, np3, fp3 = lancaster(Lp, 1710 + 9j, maxiter=10, tol=1e-9)
    # Rayleigh-quotient iteration from 1710+9i: both sides converge to the G1 eigenvalue (the oracle in 3 steps)
    assert fo == 0 and fp3 == 0, (fo, fp3)
    assert abs(sp3.params["ω"] - so.params["ω"]) < 1e-8 * abs(w) and abs(sp3.params["ω"] - w) < 1e-8 * abs(w)
    assert no_ <= np3 <= no_ + 4
    Lp._drop_device()


def test_perturb_device_call_all_modes():
    """wae_perturb (one device call) vs the host-orchestrated recurrence and vs the oracle, for perturb!,
    perturb_fast! and perturb_norm! (LinOpFam.jl:546-618)."""
    from wae_amd.nlevp import perturb_, perturb_norm_
    from wae_amd.nlevp import perturbation as PP
    Lo = F.rijke_family(n=0.01, tau=0.001)
    Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    so, _, _ = OS.householder(Lo, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    sp_, _, _ = householder(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    N = 6
    for name, fo, fp in (("perturb", OS.perturb_, perturb_), ("fast", OS.perturb_fast_, perturb_fast_),
                         ("norm", OS.perturb_norm_, perturb_norm_)):
        fo(so, Lo, "τ", N)
        fp(sp_, Lp, "τ", N)
        lo, lp = so.eigval_pert["τ/Taylor"], sp_.eigval_pert["τ/Taylor"]
        for k in range(1, N + 1):
            assert abs(lp[k] - lo[k]) < 1e-7 * abs(lo[k]), (name, k, lp[k], lo[k])
        # eigenvector coefficients: compare up to the phase/normalisation of the base vectors via |<v_k, v_0>|-free norms
        vo, vp = so.v_pert["τ/Taylor"], sp_.v_pert["τ/Taylor"]
        ph = np.vdot(vo[0], vp[0]) / abs(np.vdot(vo[0], vp[0]))
        for k in range(1, 4):
            assert np.linalg.norm(vp[k] - ph * vo[k]) < 1e-5 * np.linalg.norm(vo[k]), (name, k)
    # device call == host-orchestrated recurrence on the same inputs
    Lp.params = sp_.params; Lp.active = ["ω", "τ"]; Lp.mode = "compact"
    try:
        l1, _ = PP._recurrence(Lp, 5, sp_.v, sp_.v_adj, normalize=True)
        l2, _ = PP._recurrence_host(Lp, 5, sp_.v, sp_.v_adj, normalize=True)
    finally:
        Lp.active = ["ω"]; Lp.mode = "all"
    assert np.allclose(l1[1:], l2[1:], rtol=1e-7)
    Lp._drop_device()


def test_beyn_annulus_small_vs_oracle_golden():
    """The bench configuration at 8 736 DoF (4 terms incl. the non-symmetric flame term, 12 flames, complex probe
    matrix) against the oracle's direct-solver Beyn result committed in tests/golden/annulus_small_beyn.json."""
    import json
    import os
    g = json.load(open(os.path.join(F.GOLDEN_DIR, "annulus_small_beyn.json")))
    from wae_amd.helmholtz.family import annulus_family
    L, pb = annulus_family("small", n=g["n"], tau=g["tau"])
    assert pb["d"] == g["d"]
    L.solver_tol = 1e-12
    L.solver_ref = 2 * np.pi * 500.0
    Gam = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    rng = np.random.default_rng(g["seed_V"])
    V = rng.standard_normal((pb["d"], g["l"])) + 1j * rng.standard_normal((pb["d"], g["l"]))
    A = compute_moment_matrices(L, Gam, V, K=1, N=g["N"])
    assert L.device().last_info["n_unconverged"] == 0
    Om, P, S = moments2eigs(A, return_sigma=True)
    want = np.array([c(e) for e in g["eigs"]])
    for w in want:
        assert np.min(np.abs(Om - w)) < 1e-8 * abs(w)
    ns = len(want)
    assert np.allclose(S[:ns], g["sigma"][:ns], rtol=1e-7) and S[ns] < 1e-8 * S[0]
    L._drop_device()


def _qep1():
    g = G["G8"]
    T = LinearOperatorFamily()
    T.push(Term(np.array(g["A2"], dtype=complex), (pow2,), (("λ",),), "λ^2", "A2"))
    T.push(Term(np.array(g["A1"], dtype=complex), (pow1,), (("λ",),), "λ", "A1"))
    T.push(Term(np.array(g["A0"], dtype=complex), (), (), "", "A0"))
    return T


def test_G8_small_dense_solvers():
    """qep1 (tutorial_00): mslp from 0 -> 1/3, pole/zero count = 5, trace iteration, all on the device path
    (d = 3: the 'multigrid' degenerates to the dense coarse solve)."""
    from wae_amd.nlevp import count_poles_and_zeros, traceiter
    g = G["G8"]
    T = _qep1()
    Gam = [2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j]
    n = count_poles_and_zeros(T, Gam)
    assert abs(n - g["count_poles_and_zeros"]) < 1e-2
    sol, it, flag = mslp(T, 0, tol=1e-10)
    assert abs(sol.params["λ"] - c(g["mslp_from_0_tol1e-10"]["omega"])) < 1e-9 and flag == 0
    assert abs(it - g["mslp_from_0_tol1e-10"]["iterations"]) <= 2
    T2 = _qep1()
    sol, it, flag = traceiter(T2, 0.4 + 0.1j, maxiter=30, tol=1e-10)
    assert flag == 0 and min(abs(sol.params["λ"] - c(e)) for e in g["eigs_inside"]) < 1e-8
    T._drop_device(); T2._drop_device()


def test_solve_driver_rijke():
    """solve(L, Γ): incremental Beyn + deflation + local refinement (solver.jl:36-184 intent) finds the two passive
    Rijke modes (272.06 Hz, 694.97 Hz) with machine-precise local refinement."""
    from wae_amd.nlevp import solve
    Lp = helmholtz_family(F.rijke_terms(), n=0.0)
    Lp.solver_ref = 2 * np.pi * 500
    Gam = np.array([150 + 50j, 150 - 50j, 1000 - 50j, 1000 + 50j]) * 2 * np.pi
    eig = solve(Lp, Gam, dl=3, N=32, tol=1e-9, maxcycles=2)
    inside = sorted(w.real / 2 / np.pi for w, v in eig.items() if v[1])
    assert len(inside) == 2 and abs(inside[0] - 272.0643) < 1e-3 and abs(inside[1] - 694.9677) < 1e-3
    Lp._drop_device()


def test_many_terms_mixed_patterns_bloch_like():
    """A Bloch-like family (Helmholtz.jl:507-512 pushes 3-6 sub-operators per matrix with phase factors
    exp(+-i b 2pi/N)): 11 terms, several sparsity patterns, complex-valued planes, non-symmetric parts; SpMV (N/T/C),
    batched solve and Beyn moments against the oracle's assembled matrices."""
    from oracle import nlevp as ON
    from wae_amd.nlevp import exp_pm
    t = F.rijke_terms()
    d = t["M"].shape[0]
    rows = np.arange(d)
    def part(A, k, m):          # rows with index % m == k
        D = sp.diags((rows % m == k).astype(float))
        return sp.csr_matrix(D @ A)
    pieces = [("M", part(t["M"], 0, 3), 1.0), ("M", part(t["M"], 1, 3), 1.0), ("M", part(t["M"], 2, 3), 1.0),
              ("K", part(t["K"], 0, 2) * (0.6 + 0.8j), 0.6 - 0.8j), ("K", part(t["K"], 1, 2), 1.0)]
    fplus, fminus = exp_pm(+1), exp_pm(-1)
    oplus, ominus = ON.exp_pm(+1), ON.exp_pm(-1)
    Lp = LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)])
    Lo = ON.LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)])
    for L, p2, p1, T_, fp, fm in ((Lp, pow2, pow1, Term, fplus, fminus), (Lo, ON.pow2, ON.pow1, ON.Term, oplus, ominus)):
        L.params["φ"] = 2 * np.pi / 12; L.params["Y"] = 1e15
        conv = (lambda A: sp.csc_matrix(A)) if L is Lo else (lambda A: A)
        for i, (nm, A, sc) in enumerate(pieces):    # push! merges equal (func, params) signatures: one symbol per term
            if nm == "M":   # ω² · e^{±i b φ} · M_i
                L.params[f"b{i}"] = 2.0
                L.push(T_(conv(A * sc), (p2, fp if i % 2 == 0 else fm), (("ω",), (f"b{i}", "φ")), f"ω^2*ph{i}", f"M{i}"))
            else:           # complex-scaled stiffness pieces with a constant coefficient
                L.params[f"s{i}"] = 1.0
                L.push(T_(conv(A * sc), (p1,), ((f"s{i}",),), f"s{i}", f"K{i}"))
        L.push(T_(conv(t["C"]), (p1, p1), (("ω",), ("Y",)), "ω*Y", "C"))
        L.params["sq"] = 1.0; L.params["st"] = 1.0
        L.push(T_(conv(t["Q"]), (p1,), (("sq",),), "sq", "Q"))
        L.push(T_(conv(t["Q"].T.tocsr() * 0.5j), (p1,), (("st",),), "st", "Qt"))
        L.push(T_(conv(-t["M"]), (p1,), (("λ",),), "-λ", "__aux__"))
    assert len(Lp.terms) == len(Lo.terms) >= 9
    x = RNG.standard_normal((d, 9)) + 1j * RNG.standard_normal((d, 9))
    z = 2 * np.pi * (300 + 30j)
    Ao, Ap = Lo(z), Lp(z)
    assert relerr(Ap @ x, Ao @ x) < 1e-13
    assert relerr(Ap.H @ x, Ao.conj().T @ x) < 1e-13
    assert relerr(Lp(z, 1) @ x, Lo(z, 1) @ x) < 1e-13
    Lp.solver_ref = 2 * np.pi * 400
    X = Ap.solve(x, tol=1e-12)
    assert relerr(X, OS._solve(Ao, x)) < 1e-7
    Xh = Ap.H.solve(x, tol=1e-12)
    assert relerr(Xh, OS._solve(sp.csc_matrix(Ao.conj().T), x)) < 1e-7
    Lp._drop_device()


def test_beyn_moments_with_projected_guesses_match_plain_and_oracle():
    """wae_beyn_moments_rb: snapshot points solved from a zero guess, every other point from the Galerkin projection on
    the snapshots.  The stopping test is unchanged (relative to the same right-hand side), so the moments agree with
    the plain path to the inner tolerance, and the eigenvalues with the oracle's golden values."""
    import json
    import os
    gold = json.load(open(os.path.join(F.GOLDEN_DIR, "annulus_small_beyn.json")))
    from wae_amd.helmholtz.family import annulus_family
    Lp, pb = annulus_family("small", n=gold["n"], tau=gold["tau"])
    Lp.solver_tol = 1e-11
    Lp.solver_ref = 2 * np.pi * 500.0
    Gam = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    d = pb["d"]
    rng = np.random.default_rng(gold["seed_V"])
    V = rng.standard_normal((d, gold["l"])) + 1j * rng.standard_normal((d, gold["l"]))
    A0 = compute_moment_matrices(Lp, Gam, V, K=1, N=gold["N"], rb=0)
    fam = Lp.device()
    its0 = fam.last_info["iters_total"]
    A1 = compute_moment_matrices(Lp, Gam, V, K=1, N=gold["N"], rb=24)
    info = fam.last_info
    assert info["n_unconverged"] == 0 and info["snapshots"] == 24
    assert relerr(A1, A0) < 1e-8
    assert info["iters_total"] < 0.6 * its0                      # the projection does most of the work
    Om, P = moments2eigs(A1)
    for w in np.array([c(e) for e in gold["eigs"]]):
        assert np.min(np.abs(Om - w)) < 1e-8 * abs(w)
    # a caller-owned store and the "rebuild from raw snapshots" mode (what the multi-GPU driver uses after its all-gather)
    import torch
    from wae_amd.nlevp.beyn import coefficient_table, gauss_points, snapshot_split
    zs, ws = gauss_points(Gam, gold["N"])
    ct = coefficient_table(Lp, zs)
    idx, rest = snapshot_split(len(zs), 24)
    l = V.shape[1]
    store = torch.empty(24 * d * l * 2, dtype=torch.float64, device="cuda:0")
    buf = torch.zeros(d * l * 2 * 2, dtype=torch.float64, device="cuda:0")
    kw = dict(K=1, tol=Lp.solver_tol, maxit=Lp.solver_maxit, out_dev=buf.data_ptr())
    fam.beyn_moments_rb(zs[idx[:12]], ws[idx[:12]], ct[idx[:12]], V, 0, 12, Q_dev=store.data_ptr(), **kw)           # "rank 0"
    fam.beyn_moments_rb(zs[idx[12:]], ws[idx[12:]], ct[idx[12:]], V, 0, 12, Q_dev=store[12 * d * l * 2:].data_ptr(),
                        accumulate=True, **kw)                                                                     # "rank 1"
    fam.beyn_moments_rb(zs[rest], ws[rest], ct[rest], V, 1, 24, slot0=24, Q_dev=store.data_ptr(), accumulate=True, **kw)
    A2 = buf.cpu().numpy().view(np.complex128).reshape((d, l, 2), order="F")
    assert relerr(A2, A0) < 1e-8
    Lp._drop_device()


def test_eig_residuals_entry(rijke):
    """wae_eig_residuals against the same quantity assembled from the oracle's matrices; host and device inputs."""
    import torch
    Lo, Lp = rijke
    d = Lo.size()
    fam = Lp.device()
    P = RNG.standard_normal((d, 5)) + 1j * RNG.standard_normal((d, 5))
    oms = 2 * np.pi * np.array([200 + 3j, 350 - 8j, 500.0, 710 + 40j, 90 + 1j])
    C = np.array([Lp.coefficients(w) for w in oms])
    want = []
    for j, w in enumerate(oms):
        cs = Lo.coefficients(w)
        parts = [ck * (t.coeff @ P[:, j]) for ck, t in zip(cs, Lo.terms) if ck is not None and ck != 0]
        want.append(np.linalg.norm(sum(parts)) / sum(np.linalg.norm(p) for p in parts))
    got = fam.eig_residuals(C, P=P)
    assert np.allclose(got, want, rtol=1e-12)
    Pt = torch.from_numpy(np.ascontiguousarray(P.T)).to("cuda:0")            # (n, d) row-major == column-major d x n
    assert np.allclose(fam.eig_residuals(C, P_dev=Pt.data_ptr()), want, rtol=1e-12)


def test_generate_subspace_and_project_match_oracle():
    """Reduced-basis Beyn (beyn.jl:429-595) on the Rijke tube with a finite outlet impedance (with Y = 1e15 the plain
    residual norm of the reference's greedy test is dominated by the penalty rows and every sample point is added):
    same greedy outcome as the oracle up to threshold ties, same subspace quality, and Beyn on the projected family
    reproduces the full-order eigenvalues."""
    from wae_amd.nlevp import generate_subspace, project
    Lo = F.rijke_family(n=0.0, Y=100.0)
    Lp = helmholtz_family(F.rijke_terms(), n=0.0, Y=100.0)
    Lp.solver_ref = 2 * np.pi * 400.0
    d = Lo.size()
    Y = np.random.default_rng(3).standard_normal((d, 3)) + 0j
    Gam = np.array([150 + 50j, 150 - 50j, 1000 - 50j, 1000 + 50j]) * 2 * np.pi
    tol = 1e-4
    Qo, ro = OS.generate_subspace_contour(Lo, Y, tol, Gam, 8)
    Qp, rp = generate_subspace(Lp, Y, tol, Gam, 8)
    assert abs(Qp.shape[1] - Qo.shape[1]) <= 2 and rp.max() <= tol and ro.max() <= tol
    assert np.allclose(Qp.conj().T @ Qp, np.eye(Qp.shape[1]), atol=1e-10)
    # (one tie decided the other way sends the two greedy runs to different later sample points, so the spaces are not
    # nested; what both guarantee is the residual bound asserted above)
    P = project(Lp, Qp)
    Po = OS.project(Lo, Qp)
    z = 2 * np.pi * (400 + 7j)
    x = RNG.standard_normal((Qp.shape[1], 2)) + 0j
    assert relerr(P(z) @ x, Po(z) @ x) < 1e-12
    Om, Pv = beyn(P, Gam, l=6, N=32, output=False)[:2]
    Of = OS.beyn(Lo, Gam, l=6, N=32)[0]
    for w in Of:
        assert np.min(np.abs(Om - w)) < 1e-7 * abs(w)
    P._drop_device()
    Lp._drop_device()


@pytest.mark.parametrize("l,nranks", [(16, 2), (8, 8)])
def test_projected_guesses_column_split_exchange(l, nranks):
    """The multi-GPU snapshot phase in one process: "rank r" takes ALL snapshot points for its share of the probe columns
    (two ranks with 8 columns each; eight ranks with ONE column each = 64 systems per lock-step chunk, the 8-GPU run of C3)
    (wae_beyn_moments_rb mode 0 with l_total/col0), the finished bases are exported, concatenated as the all-gather would,
    imported, and the remaining points run in mode 2.  Same moments as the plain path."""
    import json
    import os
    import torch
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp.beyn import coefficient_table, gauss_points, snapshot_split, spread_order
    gold = json.load(open(os.path.join(F.GOLDEN_DIR, "annulus_small_beyn.json")))
    Lp, pb = annulus_family("small", n=gold["n"], tau=gold["tau"])
    Lp.solver_tol = 1e-11
    Lp.solver_ref = 2 * np.pi * 500.0
    Gam = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    d, ls, S = pb["d"], l // nranks, 24
    V = np.random.default_rng(5).standard_normal((d, l)) + 0j
    A0 = compute_moment_matrices(Lp, Gam, V, K=1, N=32, rb=0)
    fam = Lp.device()
    its0 = fam.last_info["iters_total"]
    zs, ws = gauss_points(Gam, 32)
    ct = coefficient_table(Lp, zs)
    idx, rest = snapshot_split(len(zs), S)
    idx = spread_order(idx)
    buf = torch.zeros(d * l * 2 * 2, dtype=torch.float64, device="cuda:0")
    kw = dict(K=1, tol=Lp.solver_tol, maxit=Lp.solver_maxit, out_dev=buf.data_ptr())
    slabs, parts = [], []
    for r in range(nranks):
        local = torch.empty(S * d * ls * 2, dtype=torch.float64, device="cuda:0")
        fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V[:, r * ls:(r + 1) * ls], 0, S, Q_dev=local.data_ptr(),
                            accumulate=(r > 0), l_total=l, col0=r * ls, **kw)
        slabs.append(local)
        parts.append(fam.rb_export())
    kact = parts[0][0]
    assert np.array_equal(kact, parts[1][0]) and parts[0][1].shape == (len(kact), S, S, ls)
    store = torch.stack(slabs).view(nranks, S, d, ls, 2).permute(1, 2, 0, 3, 4).contiguous()
    fam.rb_import(store.data_ptr(), kact, np.concatenate([p[1] for p in parts], axis=3), np.concatenate([p[2] for p in parts], axis=1))
    with pytest.raises(_lib.WaeError):          # an imported basis carries no probe matrix: V=None is refused, nothing is touched
        fam.beyn_moments_rb(zs[rest], ws[rest], ct[rest], None, 2, S, Q_dev=store.data_ptr(), accumulate=True, l_total=l, **kw)
    fam.beyn_moments_rb(zs[rest], ws[rest], ct[rest], V, 2, S, Q_dev=store.data_ptr(), accumulate=True, **kw)
    assert fam.last_info["n_unconverged"] == 0
    # the imported basis does its job (round 4: the projected phase runs the light V(1,0) cycle -- cheaper steps, a few more of them:
    # 0.41 of the from-zero count with the full cycle, 0.50 with the light one on this 8 736-DoF problem)
    assert fam.last_info["iters_total"] < 0.6 * its0 * len(rest) / len(zs)
    A2 = buf.cpu().numpy().view(np.complex128).reshape((d, l, 2), order="F")
    assert relerr(A2, A0) < 1e-8
    Lp._drop_device()


def test_cycle_weights_of_the_set_up_change_the_work_not_the_moments():
    """wae_solver_setup opts[10] / opts[11] (round 4): the weights of the post-smoothing sweeps (default 0.9) and of the light cycle's one
    sweep (default 0.5; the V(1,0) cycle of the projected phase of wae_beyn_moments_rb).  They are preconditioner parameters: whatever
    their values, the moments agree with the plain path to the inner tolerance; only the iteration counts move.  (The light cycle is
    what the 216 projected points of the benchmark contour run: DESIGN 4d.)"""
    import json
    import os
    from wae_amd.helmholtz.family import annulus_family
    gold = json.load(open(os.path.join(F.GOLDEN_DIR, "annulus_small_beyn.json")))
    Gam = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    res = {}
    for tag, opts in (("default", {}), ("r3", {"jacobi_weight_post": 0.8, "jacobi_weight_light": 0.8}), ("odd", {"jacobi_weight_post": 1.0, "jacobi_weight_light": 0.3})):
        Lp, pb = annulus_family("small", n=gold["n"], tau=gold["tau"])
        Lp.solver_tol = 1e-11
        Lp.solver_ref = 2 * np.pi * 500.0
        Lp.solver_opts = dict(opts)
        d = pb["d"]
        V = np.random.default_rng(gold["seed_V"]).standard_normal((d, gold["l"])) + 0j
        if tag == "default":
            A0 = compute_moment_matrices(Lp, Gam, V, K=1, N=gold["N"], rb=0)
        A1 = compute_moment_matrices(Lp, Gam, V, K=1, N=gold["N"], rb=24)
        info = dict(Lp.device().last_info)
        assert info["n_unconverged"] == 0 and relerr(A1, A0) < 1e-8, (tag, relerr(A1, A0))
        res[tag] = info["projected_iters"] if "projected_iters" in info else info["iters_total"]
        Lp._drop_device()
    assert len(set(res.values())) > 1, res                         # the options reached the solver: the work differs


def test_householder_many_matches_single_runs():
    """Several start values refined in lock-step (wae_arnoldi_shiftinvert_batch) give what the single runs give; the
    batched Arnoldi factorisation satisfies its defining relation column by column."""
    from wae_amd.nlevp import householder_many
    Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
    Lp.solver_ref = 2 * np.pi * 400.0
    starts = [2 * np.pi * 340.0, 2 * np.pi * 700.0, 2 * np.pi * 250.0]
    single = [householder(Lp, z0, maxiter=12, tol=1e-10, resident=False) for z0 in starts]      # (the host-memory iteration, one start value at a time)
    many = householder_many(Lp, starts, maxiter=12, tol=1e-10)
    s_res, n_res, f_res = householder(Lp, starts[0], maxiter=12, tol=1e-10)                          # (default: the resident form of one start value)
    assert abs(s_res.params["ω"] - single[0][0].params["ω"]) <= 1e-9 * abs(single[0][0].params["ω"]) and f_res == single[0][2]
    w1 = c(G["G1"]["omega"])
    assert abs(many[0][0].params["ω"] - w1) < 1e-9 * abs(w1)                   # tutorial_04's eigenvalue
    for (s1, n1, f1), (s2, n2, f2) in zip(single, many):
        assert abs(s1.params["ω"] - s2.params["ω"]) <= 1e-9 * abs(s1.params["ω"])
        assert f2 in (0, 1) and abs(n1 - n2) <= 1
        ov = abs(np.vdot(s1.v, s2.v)) / (np.linalg.norm(s1.v) * np.linalg.norm(s2.v))
        assert ov > 1 - 1e-8
    # Arnoldi relation of the batch: (A_s - 0 M)^-1 M V_m = V_{m+1} H, per system
    fam = Lp.device()
    d, T = Lp.size(), len(Lp.terms)
    zs = [2 * np.pi * (300 + 5j), 2 * np.pi * (650 - 3j)]
    cA = np.array([Lp.coefficients(z) for z in zs])
    cM = np.zeros(T, dtype=complex); cM[-1] = -1.0
    V0 = RNG.standard_normal((d, 2)) + 1j * RNG.standard_normal((d, 2))
    H, V = fam.arnoldi_batch(cA, cM, 4, V0, tol=1e-12)
    for s in range(2):
        MV = fam.spmv(cM, np.asfortranarray(V[s][:, :4]))
        lhs = fam.solve(cA[s], MV, tol=1e-12)
        assert relerr(lhs, V[s] @ H[s]) < 1e-7
        assert np.allclose(V[s].conj().T @ V[s], np.eye(5), atol=1e-9)
    Lp._drop_device()


def test_device_p1_assembly_matches_reference_shaped_matrices():
    """wae_p1_assemble against (i) the Rijke-tube M and K of the golden fixture -- produced by the oracle's restatement of
    `discretize` and pinned by the tutorial eigenvalues G1/G5 -- from the tutorial mesh's geometry (tests/golden/
    rijke_mesh.npz), and (ii) the numpy assembly of the synthetic annulus.  Same pattern, values to rounding."""
    import os
    from wae_amd.helmholtz.assemble import assemble_p1
    z = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    M, K = assemble_p1(z["points"], z["tetrahedra"], z["c_tet"])
    t = F.rijke_terms()
    for A, B in ((M, t["M"]), (K, t["K"])):
        A, B = sp.csr_matrix(A), sp.csr_matrix(B)
        A.sort_indices(); B.sort_indices()
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
        assert np.max(np.abs(A.data - B.data)) <= 1e-13 * np.max(np.abs(B.data))
    pb = annulus.build("small")
    pts, tets, _ = annulus._mesh(*pb["info"]["grid"])
    ctr = pts[tets].mean(axis=1)
    c_tet = np.where(ctr[:, 2] < annulus.Z_JUMP, annulus.C_COLD, annulus.C_HOT)
    M2, K2 = assemble_p1(pts, tets, c_tet)
    for A, B in ((M2, pb["terms"]["M"]), (K2, pb["terms"]["K"])):
        assert abs(A - B).max() <= 1e-13 * abs(B).max() and A.nnz == B.nnz
    # errors are reported, not swallowed
    with pytest.raises(_lib.WaeError):
        assemble_p1(pts, tets + len(pts), c_tet)


def test_projected_guesses_odd_shapes():
    """snapshot projection with shapes that do not tile the lock-step batch: l = 5 probe columns (12 systems per chunk,
    a ragged last chunk), K = 2 (four moments), 80 quadrature points, the automatic snapshot count; Rijke tube (C1)."""
    Lp = helmholtz_family(F.rijke_terms(), n=0.0)
    Lp.solver_ref = 2 * np.pi * 400.0
    Lp.solver_tol = 1e-11
    Gam = np.array([150 + 50j, 150 - 50j, 1000 - 50j, 1000 + 50j]) * 2 * np.pi
    d = Lp.size()
    V = RNG.standard_normal((d, 5)) + 1j * RNG.standard_normal((d, 5))
    A0 = compute_moment_matrices(Lp, Gam, V, K=2, N=20, rb=0)
    A1 = compute_moment_matrices(Lp, Gam, V, K=2, N=20)              # rb=None -> automatic: 40 of 80 points
    info = Lp.device().last_info
    assert A1.shape == (d, 5, 4) and info["snapshots"] == 40 and info["n_unconverged"] == 0
    for p in range(4):
        assert relerr(A1[:, :, p], A0[:, :, p]) < 1e-8
    Om = moments2eigs(A1)[0]
    f = np.sort(Om[(Om.real > 2 * np.pi * 150) & (Om.real < 2 * np.pi * 1000) & (abs(Om.imag) < 2 * np.pi * 50)].real) / 2 / np.pi
    assert any(abs(f - 272.06) < 0.5) and any(abs(f - 694.97) < 0.5)   # G7: the two passive modes of the tube
    Lp._drop_device()


def test_shape_sensitivity_matches_oracle_fixture():
    """wae_p1_shape_sensitivity against the oracle's restatement of discrete_adjoint_shape_sensitivity
    (oracle/shape.py, full re-discretisations with a direct solver; tests/golden/rijke_shape.npz): the passive 272-Hz mode
    of the Rijke tube, four points on the outlet (interior and admittance parts) and eight on the wall.  Both sides
    difference with h = 1e-9, so they agree to the rounding of that difference, not better.  Unpinned by the reference."""
    import os
    from wae_amd.helmholtz.assemble import discrete_adjoint_shape_sensitivity
    from wae_amd.nlevp import Solution
    m = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    g = np.load(os.path.join(F.GOLDEN_DIR, "rijke_shape.npz"))
    Lp = helmholtz_family(F.rijke_terms(), n=0.0, flame=False)
    w0 = complex(g["omega"][0])
    sol = Solution({**Lp.params, "ω": w0}, g["v"], g["v_adj"], "ω")
    sens = discrete_adjoint_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], g["surface_points"], sol, Lp,
                                              bnd_tris=m["outlet_triangles"], bnd_c=m["outlet_c"], Y=1e15)
    want = g["sens"]
    scale = np.abs(want).max(axis=0)                       # per point
    assert np.all(np.abs(sens - want).max(axis=0) <= 2e-5 * scale + 1e-9)
    assert np.abs(want).max() > 1.0                        # a real gradient, not noise
    Lp._drop_device()


def test_finite_difference_shape_sensitivity_confirms_the_adjoint_gradient():
    """forward_finite_differences_shape_sensitivity (src/shape_sensitivity.jl:238-337: perturb the point, re-discretise the
    adjacent simplices, re-solve the eigenvalue with householder on the device) against the discrete-adjoint gradient of
    wae_p1_shape_sensitivity -- the reference's own consistency check between its two routes.  One outlet point and one
    wall point; h = 1e-6.  The re-solved family is L + (D₊ − D₋), linear in the displacement, so its eigenvalue carries the
    second-order term of a LINEARISED perturbation, which for a tangential move (z on the tube wall: first-order shift ~0) is
    not cancelled by the operator's own second derivative: measured, the difference to the adjoint gradient is
    ≈ 1.2e5·h in that component (11.5 at h = 1e-4 against a gradient of 145 in y), 0.1 at 1e-6; the re-solve itself
    converges to ~1e-11 rad/s, i.e. 5e-6 in the quotient."""
    import os
    from wae_amd.helmholtz import shape as SH
    from wae_amd.nlevp import Solution
    m = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    g = np.load(os.path.join(F.GOLDEN_DIR, "rijke_shape.npz"))
    Lp = helmholtz_family(F.rijke_terms(), n=0.0, flame=False)
    w0 = complex(g["omega"][0])
    sol = Solution({**Lp.params, "ω": w0}, g["v"], g["v_adj"], "ω")
    pick = g["surface_points"][[0, 8]]
    adj = SH.discrete_adjoint_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], pick, sol, Lp,
                                                bnd_tris=m["outlet_triangles"], bnd_c=m["outlet_c"], Y=1e15)
    fd = SH.forward_finite_differences_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], pick, Lp, sol,
                                                         bnd_tris=m["outlet_triangles"], bnd_c=m["outlet_c"], h=1e-6)
    scale = np.abs(adj).max(axis=0)
    assert np.all(scale > 1.0)
    assert np.all(np.abs(fd - adj).max(axis=0) <= 2e-3 * scale), (fd, adj)
    Lp._drop_device()


def test_shape_sensitivity_with_an_active_flame_matches_oracle_fixture():
    """SURVEY 8 row f4, second half: the discrete-adjoint shape gradient through ALL of dscrp -- interior, admittance boundary AND
    the flame domain (shape_sensitivity.jl:62-141; Helmholtz.jl:292-344,464-487) -- for the active-flame mode of the Rijke tube
    (n = 1, tau = 1e-3: the eigenvalue of G5), against the oracle's restatement with full re-discretisations
    (tests/golden/rijke_shape_flame.npz, written by make_fixtures.py --shape-flame).  Points: 8 wall points that touch flame
    tetrahedra, the 4 vertices of the reference tetrahedron (the reference gradient itself moves), outlet and plain wall
    points.  As in the reference the flame's volume is that of the REDUCED domain at the point.  Both sides difference with
    h = 1e-9.  Unpinned by the reference (no recorded shape gradient exists)."""
    import os
    from wae_amd.helmholtz.assemble import discrete_adjoint_shape_sensitivity
    from wae_amd.nlevp import Solution
    m = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    fl = np.load(os.path.join(F.GOLDEN_DIR, "rijke_flame.npz"))
    g = np.load(os.path.join(F.GOLDEN_DIR, "rijke_shape_flame.npz"))
    Lp = helmholtz_family(F.rijke_terms(), n=1.0, tau=1e-3)
    w0 = complex(g["omega"][0])
    assert abs(w0 - c(G["G5"]["omega"])) <= 1e-6
    sol = Solution({**Lp.params, "ω": w0}, g["v"], g["v_adj"], "ω")
    flame = {"flame_tets": fl["flame_tets"], "ref_tet": int(fl["ref_tet"]), "n_ref": fl["n_ref"], "nglobal_scaled": float(fl["nglobal_scaled"])}
    sens = discrete_adjoint_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], g["surface_points"], sol, Lp,
                                              bnd_tris=m["outlet_triangles"], bnd_c=m["outlet_c"], Y=1e15, flame=flame)
    want = g["sens"]
    scale = np.abs(want).max(axis=0)
    assert np.all(np.abs(sens - want).max(axis=0) <= 2e-5 * scale + 1e-6), np.abs(sens - want).max(axis=0) / scale
    # the flame term is what is being tested: it carries most of the gradient at the points that touch the flame ...
    part = np.abs(want - g["sens_without_flame"]).max(axis=0)
    assert g["in_flame"].sum() >= 8 and np.all(part[g["in_flame"]] > 1e-3 * scale[g["in_flame"]]) and part.max() > 1e4
    # ... and nothing where the point's reduced flame domain is empty
    none = discrete_adjoint_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], g["surface_points"], sol, Lp,
                                              bnd_tris=m["outlet_triangles"], bnd_c=m["outlet_c"], Y=1e15)
    assert np.all(np.abs(none - g["sens_without_flame"]).max(axis=0) <= 2e-5 * np.abs(g["sens_without_flame"]).max(axis=0) + 1e-6)
    assert np.array_equal(np.abs(sens - none).max(axis=0) > 0, g["in_flame"])
    # the reference's own cross-check: re-solve the eigenvalue for the displaced point (two points that touch the flame)
    from wae_amd.helmholtz import shape as SH
    pick = g["surface_points"][np.nonzero(g["in_flame"])[0][:2]]
    adj = SH.discrete_adjoint_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], pick, sol, Lp, bnd_tris=m["outlet_triangles"],
                                                bnd_c=m["outlet_c"], Y=1e15, flame=flame)
    Lp.solver_ref = w0.real
    fd = SH.forward_finite_differences_shape_sensitivity(m["points"], m["tetrahedra"], m["c_tet"], pick, Lp, sol, bnd_tris=m["outlet_triangles"],
                                                         bnd_c=m["outlet_c"], h=1e-7, flame=flame)
    sc = np.abs(adj).max(axis=0)
    assert np.all(np.abs(fd - adj).max(axis=0) <= 2e-2 * sc), (fd, adj)
    Lp._drop_device()


def test_unit_cell_shape_sensitivity_matches_oracle():
    """shape sensitivity on the unit cell of a Bloch-periodic problem (shape_sensitivity.jl:27-35,84-128;
    Meshutils.jl:946-964): cylindrical displacement directions, Bloch-boundary points move with their image points, the operator
    derivative is folded by blochify.  Device route: the Cartesian kernels on the cell's extended mesh with Bloch-extended
    eigenvectors, combined along e_r, e_phi, e_z (helmholtz/shape.py).  Oracle: the reference's loop restated with full
    re-discretisations and the oracle's blochify (oracle/shape.py, oracle/bloch.py).  Small cell (12-fold ring, 128 unknowns), wave
    number b = 1, the first azimuthal mode; points: two on the reference Bloch boundary, three others on the outlet, two that touch
    the flame.  Unpinned by the reference."""
    from oracle import bloch as OB
    from oracle import helmholtz_p1 as H
    from oracle import shape as OSH
    from wae_amd.helmholtz import shape as SH
    from wae_amd.helmholtz.bloch import bloch_family
    cell = annulus.build_unit_cell(grid=(4, 8, 4), DOS=12, tau=2e-4)
    m = cell["info"]["mesh"]
    f0 = m["flames"][0]
    ns, nxb, DOS = cell["nsector"], cell["nxbloch"], cell["DOS"]
    L = bloch_family(cell, b=1)
    L.solver_ref = 2 * np.pi * 430.0
    sol, n, flag = mslp(L, 2 * np.pi * 430.0, maxiter=30, tol=1e-10)
    assert flag in (0, 1, 2) and 350 < sol.params["ω"].real / 2 / np.pi < 480
    mesh = H.Mesh()
    mesh.points = cell["points"].copy()
    mesh.tetrahedra = m["tets"].astype(np.int64)
    mesh.triangles = m["outlet_tris"].astype(np.int64)
    mesh.domains = {"Interior": {"dimension": 3, "simplices": list(range(len(m["tets"])))},
                    "Outlet": {"dimension": 2, "simplices": list(range(len(m["outlet_tris"])))},
                    "Flame": {"dimension": 3, "simplices": [int(t) for t in f0["flame_tets"]]}}
    dscrp = {"Interior": ("interior", ()), "Outlet": ("admittance", ("Y", 1e15)),
             "Flame": ("flame", (2.0, 1.0, f0["nglobal_scaled"], list(f0["x_ref"]), list(f0["n_ref"]), "n", "τ", 1.0, 2e-4))}
    surf = np.unique(mesh.triangles)
    pick = np.array(sorted(set(int(p) for p in np.concatenate([
        surf[surf < nxb][:2], np.random.default_rng(0).choice(surf[(surf >= nxb) & (surf < ns)], 3, replace=False),
        np.unique(mesh.tetrahedra[f0["flame_tets"]])[:2]]))))
    Lb = OB.bloch_family(cell["terms_ext"], ns, DOS, 0, Y=1e15, n=1.0, tau=2e-4, b=1)
    want = OSH.discrete_adjoint_shape_sensitivity_unit(mesh, dscrp, m["c_tet"], pick, Lb, sol, ns, nxb, DOS, 1)[:, pick]
    got = SH.discrete_adjoint_shape_sensitivity_unit_cell(cell, pick, sol, L, b=1)
    scale = np.abs(want).max(axis=0)
    assert np.all(scale > 1.0)
    assert np.all(np.abs(got - want).max(axis=0) <= 2e-5 * scale), np.abs(got - want).max(axis=0) / scale
    # a flame that touches the Bloch boundary (shape_sensitivity.jl:75-106 would merge the reduced flame domains of a point and its
    # image) is not supported: refused with a clear error, not computed wrongly; without the flame part the call goes through
    on_seam = int(np.nonzero(np.isin(m["tets"], np.arange(0, nxb)).any(axis=1))[0][0])
    cell_bad = dict(cell, info=dict(cell["info"], mesh=dict(m, flames=[dict(f0, flame_tets=np.append(f0["flame_tets"], on_seam))])))
    with pytest.raises(NotImplementedError, match="Bloch boundary"):
        SH.discrete_adjoint_shape_sensitivity_unit_cell(cell_bad, pick, sol, L, b=1)
    assert SH.discrete_adjoint_shape_sensitivity_unit_cell(cell_bad, pick, sol, L, b=1, flame=False).shape == (3, len(pick))
    L._drop_device()


def test_unnormalised_basis_range_guard_branch():
    """The wide-batch GMRES keeps its basis unnormalised and re-normalises a vector only when its stored norm leaves
    [1e-100, 1e100] -- a branch ordinary problems never reach.  WAE_LAZY_LIMIT=3 (read once per process, hence the child
    process) makes nearly every iteration take it; WAE_LAZY=0 is the explicit normalisation pass.  All three must give the
    same solutions and iteration counts as the default."""
    import os
    import subprocess
    import sys
    code = r'''
import json, numpy as np
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import gauss_points
L, pb = annulus_family("small", tau=2e-4)
L.solver_tol = 1e-11; L.solver_ref = 2 * np.pi * 500.0
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
zs, _ = gauss_points(G, 16)
rng = np.random.default_rng(11)
B = rng.standard_normal((pb["d"], 64)) + 1j * rng.standard_normal((pb["d"], 64))
ct = np.array([L.coefficients(z) for z in zs])
X = fam.solve(ct, B, tol=1e-11, maxit=300)
i = fam.last_info
print(json.dumps({"its": i["iters_total"], "unconv": i["n_unconverged"], "sum": [float(np.abs(X).sum()), float(np.abs(X[::7]).sum())]}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    # "host": the recurrence (Hessenberg / Givens / flags) on the host with a synchronisation per iteration, as in round 1;
    # the default keeps it on the device (gmres_wide) and looks at the status words every 4 iterations, "sync1" every iteration
    for name, extra in (("default", {}), ("guard", {"WAE_LAZY_LIMIT": "3"}), ("explicit", {"WAE_LAZY": "0"}),
                        ("host", {"WAE_GMRES_DEVICE": "0"}), ("host_guard", {"WAE_GMRES_DEVICE": "0", "WAE_LAZY_LIMIT": "3"}),
                        ("sync1", {"WAE_GMRES_SYNC": "1"}),
                        # pair steps (two Arnoldi steps per pass over the basis, round 3): off / from the first iteration / from the
                        # default iteration on -- the same Krylov space and the same per-column stopping test
                        ("single", {"WAE_GMRES_PAIR": "-1"}), ("pair0", {"WAE_GMRES_PAIR": "0"}), ("pair1_sync1", {"WAE_GMRES_PAIR": "1", "WAE_GMRES_SYNC": "1"})):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        import json
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for name in ("guard", "explicit", "host", "host_guard", "sync1", "single", "pair0", "pair1_sync1"):
        assert out[name]["unconv"] == 0 and out["default"]["unconv"] == 0
        assert abs(out[name]["its"] - out["default"]["its"]) <= 0.02 * out["default"]["its"]
        assert np.allclose(out[name]["sum"], out["default"]["sum"], rtol=1e-8)


def test_probe_columns_beyond_the_batch_width_and_strict_reporting():
    """The reference's beyn accepts any l (beyn.jl:39-57).  With a solver batch of 4 columns, l = 6 probe columns are
    handled in groups inside wae_beyn_moments (plain path) and as column slices of wae_beyn_moments_rb (projected path);
    both equal the moments computed with a wide batch.  Then: a contour integral whose inner solves cannot converge
    (2 iterations allowed) must raise instead of returning wrong moments; a single solve warns."""
    import warnings
    from wae_amd.nlevp.linopfam import UnconvergedWarning
    Gam = np.array([150 + 50j, 150 - 50j, 1000 - 50j, 1000 + 50j]) * 2 * np.pi
    d = len(F.rijke_terms()["M"].diagonal())
    V = RNG.standard_normal((d, 6)) + 1j * RNG.standard_normal((d, 6))
    ref = None
    for batch in (64, 4):
        Lp = helmholtz_family(F.rijke_terms(), n=0.3)
        Lp.solver_ref = 2 * np.pi * 400.0
        Lp.solver_tol = 1e-11
        Lp.solver_opts = {"batch": batch}
        A0 = compute_moment_matrices(Lp, Gam, V, K=1, N=16, rb=0)
        A1 = compute_moment_matrices(Lp, Gam, V, K=1, N=16, rb=20)
        assert Lp.device().last_info["n_unconverged"] == 0 and Lp.device().last_info["snapshots"] == 20
        if ref is None:
            ref = A0
        assert relerr(A0, ref) < 1e-8 and relerr(A1, ref) < 1e-8
        if batch == 4:
            Lp.solver_maxit = 2
            with pytest.raises(_lib.WaeError):
                compute_moment_matrices(Lp, Gam, V, K=1, N=4, rb=0)
            with pytest.warns(UnconvergedWarning):
                Lp(2 * np.pi * (300 + 20j)).solve(V[:, 0], maxit=2)
            Lp.device().strict = False
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                Lp(2 * np.pi * (300 + 20j)).solve(V[:, 0], maxit=2)
                compute_moment_matrices(Lp, Gam, V, K=1, N=4, rb=0)
        Lp._drop_device()


def test_saved_family_and_solution_files_drive_the_device(tmp_path):
    """SURVEY 8f-1 on the device: a family written in the reference's text layout (LinOpFam.jl:231-294) and in the
    binary container (julia/WAEHip.jl save_family_bin, 1-based CSC) is loaded with LinearOperatorFamily(fname)
    (LinOpFam.jl:196-220), uploaded, and reproduces G1 (Householder iterates, eigenvalue) and the first Taylor
    coefficients of G2; the Solution goes through save/read_sol (save.jl:2-135) and still evaluates G3's estimate."""
    from wae_amd.nlevp import read_sol, save
    L0 = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
    w = c(G["G1"]["omega"])
    sols = []
    for binary in (False, True):
        p = str(tmp_path / ("fam.waefam" if binary else "fam.toml"))
        save(p, L0, binary=binary)
        L = LinearOperatorFamily(p)
        assert len(L.terms) == 5 and L._fam is None                    # nothing on the device until it is used
        L.solver_ref = 340 * 2 * np.pi
        sol, n, flag = householder(L, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
        assert abs(sol.params["ω"] - w) < 1e-10 * abs(w) and flag in (0, 1) and 5 <= n <= 9      # (as test_G1: the reference 7 / 6 / 5)
        for mine, ref in zip(sol.history, G["G1"]["iterates"]):
            assert abs(mine - c(ref)) < 1e-6 * abs(c(ref))
        perturb_fast_(sol, L, "τ", 8)
        for k in range(9):
            assert abs(sol.eigval_pert["τ/Taylor"][k] - c(G["G2"]["taylor"][k])) < 1e-8 * abs(c(G["G2"]["taylor"][k]))
        sols.append(sol)
        L._drop_device()
    assert abs(sols[0].params["ω"] - sols[1].params["ω"]) < 1e-12 * abs(w)      # same operator from either container
    ps = str(tmp_path / "sol.toml")
    save(ps, sols[0])
    back = read_sol(ps)
    assert back.params["ω"] == sols[0].params["ω"] and np.array_equal(back.v, sols[0].v)
    assert back("τ", 0.001 + 1e-5, 8) == sols[0]("τ", 0.001 + 1e-5, 8)
    assert abs(back("τ", 0.001 + 1e-5, 8) - c(G["G3"]["omega_exact"])) < 1e-6
    L0._drop_device()


def test_device_p1_assembly_of_boundary_mass_and_flame_operators():
    """SURVEY 8f-2, second half: the admittance boundary mass C (Helmholtz.jl:443-463; FEM.jl:435-441) and the flame operator
    Q = S (x) g (Helmholtz.jl:292-344,464-487; FEM.jl:2429-2431,2442-2448) assembled on the device, against the golden
    Rijke-tube C (nnz 141) and Q (nnz 332) -- pinned through the tutorial eigenvalues G1/G5 -- from the tutorial mesh's
    geometry, and against the numpy assembly of the synthetic annulus (12 flames = 12 calls).  With all four device-assembled
    terms the family reproduces G5 (mslp, active flame)."""
    import os
    from wae_amd.helmholtz.assemble import assemble_p1, assemble_p1_boundary, assemble_p1_flame
    z = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    fl = np.load(os.path.join(F.GOLDEN_DIR, "rijke_flame.npz"))
    t = F.rijke_terms()
    C_dev = assemble_p1_boundary(z["points"], z["outlet_triangles"], z["outlet_c"])
    Q_dev, vol = assemble_p1_flame(z["points"], z["tetrahedra"], fl["flame_tets"], int(fl["ref_tet"]), fl["n_ref"], float(fl["nglobal_scaled"]))
    assert abs(vol - float(fl["volume"])) <= 1e-13 * float(fl["volume"])
    for A, B, nnz in ((C_dev, t["C"], 141), (Q_dev, t["Q"], 332)):
        A, B = sp.csr_matrix(A), sp.csr_matrix(B)
        A.sort_indices(); B.sort_indices()
        assert A.nnz == B.nnz == nnz
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
        assert np.max(np.abs(A.data - B.data)) <= 1e-13 * np.max(np.abs(B.data))
    assert np.all(C_dev.data.real == 0)                               # purely imaginary, like the reference's (Helmholtz.jl:459)
    # the whole Rijke family from the device assembly reproduces G5
    M_dev, K_dev = assemble_p1(z["points"], z["tetrahedra"], z["c_tet"])
    Lp = helmholtz_family({"M": M_dev, "K": K_dev, "C": C_dev, "Q": Q_dev}, n=1.0, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    sol, n, flag = mslp(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    assert abs(sol.params["ω"] - c(G["G5"]["omega"])) < 1e-10 * abs(c(G["G5"]["omega"]))
    Lp._drop_device()
    # synthetic annulus: outlet faces and the 12 flame slabs, found the way helmholtz/annulus.py finds them
    pb = annulus.build("small")
    nth, nz, nr = pb["info"]["grid"]
    pts, tets, _ = annulus._mesh(nth, nz, nr)
    ctr = pts[tets].mean(axis=1)
    c_tet = np.where(ctr[:, 2] < annulus.Z_JUMP, annulus.C_COLD, annulus.C_HOT)
    top = np.isclose(pts[:, 2], annulus.HEIGHT)
    faces = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    tri_nodes = tets[:, faces]
    t_idx, f_idx = np.nonzero(top[tri_nodes].all(axis=2))
    C2 = assemble_p1_boundary(pts, tri_nodes[t_idx, f_idx], c_tet[t_idx])
    assert abs(C2 - pb["terms"]["C"]).max() <= 1e-13 * abs(pb["terms"]["C"]).max() and C2.nnz == pb["terms"]["C"].nnz
    gamma, rho, Tu, Tb, P0 = 1.4, 1.225, 300.0, 1200.0, 101325.0
    nsec = annulus.N_SECTOR
    Q02U0 = P0 * (Tb / Tu - 1) * (np.pi * (annulus.R_OUT ** 2 - annulus.R_IN ** 2) / nsec) * gamma / (gamma - 1)
    ang = np.mod(np.arctan2(ctr[:, 1], ctr[:, 0]), 2 * np.pi)
    sector = np.floor(ang / (2 * np.pi / nsec)).astype(int)
    frac = ang / (2 * np.pi / nsec) - sector
    in_flame = (ctr[:, 2] > annulus.FLAME_Z0) & (ctr[:, 2] < annulus.FLAME_Z1) & (frac > 0.25) & (frac < 0.75)
    X = pts[tets]
    Jm = np.transpose(X[:, :3, :] - X[:, 3:4, :], (0, 2, 1))
    Q2 = None
    for f in range(nsec):
        sel = np.nonzero(in_flame & (sector == f))[0]
        a0 = (f + 0.5) * 2 * np.pi / nsec
        r_mid = 0.5 * (annulus.R_IN + annulus.R_OUT)
        x_ref = np.array([r_mid * np.cos(a0), r_mid * np.sin(a0), annulus.REF_Z]) + 1e-7
        ref = -1
        for it in np.nonzero(np.linalg.norm(ctr - x_ref, axis=1) < 4 * annulus.HEIGHT / nz)[0]:      # host-side point location
            xi = np.linalg.solve(Jm[it], x_ref - X[it, 3])
            xi = np.append(xi, 1 - xi.sum())
            if np.all((xi >= 0) & (xi <= 1)):
                ref = it
                break
        Qf, _ = assemble_p1_flame(pts, tets, sel, ref, [0.0, 0.0, 1.0], (gamma - 1) / rho * Q02U0)
        Q2 = Qf if Q2 is None else Q2 + Qf
    # values only: where grad(phi_b).n_ref vanishes the cofactor formula gives an exact 0 and numpy's inverse a rounding-sized
    # number, so the stored patterns differ in entries of relative size 1e-16 (pattern equality is shown on the Rijke fixture)
    Qn = sp.csr_matrix(pb["terms"]["Q"])
    assert abs(Q2 - Qn).max() <= 1e-13 * abs(Qn).max()
    assert (abs(Q2) > 1e-10 * abs(Qn).max()).nnz == (abs(Qn) > 1e-10 * abs(Qn).max()).nnz
    with pytest.raises(_lib.WaeError):
        assemble_p1_flame(pts, tets, sel, len(tets) + 3, [0.0, 0.0, 1.0], 1.0)


def test_long_rows_are_split_off_and_summed_by_a_workgroup(monkeypatch):
    """Rows with many entries (the reference nodes' rows of the TRANSPOSED flame term: one entry per flame node, ~5 000 at 1M DoF)
    leave the groups' CSR arrays and are summed by a pre-kernel (OpDev::long_*).  With the threshold lowered to 32 entries the
    small annulus has such rows in the T orientation of Q (75 entries each) and, through the dense admittance-like term
    added here, in the N orientation too: every batch width, every op, the per-term-input product and a solve agree with scipy."""
    monkeypatch.setenv("WAE_LONG_ROW", "32")
    pb = annulus.build("small")
    T = dict(pb["terms"])
    d = pb["d"]
    # a few dense-ish rows in N orientation: couple node 5 and node 77 to 200 other nodes (non-symmetric, complex)
    rng = np.random.default_rng(11)
    cols = rng.choice(d, 200, replace=False)
    E = sp.coo_matrix((rng.standard_normal(400) + 1j * rng.standard_normal(400), (np.repeat([5, 77], 200), np.tile(cols, 2))), shape=(d, d)).tocsr()
    Lp = helmholtz_family(T, n=0.8, tau=3e-4)
    Lp.push(Term(E, (pow1,), (("ω",),), "ω", "E"))
    z = 2 * np.pi * (500 + 20j)
    A = (z * z * T["M"] + T["K"] + z * 1e15 * T["C"] + 0.8 * np.exp(-1j * z * 3e-4) * T["Q"] + z * E).tocsr()
    for r in (1, 3, 8, 16, 33):
        X = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
        assert relerr(Lp(z) @ X, A @ X) < 1e-13
        assert relerr(Lp(z).H @ X, A.conj().T @ X) < 1e-13
    fam = Lp.device()
    Tn = len(Lp.terms)
    Xm = rng.standard_normal((d, Tn)) + 1j * rng.standard_normal((d, Tn))
    cs = rng.standard_normal(Tn) + 1j * rng.standard_normal(Tn)
    want = sum(cs[k] * (sp.csr_matrix(Lp.terms[k].coeff) @ Xm[:, k]) for k in range(Tn))
    assert relerr(fam.spmv_multi(cs, Xm), want) < 1e-13
    Lp.solver_ref = 2 * np.pi * 500.0
    B = rng.standard_normal((d, 4)) + 1j * rng.standard_normal((d, 4))
    import scipy.sparse.linalg as spla
    Xs = Lp(z).H.solve(B, tol=1e-12)
    assert relerr(Xs, spla.splu(A.conj().T.tocsc()).solve(B)) < 1e-7
    Lp._drop_device()


@pytest.mark.gpu
def test_coarse_level_and_restriction_tiles_agree_with_the_untiled_hierarchy(monkeypatch):
    """wae_solver_setup renumbers level 1 into tiles (4 lanes per row) and cuts the fine-to-coarse restriction into tiles as well; the
    numbering of a coarse level is invisible at the boundary.  A/B on the 20k-DoF annulus (level 1: ~2 400 unknowns, ~40 tiles;
    widths 8, 16 and 21 columns take the tile kernels, a partial last chunk included): the solutions agree with each other and
    with a sparse LU, and the iteration counts stay within two of each other.  A third configuration runs the fine level on the
    three-buffer form of the tile kernel (WAE_TILE_NBUF=3: windows of 400 rows, two gathers in flight; slower, kept as an option)."""
    import scipy.sparse.linalg as spla
    from wae_amd.helmholtz.family import annulus_family
    rng = np.random.default_rng(5)
    z = 2 * np.pi * (430 + 15j)
    sols, iters = {}, {}
    for flag in ("1", "0", "3buf"):
        monkeypatch.setenv("WAE_TILE_LEVEL1", "0" if flag == "0" else "1")
        monkeypatch.setenv("WAE_TILE_NBUF", "3" if flag == "3buf" else "2")
        L, pb = annulus_family("20k", tau=2e-4)
        T, p = pb["terms"], pb["params"]
        d = pb["d"]
        if flag == "1":
            A = (z * z * T["M"] + T["K"] + z * p["Y"] * T["C"] + p["n"] * np.exp(-1j * z * p["τ"]) * T["Q"]).tocsc()
            lu = spla.splu(A)
            Bs = {r: rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r)) for r in (8, 16, 21)}
        L.solver_ref = 2 * np.pi * 430.0
        for r, B in Bs.items():
            X = L(z).solve(B, tol=1e-12)
            assert relerr(X, lu.solve(B)) < 1e-7
            sols[(flag, r)] = X
            iters[(flag, r)] = L.device().last_info["iters_max"]
        L._drop_device()
    for r in Bs:
        for other in ("0", "3buf"):
            assert relerr(sols[("1", r)], sols[(other, r)]) < 1e-9
            # (tol = 1e-12 sits at the rounding floor of the fused products; the storage forms sum in different orders, so the last
            # digit is reached a few steps apart: 28 against 32 observed with round 4's smoother weights)
            assert abs(iters[("1", r)] - iters[(other, r)]) <= max(2, int(0.15 * iters[("1", r)]))


def test_pair_steps_of_the_narrow_recurrence():
    """tests/narrow_worker.py with the pair steps of the narrow batches forced on (WAE_NARROW_PAIR=1) and off (=0) on the 8 736-DoF
    annulus: solves against a sparse LU (op N and C, 1 / 8 columns, from zero and next to an eigenvalue with a guess direction)
    and one Newton-type refinement; both settings must pass and agree on the eigenvalue."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for pair in ("1", "0"):          # (one after the other: two processes sharing the GPU took twice as long as the two in sequence)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "narrow_worker.py")], env=dict(os.environ, WAE_NARROW_PAIR=pair),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out[pair] = json.loads(r.stdout.strip().split("\n")[-1])
        assert out[pair]["checks"] >= 20 and out[pair]["max_steps"] >= 20
    w1, w0 = complex(*out["1"]["eig"]), complex(*out["0"]["eig"])
    assert abs(w1 - w0) <= 1e-8 * abs(w0), (w1, w0)


def test_snapshot_basis_shortcuts_change_nothing_but_the_bytes():
    """The snapshot basis of the projected contour integral (wae_beyn_moments_rb) is extended with a block Gram-Schmidt update that
    reads the basis once for all new vectors, and the projected real symmetric terms take their new rows from their new columns
    (csrc/lib.hip rb_append_block).  Both off (WAE_RB_MULTI_AXPY=0, WAE_RB_HERMITIAN=0; read once per process, hence the child
    processes) must give the same moments and -- the basis only provides initial guesses -- the same iteration counts to 1 %."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import json, numpy as np
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import compute_moment_matrices
L, pb = annulus_family("small", tau=2e-4)
L.solver_tol = 1e-11; L.solver_ref = 2 * np.pi * 500.0
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
V = np.asfortranarray(np.random.default_rng(3).standard_normal((pb["d"], 8)) + 0j)
A = compute_moment_matrices(L, G, V, K=1, N=16, rb=24)
i = fam.last_info
print(json.dumps({"its": i["iters_total"], "unconv": i["n_unconverged"], "snap": i.get("snapshots"),
                  "sum": [float(np.abs(A[0]).sum()), float(np.abs(A[1][::5]).sum())]}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, extra in (("default", {}), ("plain", {"WAE_RB_MULTI_AXPY": "0", "WAE_RB_HERMITIAN": "0"})):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["default"]["unconv"] == 0 and out["plain"]["unconv"] == 0 and out["default"]["snap"] == 24
    assert abs(out["default"]["its"] - out["plain"]["its"]) <= 0.01 * out["plain"]["its"] + 2
    assert np.allclose(out["default"]["sum"], out["plain"]["sum"], rtol=1e-8)
