"""Pin the CPU oracle against the reference's recorded tutorial outputs.

The reference is Julia and has no test-suite; these executed-tutorial values are the only
known-answer data it ships.  Tolerances: the oracle uses SuperLU/ARPACK(scipy) where the reference
uses UMFPACK/ARPACK(Julia); iterates agree to ~1e-10 absolute, Taylor coefficients to ~1e-12 relative.
"""
import numpy as np
import pytest

from oracle import fixtures as F
from oracle import solvers as S
from oracle.nlevp import conv_radius

G = F.golden()
c = lambda p: complex(p[0], p[1])


def test_G9_mesh_stats_and_terms():
    z = np.load(F.GOLDEN_DIR + "/rijke_p1.npz")
    assert (int(z["npoints"]), int(z["ntriangles"]), int(z["ntetrahedra"])) == (
        G["G9"]["points"], G["G9"]["triangles"], G["G9"]["tetrahedra"])
    t = F.rijke_terms()
    # nnz(M)=nnz(K)=11338, nnz(C)=141, nnz(Q)=332
    assert (t["M"].nnz, t["K"].nnz, t["C"].nnz, t["Q"].nnz) == (11338, 11338, 141, 332)
    assert abs(t["M"] - t["M"].T).max() < 1e-20 and abs(t["K"] - t["K"].T).max() < 1e-9


@pytest.fixture(scope="module")
def g1():
    L = F.rijke_family(n=0.01, tau=0.001)
    sol, n, flag = S.householder(L, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    return L, sol, n, flag


def test_G1_householder(g1):
    L, sol, n, flag = g1
    assert abs(sol.params["ω"] - c(G["G1"]["omega"])) < 1e-9
    assert flag in (0, 1) and 5 <= n <= 8
    for mine, ref in zip(sol.history, G["G1"]["iterates"]):
        assert abs(mine - c(ref)) < 1e-6 * abs(c(ref))
    # normalisations of Householder.jl:189-190
    M = -L.terms[-1].coeff
    assert abs(np.vdot(sol.v, M @ sol.v) - 1) < 1e-10
    assert abs(np.vdot(sol.v_adj, L(sol.params["ω"], 1) @ sol.v) - 1) < 1e-8


def test_G2_G3_G4_perturb_fast(g1):
    L, sol, _, _ = g1
    S.perturb_fast_(sol, L, "τ", 30)
    lam = sol.eigval_pert["τ/Taylor"]
    for k, ref in enumerate(G["G2"]["taylor"]):
        assert abs(lam[k] - c(ref)) < 2e-11 * abs(c(ref)), k
    sol.eigval_pert["τ/Taylor"] = lam[:21]
    est20 = sol("τ", 0.001 + 1e-5, 20)
    assert abs(est20 - c(G["G3"]["taylor20_estimate"])) < 1e-9
    # (the notebook's "first-order approx" printout, cell 19, is inconsistent with its own cell-14 state --
    #  ω0+ω1·1e-5 = 272.2925 Hz, not 272.4516 Hz -- a stale execution; it is recorded in golden.json but not a pin)
    # perturb! (on-the-fly partitions) gives the same eigenvalue coefficients
    L2 = F.rijke_family(n=0.01, tau=0.001)
    sol2, _, _ = S.householder(L2, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    S.perturb_(sol2, L2, "τ", 6)
    for k in range(1, 7):
        assert abs(sol2.eigval_pert["τ/Taylor"][k] - lam[k]) < 1e-8 * abs(lam[k])


def test_G3_exact_shifted_tau():
    L = F.rijke_family(n=0.01, tau=0.001 + 1e-5)
    sol, n, flag = S.householder(L, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    assert abs(sol.params["ω"] - c(G["G3"]["omega_exact"])) < 1e-9
    L = F.rijke_family(n=0.01, tau=G["G4b"]["tau"])
    sol, n, flag = S.householder(L, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    assert abs(sol.params["ω"] - c(G["G4b"]["omega_n0.01"])) < 1e-9


def test_G5_G6_mslp_active_flame():
    L = F.rijke_family(n=1.0, tau=0.001)
    sol, n, flag = S.mslp(L, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
    assert abs(sol.params["ω"] - c(G["G5"]["omega"])) < 1e-9
    assert flag == 0 and n == G["G5"]["iterations"]
    S.perturb_fast_(sol, L, "τ", 30)
    lam30 = sol.eigval_pert["τ/Taylor"]
    r = conv_radius(lam30)                                   # G4: full 30-entry table of the md tutorial
    assert np.allclose(r, G["G4"]["conv_radius"], rtol=1e-8, atol=0)
    assert abs(sol("τ", 0.0015, 30) / 2 / np.pi - c(G["G4"]["taylor30_estimate_over_2pi_at_tau_plus_5e-4"])) < 1e-7
    lam = sol.eigval_pert["τ/Taylor"] = lam30[:21]
    for k, ref in enumerate(G["G6"]["taylor_6digits"]):
        assert abs(lam[k] - c(ref)) < 2e-5 * abs(c(ref))
    assert abs(sol("τ", 0.0015, 20) - c(G["G6"]["taylor20_estimate"])) < 1e-7
    L.params["τ"] = 0.0015
    sol2, _, _ = S.mslp(L, sol("τ", 0.0015, 20), maxiter=20, tol=1e-11)
    assert abs(sol2.params["ω"] - c(G["G6"]["omega_exact"])) < 1e-9


def test_G7_beyn_rijke_passive():
    """G7 is a qualitative pin only ("two eigenmodes oscillating at 272 and 695 Hz", output not recorded):
    Gauss-Legendre on the thin rectangle leaves the Beyn estimates a few Hz off, so compare loosely and
    check that exactly two singular values are significant.  Parity of Beyn on FEM problems is otherwise unpinned."""
    L = F.rijke_family(n=0.0)
    Gam = np.array([150 + 5j, 150 - 5j, 1000 - 5j, 1000 + 5j]) * 2 * np.pi
    Om, P, Sig = S.beyn(L, Gam, l=5, N=256, do_pos_test=False, return_sigma=True)
    assert np.sum(Sig > 1e-8 * Sig[0]) == 2
    f = Om / 2 / np.pi
    for hz in G["G7"]["modes_hz"]:
        assert np.min(np.abs(f - hz)) < 9.0
    # local refinement of the estimates lands on the two real modes
    for hz, exact in zip(G["G7"]["modes_hz"], (272.0643, 694.9677)):
        sol, n, flag = S.mslp(L, hz * 2 * np.pi, maxiter=20, tol=1e-9)
        assert abs(sol.params["ω"] / 2 / np.pi - exact) < 1e-2


def test_G8_qep1_beyn_mslp_count():
    T = F.qep1()
    Gam = [2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j]
    Om, P, Sig = S.beyn(T, Gam, l=6, return_sigma=True)
    assert np.sum(Sig < 1e-10 * Sig[0]) == G["G8"]["n_small_sigma"]
    want = [c(e) for e in G["G8"]["eigs_inside"]]
    for w in want:
        assert np.min(np.abs(Om - w)) < 1e-9
    n = S.count_poles_and_zeros(T, Gam)
    assert abs(n - G["G8"]["count_poles_and_zeros"]) < 1e-2
    sol, it, flag = S.mslp(T, 0, tol=1e-10)
    assert abs(sol.params["λ"] - c(G["G8"]["mslp_from_0_tol1e-10"]["omega"])) < 1e-9
    assert flag == 0 and abs(it - G["G8"]["mslp_from_0_tol1e-10"]["iterations"]) <= 1


def test_newton_variants_agree_with_G1():
    L = F.rijke_family(n=0.01, tau=0.001)
    w = c(G["G1"]["omega"])
    sol, n, flag = S.inveriter(L, 1710 + 9j, maxiter=20, tol=1e-10)
    assert flag == 0 and abs(sol.params["ω"] - w) < 1e-8
    x = sol.v
    sol, n, flag = S.rf2s(L, 1710 + 9j, maxiter=20, tol=1e-10, x0=x, y0=np.conj(x))
    assert flag == 0 and abs(sol.params["ω"] - w) < 1e-8
    sol, n, flag = S.lancaster(L, 1710 + 9j, maxiter=20, tol=1e-10)
    assert abs(L.params["ω"] - w) < 1e-6 or flag != 0
