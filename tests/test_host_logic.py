"""CPU tests of the host-side mirror (no GPU): coefficient evaluation, functor semantics, partitions,
Beyn host tail -- compared with the oracle restatement on the same inputs."""
import numpy as np
import scipy.sparse as sp

from oracle import fixtures as F
from oracle import nlevp as ON
from oracle import solvers as OS
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import (algebra, gauss_points, moments2eigs, multi_indices_at_order, partitions, wn)


def _pair(n=0.3, tau=2e-3):
    Lo = F.rijke_family(n=n, tau=tau)
    Lp = helmholtz_family(F.rijke_terms(), n=n, tau=tau)
    return Lo, Lp


def test_coefficients_match_oracle_all_modes():
    Lo, Lp = _pair()
    z = 1700.0 + 12j
    for args in [(z,), (z, 1), (z, 2), (z, 3)]:
        co = Lo.coefficients(*args)
        cp = Lp.coefficients(*args)
        for a, b in zip(co, cp):
            assert (a is None and b == 0) or abs(a - b) <= 1e-15 * max(1, abs(a))
    for L in (Lo, Lp):
        L.active = ["ω", "τ"]; L.mode = "compact"
    for m in range(4):
        for n in range(4):
            co, cp = Lo.coefficients(m, n), Lp.coefficients(m, n)
            for a, b in zip(co, cp):
                assert (a is None and b == 0) or abs(a - b) <= 1e-14 * max(1, abs(a))
    for L in (Lo, Lp):
        L.active = ["λ", "ω"]; L.mode = "householder"; L.params["λ"] = 0.5 - 0.1j
    for m in range(3):
        for n in range(3):
            co, cp = Lo.coefficients(m, n), Lp.coefficients(m, n)
            for a, b in zip(co, cp):
                assert (a is None and b == 0) or abs(a - b) <= 1e-14 * max(1, abs(a))


def test_functor_mutates_params_like_reference():
    _, Lp = _pair()
    Lp.coefficients(123.0 + 4j)
    assert Lp.params["ω"] == 123.0 + 4j          # LinOpFam.jl:483-487
    Lp.mode = "compact"
    Lp.coefficients(1)
    assert Lp.params["ω"] == 123.0 + 4j


def test_algebra_against_oracle():
    for z in (0.3 + 0.2j, -2.0 + 1j):
        for k in range(4):
            assert algebra.pow2(z, k) == ON.pow2(z, k) and algebra.pow1(z, k) == ON.pow1(z, k)
            assert abs(algebra.pow_(z, k, 5) - ON.pow_(z, k, 5)) < 1e-13 * max(1, abs(ON.pow_(z, k, 5)))
        for m in range(4):
            for n in range(4):
                a, b = algebra.exp_delay(z * 1000, 1e-3 + 1e-4j, m, n), ON.exp_delay(z * 1000, 1e-3 + 1e-4j, m, n)
                assert abs(a - b) <= 1e-13 * max(1.0, abs(b))


def test_partitions_and_multiindices():
    for n in range(1, 9):
        assert list(partitions(n)) == list(OS.partitions(n))
    for k in range(1, 7):
        assert multi_indices_at_order(k) == OS.multi_indices_at_order(k)


def test_beyn_host_tail_and_contour():
    G = [2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j]
    z, w = gauss_points(G, 16)
    zo, wo = OS.contour_points(G, 16)
    assert np.array_equal(z, zo) and np.array_equal(w, wo)
    assert wn(0.1 + 0.1j, G) == OS.wn(0.1 + 0.1j, G) != 0 and wn(3 + 0j, G) == 0
    T = F.qep1()
    A = OS.compute_moment_matrices(T, G, OS.initial_V(3, 6), K=2, N=16)
    Om, P, S = moments2eigs(A, return_sigma=True)
    Oo, Po, So = OS.moments2eigs(A, return_sigma=True)
    assert np.allclose(np.sort_complex(Om), np.sort_complex(Oo)) and np.allclose(S, So)


def test_pade_estimate_pol_and_vector_pade_match_oracle():
    from wae_amd.nlevp import Solution, conv_radius, estimate_pol, pade, pade_
    rng = np.random.default_rng(5)
    w = rng.standard_normal(9) + 1j * rng.standard_normal(9)
    for Lp, Mp in ((4, 4), (3, 2), (8, 0), (0, 3)):
        a, b = pade(w, Lp, Mp)
        ao, bo = ON.pade(w, Lp, Mp)
        assert np.allclose(a, ao) and np.allclose(b, bo)
    de, k = estimate_pol(w)
    deo, ko = ON.estimate_pol(w)
    assert np.allclose(de, deo) and np.allclose(k, ko)
    assert np.allclose(conv_radius(w), ON.conv_radius(w))
    # series of f(e) = 1/(1-2e) * v:  Pade [1/1] reproduces it exactly, component-wise
    v = rng.standard_normal(5) + 0j
    sol = Solution({"ω": 1.0 + 0j, "τ": 0.5 + 0j}, v, v, "ω")
    sol.eigval_pert["τ/Taylor"] = np.array([2.0 ** n for n in range(5)], dtype=complex)
    sol.v_pert["τ/Taylor"] = [v * 2.0 ** n for n in range(5)]
    val, vec = sol("τ", 0.5 + 0.1, 1, 1, vector=True)
    assert abs(val - 1 / (1 - 0.2)) < 1e-12 and np.allclose(vec, v / (1 - 0.2))
    so = ON.Solution({"ω": 1.0 + 0j, "τ": 0.5 + 0j}, v, v, "ω")
    so.v_pert["τ/Taylor"] = sol.v_pert["τ/Taylor"]
    A, B = ON.pade_vector(so, "τ", 1, 1)
    Ap, Bp = sol.v_pert["τ/[1/1]"]
    assert np.allclose(A, Ap) and np.allclose(B, Bp)
