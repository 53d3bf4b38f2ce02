"""CPU tests of the host-side mirror (no GPU): coefficient evaluation, functor semantics, partitions,
Beyn host tail -- compared with the oracle restatement on the same inputs."""
import numpy as np
import pytest
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


def test_gram_svd_of_a_rank_deficient_tall_matrix_matches_lapack():
    """`_svd_by_gram` (the factorisation of the Hankel matrix B0 of beyn.jl:289-323 on the device path): a tall matrix with a
    group of 7 singular values of order one and 9 more ten orders below (what the moments of a contour holding 7 eigenvalues look
    like).  Both groups to 1e-6 relative (the lower one is only ever used as the rank gap), the kept triplets reproduce the matrix
    to the noise level, the basis is orthonormal to rounding -- and the tall-skinny product helper against a plain product."""
    import torch
    from wae_amd.nlevp.distributed import _svd_by_gram, _tall_gram, moments2eigs_device
    rng = np.random.default_rng(17)
    d, n, k = 5000, 16, 7
    Q1, _ = np.linalg.qr(rng.standard_normal((d, n)) + 1j * rng.standard_normal((d, n)))
    Q2, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    s = np.concatenate([np.linspace(3.0, 0.2, k), 1e-10 * np.linspace(2.0, 0.1, n - k)])
    B0 = (Q1 * s) @ Q2.conj().T
    t = torch.from_numpy(B0)
    assert np.allclose(_tall_gram(t, t).numpy(), B0.conj().T @ B0, atol=1e-13)
    assert np.allclose(_tall_gram(t[:700], t[:700]).numpy(), B0[:700].conj().T @ B0[:700], atol=1e-13)      # ragged last block
    U, S, Wh, Sall = _svd_by_gram(t, 1e-6)
    assert U.shape == (d, k) and S.shape == (k,) and Wh.shape == (k, n)
    assert np.allclose(S.numpy(), s[:k], rtol=1e-12)
    assert np.allclose(Sall.numpy()[:k], s[:k], rtol=1e-12) and np.allclose(Sall.numpy()[k:], s[k:], rtol=1e-6)
    Un = U.numpy()
    assert np.linalg.norm(Un.conj().T @ Un - np.eye(k)) < 1e-13
    assert np.linalg.norm((Un * S.numpy()) @ Wh.numpy() - B0) < 1e-9
    _, s_l, _ = np.linalg.svd(B0, full_matrices=False)
    assert np.allclose(Sall.numpy()[:k], s_l[:k], rtol=1e-12)
    # a threshold below what one Gram matrix resolves (1e-8 of the largest): the second group is taken by a second stage
    U2, S2, Wh2, Sall2 = _svd_by_gram(t, 1e-12)
    assert U2.shape == (d, n) and np.allclose(S2.numpy(), s, rtol=1e-6) and len(Sall2) == n
    assert np.linalg.norm(U2.numpy().conj().T @ U2.numpy() - np.eye(n)) < 1e-12
    assert np.linalg.norm((U2.numpy() * S2.numpy()) @ Wh2.numpy() - B0) < 1e-14
    with pytest.raises(ValueError):
        _svd_by_gram(torch.zeros((50, 4), dtype=torch.complex128), 1e-6)
    # the whole host tail through both factorisations: K = 2 moments of a diagonal problem, eigenvalues agree
    lam = np.array([0.3 + 0.1j, -0.2 + 0.4j, 0.1 - 0.5j])
    dd, l = 600, 4
    V = rng.standard_normal((dd, 3)) + 1j * rng.standard_normal((dd, 3))
    W = rng.standard_normal((3, l)) + 1j * rng.standard_normal((3, l))
    A = np.stack([(V * lam ** p) @ W for p in range(4)], axis=2)                         # (d, l, 2K), A_p = V Λ^p W
    buf = torch.from_numpy(np.ascontiguousarray(A.transpose(2, 1, 0)).view(np.float64).reshape(-1).copy())
    Om_q, _, S_q = moments2eigs_device(buf, (dd, l, 4), tol_sigma=1e-8)
    Om_g, P_g, S_g = moments2eigs_device(buf, (dd, l, 4), gram_rel_tol=1e-8)
    assert len(Om_g) == 3 and np.allclose(np.sort_complex(Om_g), np.sort_complex(lam), atol=1e-10)
    assert np.allclose(np.sort_complex(Om_q), np.sort_complex(lam), atol=1e-10)
    assert np.allclose(S_g[:3], S_q[:3], rtol=1e-10) and P_g.shape == (dd, 3)


def test_conjugate_span_start_of_the_left_processes():
    """`householder_many` without adjoint start vectors: conj(v) for an isolated mode (what `householder` starts from,
    Householder.jl:84-86), the bi-orthogonal combination for a spinning pair, whose members have v^T v = 0."""
    from wae_amd.nlevp.local_solvers import _conjugate_span_start
    n = 400
    phi = np.linspace(0, 2 * np.pi, n, endpoint=False)
    spin_p, spin_m = np.exp(3j * phi), np.exp(-3j * phi)                # a degenerate pair: each is the other's conjugate
    lone = np.cos(5 * phi) * (1 + 0.2j)
    V = np.stack([spin_p, lone, spin_m], axis=1)
    W = _conjugate_span_start(V)
    B = W.conj().T @ V                                                   # W^H V: diagonal = bi-orthogonal
    assert np.allclose(B - np.diag(np.diag(B)), 0, atol=1e-10) and np.all(np.abs(np.diag(B)) > 0.5)
    # the plain conjugates are orthogonal to the spinning modes they are supposed to pair with
    assert abs(np.vdot(np.conj(spin_p), spin_p)) < 1e-9 * n
    assert np.allclose(_conjugate_span_start(V[:, 1:2]), np.conj(V[:, 1:2]))       # one vector: the reference's start
    # numerically dependent start vectors: falls back to the plain conjugates
    Vd = np.stack([spin_p, spin_p * (1 + 1e-13)], axis=1)
    assert np.allclose(_conjugate_span_start(Vd), np.conj(Vd))


def test_householder_many_of_an_empty_start_list_is_an_empty_list():
    """the batched refinement with nothing to refine (a Beyn pass that found no eigenvalue inside): no device, no library call"""
    from wae_amd.nlevp import householder_many
    L = helmholtz_family(F.rijke_terms(), n=0.5)
    stats = {}
    assert householder_many(L, [], stats=stats) == []
    assert stats["newton_rounds"] == 0 and stats["inner_column_iterations"] == 0
    assert L._fam is None                                    # the family was not even uploaded
