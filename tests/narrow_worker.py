"""Child process of tests/test_gpu_parity.py::test_pair_steps_of_the_narrow_recurrence: the two-pass Krylov recurrence of the narrow
batches (<= 8 columns: the Newton-type solvers, csrc/lib.hip gmres) takes two Arnoldi steps per reading of the basis once a basis
vector is >= 4 MB (32 768 DoF at 8 columns); WAE_NARROW_PAIR (read once per process, hence this child) forces that form onto the
8 736-DoF annulus, where a sparse LU is affordable, or switches it off.

    python tests/narrow_worker.py          (environment: WAE_NARROW_PAIR = 1 | 0)

Checks: (a) `L(z)\\b` (beyn.jl:65) for 1 and 8 shifted systems from a zero guess against scipy's sparse LU, op N and C, at a
tolerance (1e-12) that takes 40-90 steps of one recurrence, i.e. many pair steps; (b) the same close to an eigenvalue with the
eigenvector estimate as guess direction (the deflated recurrence: `solve(..., guess=)` as inverse iteration uses it); (c)
`householder` (Householder.jl:70-192) from a Beyn estimate: eigenvalue, iteration count and flag.  Prints one JSON line with the
eigenvalue so that the parent can compare the two settings."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    import wae_amd  # noqa: F401
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp import beyn, householder

    rng = np.random.default_rng(5)
    L, pb = annulus_family("small", tau=2e-4)
    d, T = pb["d"], pb["terms"]
    L.solver_tol = 1e-12
    L.solver_ref = 2 * np.pi * 500.0
    L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    fam = L.ensure_solver()
    names = list(T.keys()) if isinstance(T, dict) else None
    mats = [sp.csc_matrix(T[k]) for k in names] if names else [sp.csc_matrix(t) for t in T]

    def dense_op(c, op):
        A = sum(ck * Ak for ck, Ak in zip(c, mats)).tocsc()
        return A.conj().T.tocsc() if op == 2 else A

    nchecks, its = 0, []
    zs = 2 * np.pi * (np.array([310.0, 455.0, 520.0, 610.0, 700.0, 745.0, 820.0, 905.0]) + 1j * np.linspace(-60, 60, 8))
    for r in (1, 8):
        B = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
        ct = np.array([L.coefficients(z) for z in zs[:r]])
        Xs = {}
        for op in (0, 2):
            Xs[op] = fam.solve(ct, B, op=op, tol=1e-12, maxit=400)
            assert fam.last_info["n_unconverged"] == 0, (r, op, fam.last_info)
            its.append(int(fam.last_info["iters_max"]))
        for j in range(r):
            lu = spla.splu(dense_op(ct[j], 0))                      # ONE factorisation per system serves both orientations
            for op, trans in ((0, "N"), (2, "H")):
                ref = lu.solve(B[:, j], trans=trans)
                err = np.linalg.norm(Xs[op][:, j] - ref) / np.linalg.norm(ref)
                assert err <= 1e-8, (r, op, j, err)
                nchecks += 1
    # an eigenpair estimate from the contour integral, then the Newton-type refinement (deflated narrow solves inside)
    G = np.array([300 - 100j, 600 - 100j, 600 + 100j, 300 + 100j]) * 2 * np.pi
    Om, P = beyn(L, G, l=6, K=1, N=32)[:2]
    resb = fam.eig_residuals(np.array([L.coefficients(w_) for w_ in Om]), P=np.asfortranarray(P))
    ok = np.nonzero(resb <= 1e-4)[0]
    assert len(ok) >= 1, resb
    k = int(ok[np.argmin(np.abs(Om[ok] - np.mean(G)))])
    sol, n, flag = householder(L, Om[k], maxiter=8, tol=1e-9, v0=P[:, k], output=False)
    w = complex(sol.params[L.eigval])
    res = fam.eig_residuals(np.array([L.coefficients(w)]), P=np.asfortranarray(sol.v.reshape(d, 1)))
    assert flag in (0, 1) and n <= 4 and res[0] <= 1e-8, (flag, n, res)
    # the inverse-iteration form: a solve next to the eigenvalue with the eigenvector as guess direction
    b = rng.standard_normal(d) + 1j * rng.standard_normal(d)
    z_near = w * (1 + 1e-7)
    ctn = np.array([L.coefficients(z_near)])
    x = fam.solve(ctn, b.reshape(d, 1), tol=1e-10, maxit=400, guess=sol.v.reshape(d, 1))
    ref = spla.splu(dense_op(ctn[0], 0)).solve(b)
    cosang = abs(np.vdot(ref, x[:, 0])) / (np.linalg.norm(ref) * np.linalg.norm(x[:, 0]))
    assert 1 - cosang <= 1e-10, cosang                 # (the solution is ~1e7 x the eigenvector: compare directions and sizes)
    assert abs(np.linalg.norm(x[:, 0]) / np.linalg.norm(ref) - 1) <= 1e-4
    nchecks += 2
    print(json.dumps({"checks": nchecks, "pair": os.environ.get("WAE_NARROW_PAIR", ""), "eig": [w.real, w.imag], "newton_steps": int(n),
                      "max_steps": max(its)}))


if __name__ == "__main__":
    main()
