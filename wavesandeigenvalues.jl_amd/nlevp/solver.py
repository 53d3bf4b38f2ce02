"""solve(L, Γ; ...): incremental Beyn + analytic deflation + local refinement (reference: src/NLEVP/solver.jl:36-184).

As shipped the reference calls an un-included ``mehrmann`` (solver.jl:106, NLEVP.jl:17) and throws; this mirror
implements the documented intent with the local solver the reference's own comment proposes (solver.jl:105,
``householder(..., order=3, nev=nev, v0=v0)``), which returns the normalised adjoint vector the deflation formula
needs (residue of L(z)^{-1} at ω is v·v_adj' when v_adj' L'(ω) v = 1, Householder.jl:189-190).
"""
from __future__ import annotations

import numpy as np

from .beyn import compute_moment_matrices, inpoly, moments2eigs, wn
from .local_solvers import householder


def solve(L, G, dl=1, N=16, tol=1e-8, eigvals=None, maxcycles=1, nev=1, max_outer_cycles=1, atol_sigma=1e-12,
          rtol_sigma=1e-8, loglevel=0, order=1):
    """Returns dict ω -> [Solution, inside::bool]   (solver.jl:36-184; Δl -> dl)."""
    eigvals = {} if eigvals is None else eigvals
    d = L.size()
    A = []
    l = dl
    s0 = s = smax = 0.0
    while l <= max_outer_cycles * dl:
        V = np.zeros((d, dl), dtype=np.complex128)
        for ll in range(dl):
            V[(l - dl) + ll, ll] = 1.0                                  # solver.jl:43-47
        A.append(compute_moment_matrices(L, G, V, K=1, N=N))
        if l > dl:
            _, _, Sig = moments2eigs(A, return_sigma=True)
            smax, s0, s = max(smax, Sig.max()), Sig.max(), 0.0
        for om, val in eigvals.items():                                 # deflate known pairs, solver.jl:57-64
            w = wn(om, G)
            for ll in range(dl):
                moment = -2j * np.pi * w * val[0].v * np.conj(val[0].v_adj[(l - dl) + ll])
                A[-1][:, ll, 0] += moment
                A[-1][:, ll, 1] += om * moment
        neig = sum(1 for v in eigvals.values() if v[1])
        cycle = 0
        while cycle < maxcycles:
            cycle += 1
            Om, P, Sig = moments2eigs(A, return_sigma=True)
            smax, s0, s = max(smax, s), s, Sig.max()
            for idx in range(len(Om)):
                om = Om[idx]
                v0 = P[:, idx] / np.linalg.norm(P[:, idx])
                for val in eigvals.values():                            # solver.jl:94-100
                    v = val[0].v / np.linalg.norm(val[0].v)
                    v0 = v0 - np.vdot(v, v0) * v
                    v0 = v0 / np.linalg.norm(v0)
                sol, nn, flag = householder(L, om, maxiter=10, tol=tol, order=order, nev=nev, v0=v0)
                om = sol.params[sol.eigval]
                is_new = flag >= 0 and all(abs(om - k) >= 10 * tol for k in eigvals)
                if loglevel >= 2:
                    print(f"conv:{om} flag:{flag} new:{is_new}")
                if is_new and inpoly(om, G):
                    w = wn(om, G)
                    for ia, Ai in enumerate(A):
                        for ll in range(dl):
                            moment = -2j * np.pi * w * sol.v * np.conj(sol.v_adj[ia * dl + ll])
                            Ai[:, ll, 0] += moment
                            Ai[:, ll, 1] += om * moment
                    eigvals[om] = [sol, True]
                elif is_new:
                    eigvals[om] = [sol, False]
            n_in = sum(1 for v in eigvals.values() if v[1])
            if n_in == neig:
                break
            neig = n_in
        if loglevel >= 1:
            print(f"maximum σmax:{smax}\nfinal σ:{s}\ncycles:{cycle}")
        if smax > 0 and (s / smax < rtol_sigma or s < atol_sigma):
            break
        l += dl
    return eigvals
