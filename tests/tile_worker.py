"""Child process of tests/test_gpu_tile_parity.py: the tile kernel's work list has two regimes, chosen by the number of tiles
against the number of compute units (csrc/kernels.hip, launch_spmv_tile): with fewer tiles than CUs every tile is cut into parts;
otherwise the workgroups are persistent -- a static first tile, then tiles drawn from per-XCD counters, stolen from other
shares once the own one is used up, the last tiles handed out in parts.  On a 256-CU device only the benchmark sizes reach the
second regime; WAE_TILE_GRID (read once per process, hence this child) makes the kernel believe in 8 CUs so that the 8 736-DoF
annulus (35 fine tiles, ~16 level-1 tiles) runs it at a size the oracle can afford.

    python tests/tile_worker.py            (environment: WAE_TILE_GRID, WAE_TILE_TAIL)

Checks, all against scipy (the reference's `L(z)*x`, LinOpFam.jl:482-529, and `L(z)\\b`, beyn.jl:65):
  * every fused form of the product (A X, residual, Jacobi sweep, B + A X, the row-scaled forms, product + first sweep) at widths
    8 ... 100 (partial last chunks included), one system per launch and one system per column (the two instantiations of the
    kernel), op N and C, with and without a converged-chunk mask;
  * level 1 and the restriction through the 4-lane tile kernel against the plain CSR kernels of the same hierarchy;
  * a lock-step solve of 64 shifted systems against a sparse LU.
Prints one JSON line; exit code 0 = all checks passed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import scipy.sparse.linalg as spla

    import wae_amd  # noqa: F401
    from _tilecheck import TermProducts, assert_close, check_modes
    from wae_amd.helmholtz.family import annulus_family

    rng = np.random.default_rng(11)
    L, pb = annulus_family("small", tau=2e-4)
    d, T = pb["d"], pb["terms"]
    L.solver_tol = 1e-12
    L.solver_ref = 2 * np.pi * 500.0
    L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    fam = L.ensure_solver()
    zs = 2 * np.pi * (np.linspace(160, 990, 100) + 1j * np.linspace(-140, 140, 100))
    nchecks = 0
    for r in (8, 11, 16, 33, 64, 100):
        X = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
        tpN, tpC = TermProducts(T, X, "N"), TermProducts(T, X, "C")
        ct1 = np.array([L.coefficients(zs[3])])
        ctr = np.array([L.coefficients(z) for z in zs[:r]])
        nch = (r + 7) // 8
        mask = np.ones(nch, dtype=bool)
        mask[rng.integers(0, nch, size=max(1, nch // 3))] = False            # some chunks converged ...
        if nch > 2:
            mask[0] = False                                                   # ... the first one among them
        if not mask.any():
            mask[-1] = True
        for ct, name in ((ct1, "one system"), (ctr, "one system per column")):
            check_modes(fam, tpN, ct, X, rng, f"r={r} {name} op N")
            check_modes(fam, tpN, ct, X, rng, f"r={r} {name} op N masked", cmask=mask, modes=(0, 1, 2, 6))
            check_modes(fam, tpC, ct, X, rng, f"r={r} {name} op C", op=2, modes=(0, 1, 2))
            nchecks += 14
            # the public entries take the same path
            want, bound, _ = tpN.apply(ct)
            assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], X), want, bound, f"r={r} {name} wae_spmv_sum(_cols)")
            nchecks += 1
    # level 1 / restriction: 4 lanes per row, against the CSR kernels on the same (renumbered) hierarchy
    sizes = {(w, lv): (ni, no) for w, lv, ni, no in fam.level_sizes()}
    for r in (8, 20, 64):
        for which, lv in ((0, 1), (1, 0)):
            if (which, lv) not in sizes:
                continue
            ni, no = sizes[(which, lv)]
            X = rng.standard_normal((ni, r)) + 1j * rng.standard_normal((ni, r))
            ct = np.array([L.coefficients(zs[5])])
            Bm = rng.standard_normal((no, r)) + 1j * rng.standard_normal((no, r))
            for mode in ((0, 1, 2, 6) if which == 0 else (0,)):
                a = fam.debug_spmv(ct, X, mode=mode, B=None if mode in (0, 6) else Bm, level=lv, which=which)
                b = fam.debug_spmv(ct, X, mode=mode, B=None if mode in (0, 6) else Bm, level=lv, which=which, no_tiles=True)
                for u, v in zip(a if mode == 6 else (a,), b if mode == 6 else (b,)):
                    assert np.max(np.abs(u - v)) <= 1e-12 * np.max(np.abs(v)), (which, lv, mode, r)
                nchecks += 1
    # lock-step solve of 64 shifted systems through the whole hierarchy
    B = rng.standard_normal((d, 64)) + 1j * rng.standard_normal((d, 64))
    ct = np.array([L.coefficients(z) for z in zs[:64]])
    Xs = fam.solve(ct, B, tol=1e-12, maxit=300)
    assert fam.last_info["n_unconverged"] == 0
    p = pb["params"]
    for j in (0, 21, 63):
        z = zs[j]
        A = (z * z * T["M"] + T["K"] + z * p["Y"] * T["C"] + p["n"] * np.exp(-1j * z * p["τ"]) * T["Q"]).tocsc()
        ref = spla.splu(A).solve(B[:, j])
        assert np.max(np.abs(Xs[:, j] - ref)) <= 1e-8 * np.max(np.abs(ref)), j
        nchecks += 1
    L._drop_device()
    print(json.dumps({"checks": nchecks, "grid": os.environ.get("WAE_TILE_GRID"), "tail": os.environ.get("WAE_TILE_TAIL")}))


if __name__ == "__main__":
    main()
