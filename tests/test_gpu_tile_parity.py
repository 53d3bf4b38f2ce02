"""GPU parity of the dominant kernel -- the tile form of the fused multi-term operator product `L(z)*X` (LinOpFam.jl:482-529) --
in the regime the benchmark runs it in, against scipy.

`launch_spmv_tile` (csrc/kernels.hip) has two work-list regimes.  Below one tile per compute unit every tile is cut into parts
(what every small parity case exercises); from 256 tiles on (C2: 780, C3: 3 888 on the fine level, 1 828 on level 1) the
workgroups are persistent, draw tiles from per-XCD counters, steal, and share out the last tiles.  Here the second regime is
checked (a) at BASELINE's full sizes C2 and C3 against scipy products of the caller's own term matrices -- one system per launch
(`spmv_tile_kernel<true,2,2>`) and one system per column (`<false,2,2>`, the instantiation a multi-GPU rank and the Newton-type
solvers run), op N and C, every fused form the solver uses, with a converged-chunk mask -- and (b) in a child process that
forces the regime onto an 8 736-DoF problem (WAE_TILE_GRID=8) with both tail settings, where a sparse LU is affordable too.
Tolerances: tests/_tilecheck.py (1e-13 per column in the max-norm and element-wise against the row's rounding bound)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from _tilecheck import FamilyProducts, TermProducts, assert_close, check_modes
from wae_amd.helmholtz.family import annulus_family

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("preset,d_expect", [("C2", 199680), ("C3", 995328)])
def test_tile_kernel_at_benchmark_size_against_scipy(preset, d_expect, request):
    rng = np.random.default_rng(21)
    big = preset == "C3"              # (the 1M-DoF case shares its family with test_c3_one_million_dof_pass and runs the shorter list of
    if big:                           # forms: every form and both orientations at r = 64 are covered at C2 in the same work-list regime)
        L, pb = request.getfixturevalue("c3_family")
    else:
        L, pb = annulus_family(preset, tau=2e-4)
        L.solver_ref = 2 * np.pi * 500.0
        L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    d, T = pb["d"], pb["terms"]
    assert d == d_expect
    L.solver_tol = 1e-10
    fam = L.device()
    zs = 2 * np.pi * (np.linspace(155, 995, 64) + 1j * np.linspace(-145, 145, 64))
    ct1 = np.array([L.coefficients(zs[40])])
    ct64 = np.array([L.coefficients(z) for z in zs])
    X = rng.standard_normal((d, 64)) + 1j * rng.standard_normal((d, 64))
    tpN = TermProducts(T, X, "N")
    mask = np.array([1, 0, 1, 1, 0, 0, 1, 1], dtype=bool)
    # r = 64: one system per launch, one system per column; every fused form; masked chunks keep their contents
    check_modes(fam, tpN, ct1, X, rng, f"{preset} r=64 one system", modes=(0, 1, 2, 6) if big else (0, 1, 2, 3, 4, 5, 6))
    check_modes(fam, tpN, ct64, X, rng, f"{preset} r=64 one system per column", modes=(0, 2) if big else (0, 1, 2, 3, 4, 5, 6))
    check_modes(fam, tpN, ct1, X, rng, f"{preset} r=64 one system, masked", cmask=mask, modes=(1, 6) if big else (0, 1, 2, 6))
    if not big:
        check_modes(fam, tpN, ct64, X, rng, f"{preset} r=64 one system per column, masked", cmask=mask, modes=(0, 2))
    for ct in (ct1, ct64):                                          # the public entries (wae_spmv_sum / wae_spmv_sum_cols)
        want, bound, _ = tpN.apply(ct)
        assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], X), want, bound, f"{preset} wae_spmv_sum r=64")
    if not big:                                                     # (op C at 1M DoF: the r = 8 check below)
        tpC = TermProducts(T, X, "C")
        for ct in (ct1, ct64):
            want, bound, _ = tpC.apply(ct)
            assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], X, op=2), want, bound, f"{preset} wae_spmv_sum r=64 op C")
        del tpC
    # r = 8 (the width of the first snapshot chunks): columns 8..15 of the same X
    X8 = np.ascontiguousarray(X[:, 8:16])
    tp8 = TermProducts(T, X8, "N")
    check_modes(fam, tp8, ct1, X8, rng, f"{preset} r=8 one system", modes=(0, 1, 2, 6))
    check_modes(fam, tp8, ct64[:8], X8, rng, f"{preset} r=8 one system per column", modes=(0, 2))
    tp8c = TermProducts(T, X8, "C")
    want, bound, _ = tp8c.apply(ct1)
    assert_close(fam.spmv(ct1[0], X8, op=2), want, bound, f"{preset} wae_spmv_sum r=8 op C")
    # fewer than 8 columns on a large operator (the Newton-type solvers' batches: general CSR kernel, teams of 8 lanes per row -- 8 / C
    # lanes per row at C column lanes), both orientations, one system and one system per column
    if preset == "C2":
        for r in (4, 3, 2, 1):
            Xr = np.ascontiguousarray(X[:, 16:16 + r])
            for opn, opi in (("N", 0), ("C", 2)):
                tpr = TermProducts(T, Xr, opn)
                for ct in (ct1, ct64[:r]):
                    want, bound, _ = tpr.apply(ct)
                    assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], Xr, op=opi), want, bound, f"C2 wae_spmv_sum r={r} op {opn}")
    # more than 256 columns through the public entry (column groups inside the library)
    if preset == "C2":
        X300 = rng.standard_normal((d, 300)) + 1j * rng.standard_normal((d, 300))
        tp300 = TermProducts(T, X300, "N")
        want, bound, _ = tp300.apply(ct1)
        assert_close(fam.spmv(ct1[0], X300), want, bound, "C2 wae_spmv_sum r=300")
        del tp300, X300
    # level 1 and the restriction (4 lanes per row, 1 828 / 4 804 tiles at C3) against the CSR kernels of the same hierarchy
    fam = L.ensure_solver()
    sizes = {(w, lv): (ni, no) for w, lv, ni, no in fam.level_sizes()}
    for which, lv in ((0, 1), (1, 0)):
        ni, no = sizes[(which, lv)]
        Xl = rng.standard_normal((ni, 64)) + 1j * rng.standard_normal((ni, 64))
        Bl = rng.standard_normal((no, 64)) + 1j * rng.standard_normal((no, 64))
        for mode in ((0, 1, 2, 6) if which == 0 else (0,)):
            a = fam.debug_spmv(ct1, Xl, mode=mode, B=None if mode in (0, 6) else Bl, level=lv, which=which)
            b = fam.debug_spmv(ct1, Xl, mode=mode, B=None if mode in (0, 6) else Bl, level=lv, which=which, no_tiles=True)
            for u, v in zip(a if mode == 6 else (a,), b if mode == 6 else (b,)):
                assert np.max(np.abs(u - v)) <= 1e-12 * np.max(np.abs(v)), (which, lv, mode)
    # the prolongation by fine tile (round 4: the coarse rows a tile talks to staged in LDS) against the gather kernel it replaces:
    # Y = B + P X, also with a converged-chunk mask; and the restriction once more under a mask
    nf, nc = sizes[(1, 0)]
    Xc = rng.standard_normal((nc, 64)) + 1j * rng.standard_normal((nc, 64))
    Bf = rng.standard_normal((nf, 64)) + 1j * rng.standard_normal((nf, 64))
    for cm in (None, mask):
        a = fam.debug_spmv(ct1, Xc, mode=3, B=Bf, Y0=Bf, level=0, which=2, cmask=cm)
        b = fam.debug_spmv(ct1, Xc, mode=3, B=Bf, Y0=Bf, level=0, which=2, cmask=cm, no_tiles=True)
        assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b)) and np.max(np.abs(a - Bf)) > 0.1
        if cm is not None:
            off = np.nonzero(~np.repeat(cm, 8))[0]
            assert np.array_equal(a[:, off], Bf[:, off])
    Xr = rng.standard_normal((nf, 64)) + 1j * rng.standard_normal((nf, 64))
    a = fam.debug_spmv(ct1, Xr, mode=0, level=0, which=1, cmask=mask, Y0=np.full((nc, 64), 3 + 7j))
    b = fam.debug_spmv(ct1, Xr, mode=0, level=0, which=1, cmask=mask, Y0=np.full((nc, 64), 3 + 7j), no_tiles=True)
    assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
    # a lock-step solve of 64 shifted systems; the residual is formed with scipy, not with the kernel under test
    B = rng.standard_normal((d, 64)) + 1j * rng.standard_normal((d, 64))
    Xs = fam.solve(ct64, B, tol=1e-10, maxit=300)
    assert fam.last_info["n_unconverged"] == 0 and fam.last_info["relres_max"] <= 1e-10
    AXs, _, dg = TermProducts(T, Xs, "N").apply(ct64)
    R = B - AXs
    for j in range(64):                                             # error-like (diagonal-scaled) measure, as the solver's own
        assert np.linalg.norm(R[:, j] / dg[:, j]) <= 1e-8 * np.linalg.norm(B[:, j] / dg[:, j]), j
    if not big:
        L._drop_device()


def test_tile_kernel_on_the_bloch_unit_cell_at_benchmark_size_against_scipy():
    """BASELINE configs[3] (C4): the Bloch unit cell at d = 200 000 (DOS = 32) -- 10 terms + aux in 8 sparsity patterns, the seam parts
    of M, K, C and the flame term as side rows with COMPLEX per-plane coefficients exp(+-i b 2 pi / DOS) (src/Bloch.jl:4-112,
    src/Helmholtz.jl:508-513) -- in the persistent work-list regime of the tile kernel (782 fine tiles), against scipy products of
    the cell's own term matrices: wave numbers b = 0 (no phase), 5 and 16 (phase -1), one system per launch and one per column, op N
    and C, the fused forms 0 / 1 / 2 / 6, r = 64 and r = 8.  (Until round 4 this family met scipy at 728 DoF only.)"""
    from wae_amd.helmholtz import annulus
    from wae_amd.helmholtz.bloch import bloch_family
    rng = np.random.default_rng(33)
    cell = annulus.build_unit_cell(grid=annulus.PRESETS["C4"], DOS=32, tau=2e-4)
    d = cell["nsector"]
    assert abs(d - 200_000) <= 2_000
    L = bloch_family(cell)
    assert len(L.terms) >= 11
    fam = L.device()
    zs = 2 * np.pi * (np.linspace(155, 995, 64) + 1j * np.linspace(-145, 145, 64))
    X = rng.standard_normal((d, 64)) + 1j * rng.standard_normal((d, 64))
    X8 = np.ascontiguousarray(X[:, 8:16])
    tpN, tpC = FamilyProducts(L, X, "N"), FamilyProducts(L, X, "C")
    tp8 = FamilyProducts(L, X8, "N")
    for b in (0, 5, 16):
        L.params["b"] = b
        ct1 = np.array([L.coefficients(zs[40])])
        ct64 = np.array([L.coefficients(z) for z in zs])
        assert b == 0 or np.any(np.abs(ct1.imag) > 0)               # (the seam parts carry their phase factors)
        check_modes(fam, tpN, ct1, X, rng, f"C4 b={b} r=64 one system", modes=(0, 1, 2, 6))
        check_modes(fam, tpN, ct64, X, rng, f"C4 b={b} r=64 one system per column", modes=(0, 1, 2, 6))
        check_modes(fam, tp8, ct1, X8, rng, f"C4 b={b} r=8 one system", modes=(0, 1, 2, 6))
        check_modes(fam, tp8, ct64[:8], X8, rng, f"C4 b={b} r=8 one system per column", modes=(0, 2))
        for ct in (ct1, ct64):                                       # the public entries, both orientations
            want, bound, _ = tpN.apply(ct)
            assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], X), want, bound, f"C4 b={b} wae_spmv_sum r=64")
            want, bound, _ = tpC.apply(ct)
            assert_close(fam.spmv(ct if len(ct) > 1 else ct[0], X, op=2), want, bound, f"C4 b={b} wae_spmv_sum r=64 op C")
    L._drop_device()


def test_persistent_work_list_forced_onto_a_small_problem():
    """tests/tile_worker.py under WAE_TILE_GRID=8: 35 fine tiles on "8 CUs" -- static + dynamic draw, stealing, tail parts -- in three
    settings, each a child process (the switches are read once per process):
    tail 1 / 4 (parts of the last tiles); waves = 16: the 16-wavefront form of the fine-level kernel (four lanes per row; an option,
    slower: DESIGN 4b); long_row = 12 (WAE_LONG_ROW): the rows of the TRANSPOSED flame term with more than 12 entries take the
    long-side-row path of the tile kernel's transposed orientation (op = C checks of the worker), as the reference nodes' rows do at
    the benchmark sizes."""
    for tail, waves, long_row in (("1", "8", ""), ("4", "8", "12"), ("4", "16", "")):      # (in sequence: side by side they took longer)
        env = dict(os.environ, WAE_TILE_GRID="8", WAE_TILE_TAIL=tail, WAE_TILE_WAVES=waves)
        if long_row:
            env["WAE_LONG_ROW"] = long_row
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tile_worker.py")], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        res = json.loads(p.stdout.strip().split("\n")[-1])
        assert res["checks"] > 100 and res["grid"] == "8" and res["tail"] == tail
