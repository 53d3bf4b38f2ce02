#!/usr/bin/env python3
"""Development probe run on the GPU box: SpMV timing sweep and solver statistics (not a test, not the bench)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd  # noqa: E402
from wae_amd.helmholtz.family import annulus_family  # noqa: E402
from wae_amd.nlevp import compute_moment_matrices, gauss_points, moments2eigs  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "20k"
what = sys.argv[2] if len(sys.argv) > 2 else "all"
t0 = time.time()
L, pb = annulus_family(preset)
print("built", preset, pb["d"], pb["info"], f"{time.time() - t0:.1f}s", flush=True)
fam = L.device()
z = 2 * np.pi * (500 + 20j)
cz = L.coefficients(z)
if what in ("all", "spmv"):
    for r in (1, 8, 64):
        base = fam.spmv_bytes(r=r, mask=[1, 1, 1, 1, 0])
        for C, S in ((1, 8), (2, 8), (4, 4), (8, 1), (8, 2), (16, 2)):
            if C > r:
                continue
            os.environ["WAE_SPMV_C"], os.environ["WAE_SPMV_S"] = str(C), str(S)
            for lds in ((1, 0) if (C, S) == (8, 1) else (0,)):
                os.environ["WAE_SPMV_LDS"] = str(lds)
                ms = fam.bench_spmv(cz, r=r, reps=20)
                print(f"spmv r={r} C={C} S={S} lds={lds}: {ms * 1e3:.1f} us  {base / ms / 1e6:.0f} GB/s (algorithmic)", flush=True)
            os.environ.pop("WAE_SPMV_LDS", None)
    os.environ.pop("WAE_SPMV_C"); os.environ.pop("WAE_SPMV_S")
if what in ("all", "solve"):
    L.solver_ref = 2 * np.pi * 500
    t0 = time.time()
    L.ensure_solver()
    print(f"solver setup {time.time() - t0:.2f}s", flush=True)
    d = pb["d"]
    B = np.zeros((d, 8), dtype=complex); B[:8, :8] = np.eye(8)
    for zz in (2 * np.pi * (150 + 5j), 2 * np.pi * (575 + 50j), 2 * np.pi * (1000 - 2j)):
        t0 = time.time()
        X = L(zz).solve(B, tol=1e-10)
        print(f"solve z/2pi={zz / 2 / np.pi:.0f} {time.time() - t0:.3f}s", fam.last_info, flush=True)
    Gam = np.array([150 + 100j, 150 - 100j, 1000 - 100j, 1000 + 100j]) * 2 * np.pi
    for N in (8, 32):
        t0 = time.time()
        A = compute_moment_matrices(L, Gam, l=8, K=1, N=N)
        dt = time.time() - t0
        Om, P, S = moments2eigs(A, return_sigma=True)
        print(f"beyn N={N}: {dt:.2f}s", fam.last_info, "sigma", S, "Om/2pi", np.sort_complex(Om / 2 / np.pi), flush=True)
