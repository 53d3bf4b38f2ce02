"""The narrow shifted solves of the Newton-type path at C3: op N against op C (the left Arnoldi process), 8 and 4 columns, with and
without a deflated guess direction: seconds and lock-step steps.
    python dev/probes/narrow_ops.py [PRESET [R OP]]      (R, OP: that one case only -- for a kernel trace)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
L, pb = annulus_family(preset, tau=2e-4)
L.solver_tol, L.solver_ref = 1e-12, 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
d = pb["d"]
rng = np.random.default_rng(3)
B = np.asfortranarray(rng.standard_normal((d, 8)) + 1j * rng.standard_normal((d, 8)))
Gd = np.asfortranarray(rng.standard_normal((d, 8)) + 1j * rng.standard_normal((d, 8)))
zs = 2 * np.pi * (np.array([310.0, 455.0, 520.0, 610.0, 700.0, 745.0, 820.0, 905.0]) + 1j * np.linspace(-60, 60, 8))
ct = np.array([L.coefficients(z) for z in zs])
only = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
for r in (8, 4, 2, 1):
    for op in (0, 2):
        if (only and (r, op) != only) or (not only and r < 4):
            continue
        for g in (None, Gd):
            for rep in range(2):
                t0 = time.perf_counter()
                X = fam.solve(ct[:r], np.asfortranarray(B[:, :r]), op=op, tol=1e-10, maxit=400, guess=None if g is None else np.asfortranarray(g[:, :r]))
                dt = time.perf_counter() - t0
            print("r=%d op=%d guess=%d: %.1f ms, lock-step steps %d, column steps %d" % (r, op, g is not None, 1e3 * dt, fam.last_info["iters_max"],
                                                                                       fam.last_info.get("iters_total", -1)), flush=True)
