"""One reference-shaped quadrature point at BASELINE configs[1] size (C2: 199 680 DoF) on one host core: assemble L(z), sparse
LU, l = 16 solves (beyn.jl:62-71; scipy SuperLU standing in for UMFPACK).  The measured figure bench.py's cpu_baseline cites
next to its in-run ladder (which stops at 64k DoF).  Writes profiles/r03_cpu_point_C2.json.
    python dev/cpu_point_c2.py [preset]"""
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from threadpoolctl import threadpool_limits  # noqa: E402

import wae_amd  # noqa: E402,F401
from oracle import solvers as OS  # noqa: E402
from wae_amd.helmholtz import annulus  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "C2"
l, tau, n = 16, 2e-4, 1.0
z = 2 * np.pi * (575 + 150j)
with threadpool_limits(limits=1):
    t0 = time.time()
    pb = annulus.build(preset, n=n, tau=tau)
    t_build = time.time() - t0
    T = pb["terms"]
    t0 = time.time()
    A = (z * z * T["M"] + T["K"] + z * 1e15 * T["C"] + n * np.exp(-1j * z * tau) * T["Q"]).tocsc()
    t_asm = time.time() - t0
    t0 = time.time()
    X = OS._solve(A, OS.initial_V(pb["d"], l))
    t_solve = time.time() - t0
    r = A @ X[:, 0]
    r[0] -= 1.0
out = {"preset": preset, "d": int(pb["d"]), "l": l, "cores": 1, "host_cores": os.cpu_count(), "assemble_seconds": t_asm,
       "lu_plus_solves_seconds": t_solve, "seconds_per_point": t_asm + t_solve, "peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
       "problem_build_seconds": t_build, "residual_inf_col0": float(np.max(np.abs(r))), "z_hz": [575.0, 150.0],
       "host": os.uname().nodename, "what": "assemble L(z) + SuperLU + 16 solves, one core (threadpoolctl limit 1)"}
print(json.dumps(out))
with open(os.path.join(ROOT, "profiles", f"r03_cpu_point_{preset}.json"), "w") as f:
    json.dump(out, f, indent=1)
