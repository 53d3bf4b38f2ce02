import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wae_amd
from oracle import fixtures as F
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import householder, mslp
for n_, meth in ((0.01, householder), (1.0, mslp)):
    Lp = helmholtz_family(F.rijke_terms(), n=n_, tau=0.001)
    Lp.solver_ref = 340 * 2 * np.pi
    for rep in range(2):
        sol, n, flag = meth(Lp, 340 * 2 * np.pi, maxiter=20, tol=1e-11)
        print(meth.__name__, "n", n, "flag", flag, "steps", [abs(a - b) for a, b in zip(sol.history[1:], sol.history[:-1])][-4:])
    Lp._drop_device()
