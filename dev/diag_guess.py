import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from oracle import fixtures as F, solvers as OS
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import inveriter
Lo = F.rijke_family(n=0.01, tau=0.001)
Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
Lp.solver_ref = 340 * 2 * np.pi
d = Lo.size()
rng = np.random.default_rng(1)
b = rng.standard_normal(d) + 1j * rng.standard_normal(d)
g = rng.standard_normal(d) + 0j
z = 1710 + 9j
xo = OS._solve(Lo(z), b)
x1 = Lp(z).solve(b); i1 = dict(Lp.device().last_info)
x2 = Lp(z).solve(b, guess=g); i2 = dict(Lp.device().last_info)
x3 = Lp(z).solve(b, guess=xo * (1 + 1e-3 * rng.standard_normal(d))); i3 = dict(Lp.device().last_info)
for x, i in ((x1, i1), (x2, i2), (x3, i3)):
    print(np.linalg.norm(x - xo) / np.linalg.norm(xo), i)
sol, n, flag = inveriter(Lp, 1710 + 9j, maxiter=20, tol=1e-9, output=True)
print(sol.params["ω"], n, flag)
so, no_, fo = OS.inveriter(Lo, 1710 + 9j, maxiter=20, tol=1e-9)
print(so.params["ω"], no_, fo)
