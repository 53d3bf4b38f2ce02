import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from oracle import fixtures as F
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import householder, eigs
Lp = helmholtz_family(F.rijke_terms(), n=0.01, tau=0.001)
Lp.solver_ref = 340 * 2 * np.pi
fam = Lp.ensure_solver()
orig = fam.arnoldi
def traced(*a, **k):
    t=time.time(); r = orig(*a, **k); print('   arnoldi m=%d %.3fs'%(a[2], time.time()-t), fam.last_info, flush=True); return r
fam.arnoldi = traced
t=time.time()
sol, n, flag = householder(Lp, 340 * 2 * np.pi, maxiter=int(sys.argv[1]) if len(sys.argv)>1 else 8, tol=1e-11, output=True)
print('done', sol.params['ω'], n, flag, time.time()-t)
from wae_amd.nlevp import mslp
Lp2 = helmholtz_family(F.rijke_terms(), n=1.0, tau=0.001); Lp2.solver_ref = 340*2*np.pi
fam2 = Lp2.ensure_solver(); o2 = fam2.arnoldi
def tr2(*a, **k):
    r = o2(*a, **k); print("   arnoldi", fam2.last_info, flush=True); return r
fam2.arnoldi = tr2
t=time.time(); sol2, n2, f2 = mslp(Lp2, 340*2*np.pi, maxiter=20, tol=1e-11, output=True); print("mslp", sol2.params["ω"], n2, f2, time.time()-t)
