import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from wae_amd.nlevp import LinearOperatorFamily, Term, pow1, pow2, mslp, eigs
import wae_amd.nlevp.local_solvers as LS
A2 = np.array([[0, 6, 0], [0, 6, 0], [0, 0, 1]], dtype=complex); A1 = np.array([[1, -6, 0], [2, -7, 0], [0, 0, 0]], dtype=complex); A0 = np.eye(3, dtype=complex)
T = LinearOperatorFamily()
T.push(Term(A2, (pow2,), (("λ",),), "λ^2", "A2")); T.push(Term(A1, (pow1,), (("λ",),), "λ", "A1")); T.push(Term(A0, (), (), "", "A0"))
import traceback
try:
    sol, it, flag = mslp(T, 0, tol=1e-10, output=True)
    print(sol.params, it, flag, T.device().last_info)
except Exception:
    traceback.print_exc()
try:
    T.params["λ"] = 0; T.params["__aux__"] = 0
    A = T(0); M = T.term_operator(len(T.terms) - 1, -1.0)
    print("A", A.toarray(), "M", M.toarray())
    print(eigs(A, M, nev=1, v0=np.ones(3, dtype=complex)))
except Exception:
    traceback.print_exc()
