"""Build oracle operator families from the committed fixtures (test infrastructure, see oracle/__init__.py)."""
from __future__ import annotations

import json
import os

import numpy as np
import scipy.sparse as sp

from .nlevp import LinearOperatorFamily, Term, exp_delay, pow1, pow2

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def rijke_terms():
    """CSR terms M, K, C, Q (scipy) of the Rijke-tube P1 fixture."""
    z = np.load(os.path.join(GOLDEN_DIR, "rijke_p1.npz"))
    d = int(z["d"])
    out = {}
    for name in ("M", "K", "C", "Q"):
        out[name] = sp.csr_matrix((z[f"{name}_data"], z[f"{name}_indices"], z[f"{name}_indptr"]), shape=(d, d))
    return out


def rijke_family(n=0.01, tau=0.001, Y=1e15):
    """L = ω²M + K + ωY C + n e^{-iωτ} Q  (+ aux term -λM), term order as Helmholtz.discretize pushes it
    for the tutorial's descriptor (docs/src/tutorial_04_perturbation_theory.md:52)."""
    t = rijke_terms()
    L = LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)])
    L.push(Term(sp.csc_matrix(t["M"]), (pow2,), (("ω",),), "ω^2", "M"))
    L.push(Term(sp.csc_matrix(t["K"]), (), (), "", "K"))
    L.params["Y"] = complex(Y)
    L.push(Term(sp.csc_matrix(t["C"]), (pow1, pow1), (("ω",), ("Y",)), "ω*Y", "C"))
    L.params["n"] = complex(n)
    L.params["τ"] = complex(tau)
    L.push(Term(sp.csc_matrix(t["Q"]), (pow1, exp_delay), (("n",), ("ω", "τ")), "n*exp(-iωτ)", "Q"))
    L.push(Term(sp.csc_matrix(-t["M"]), (pow1,), (("λ",),), "-λ", "__aux__"))
    return L


def qep1():
    """NLEVP-collection qep1 as used in docs/src/tutorial_00_NLEVP.md:32-42,70-101 (dense 3x3 terms)."""
    g = golden()["G8"]
    T = LinearOperatorFamily()
    T.push(Term(np.array(g["A2"], dtype=complex), (pow2,), (("λ",),), "λ^2", "A2"))
    T.push(Term(np.array(g["A1"], dtype=complex), (pow1,), (("λ",),), "λ", "A1"))
    T.push(Term(np.array(g["A0"], dtype=complex), (), (), "", "A0"))
    return T
