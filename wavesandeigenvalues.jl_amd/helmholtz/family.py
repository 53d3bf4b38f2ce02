"""Assemble a device-backed LinearOperatorFamily from Helmholtz term matrices -- the family
``Helmholtz.discretize`` returns in the reference (src/Helmholtz.jl:507-522,571-574):

    L(ω) = ω²·M + K + ω·Y·C + n·exp(-iωτ)·Q      and the auxiliary term  -λ·M  ("__aux__") last.
"""
from __future__ import annotations

import numpy as np

from ..nlevp.algebra import exp_delay, pow1, pow2
from ..nlevp.linopfam import LinearOperatorFamily, Term
from . import annulus


def helmholtz_family(terms, Y=1e15, n=1.0, tau=1e-3, device=0, flame=True):
    """terms: dict with scipy matrices M, K, C, Q (as produced by discretize / the fixtures)."""
    L = LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)], device=device)
    # M, K, C are symmetric by construction (element matrices of src/FEM/FEM.jl:435-441,704-710,1745-1766) but, summed in floating
    # point, only to rounding: tell the library so, and `A'*y` / `A'\\b` stay on the fast path (include/waehip.h, opts[0])
    L.symmetry_tol = 1e-14
    L.push(Term(terms["M"], (pow2,), (("ω",),), "ω^2", "M"))
    L.push(Term(terms["K"], (), (), "", "K"))
    L.params["Y"] = complex(Y)
    L.push(Term(terms["C"], (pow1, pow1), (("ω",), ("Y",)), "ω*Y", "C"))
    if flame:
        L.params["n"] = complex(n)
        L.params["τ"] = complex(tau)
        L.push(Term(terms["Q"], (pow1, exp_delay), (("n",), ("ω", "τ")), "n*exp(-iωτ)", "Q"))
    L.push(Term(-terms["M"], (pow1,), (("λ",),), "-λ", "__aux__"))
    return L


def annulus_family(preset="C2", device=0, **kw):
    """The synthetic annular combustor as a device-backed family; returns (L, problem dict)."""
    pb = annulus.build(preset, **{k: v for k, v in kw.items() if k in ("Y", "n", "tau", "grid")})
    p = pb["params"]
    L = helmholtz_family(pb["terms"], Y=p["Y"], n=p["n"], tau=p["τ"], device=device)
    return L, pb
