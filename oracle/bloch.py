"""Oracle restatement of the Bloch-periodic term splitting (test infrastructure, see oracle/__init__.py).

Follows src/Bloch.jl:4-113 (blochify, point DoFs only -- P1), src/Bloch.jl:118-143 (bloch_expand) and the part of
discretize that turns the split triplets into terms, src/Helmholtz.jl:84-105,508-513,543-574.  Written as the plain
per-entry loop of the reference (1-based index logic kept, shifted at the array boundary) so that the vectorised
producer in the package can be checked against it entry by entry.  Parity of this file is unpinned by any executed
reference output (no tutorial runs a Bloch case); it is pinned structurally by the unit-cell == full-ring identity
tested in tests/test_bloch.py.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import nlevp as O


def blochify(ii, jj, mm, naxis, nsector, axis=True):
    """ii, jj 1-based (as in the reference), point DoFs only.  Returns (I, J, M) tuples of 3 (naxis == 0) or 6 lists."""
    blochshift = nsector - naxis
    names = ("", "plus", "minus", "axis", "plus_axis", "minus_axis")
    I = {k: [] for k in names}
    J = {k: [] for k in names}
    M = {k: [] for k in names}
    for i, j, m in zip(ii, jj, mm):
        i_check = i > nsector
        if i_check:
            i -= blochshift
        j_check = j > nsector
        if j_check:
            j -= blochshift
        axis_check = bool(axis and (i <= naxis or j <= naxis))
        if (not i_check and not j_check) or (i_check and j_check):
            key = "axis" if axis_check else ""
        elif not i_check and j_check:
            key = "plus_axis" if axis_check else "plus"
        else:
            key = "minus_axis" if axis_check else "minus"
        I[key].append(i)
        J[key].append(j)
        M[key].append(m)
    keys = names[:3] if naxis == 0 else names
    return tuple(I[k] for k in keys), tuple(J[k] for k in keys), tuple(M[k] for k in keys)


def split_matrix(A, naxis, nsector):
    """Apply blochify to a scipy matrix on the extended numbering; returns the list of nsector x nsector CSC parts."""
    A = sp.coo_matrix(A)
    Is, Js, Ms = blochify((A.row + 1).tolist(), (A.col + 1).tolist(), A.data.tolist(), naxis, nsector)
    return [sp.csc_matrix((np.asarray(m, dtype=complex), (np.asarray(i, dtype=int) - 1, np.asarray(j, dtype=int) - 1)),
                          shape=(nsector, nsector)) for i, j, m in zip(Is, Js, Ms)]


def bloch_family(terms_ext, nsector, DOS, naxis=0, Y=1e15, n=1.0, tau=1e-3, b=0):
    """The family discretize returns for a mesh with a degree of symmetry (Helmholtz.jl:84-105,508-513,543-574)."""
    dphi = 2 * np.pi / DOS
    exp_plus = lambda z, k: O.exp_az(z, dphi * 1.0j, k)            # noqa: E731   Helmholtz.jl:90-91
    exp_minus = lambda z, k: O.exp_az(z, -dphi * 1.0j, k)          # noqa: E731
    y = np.zeros(DOS, dtype=complex)
    y[0] = 1.0 / DOS
    bloch_filt = O.generate_Sigma_y_exp_ikx(np.fft.fft(y))         # Helmholtz.jl:92-95
    anti = O.generate_1_gz(bloch_filt)
    bfp = O.generate_gz_hz(bloch_filt, exp_plus)
    bfm = O.generate_gz_hz(bloch_filt, exp_minus)
    extra = [((), ()), ((exp_plus,), (("b",),)), ((exp_minus,), (("b",),)),
             ((bloch_filt,), (("b",),)), ((bfp,), (("b",),)), ((bfm,), (("b",),))]
    L = O.LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)])
    L.params["Y"] = complex(Y)
    L.params["n"] = complex(n)
    L.params["τ"] = complex(tau)
    ops = [("M", (O.pow2,), (("ω",),)), ("K", (), ()), ("C", (O.pow1, O.pow1), (("ω",), ("Y",))),
           ("Q", (O.pow1, O.exp_delay), (("n",), ("ω", "τ")))]
    for name, func, arg in ops:
        for part, (f, a) in zip(split_matrix(terms_ext[name], naxis, nsector), extra):
            if part.nnz:
                L.push(O.Term(part, (*func, *f), (*arg, *a), "", name))
    A = sp.coo_matrix(terms_ext["M"])
    Is, Js, Ms = blochify((A.row + 1).tolist(), (A.col + 1).tolist(), A.data.tolist(), naxis, nsector, axis=False)
    I = np.concatenate([np.asarray(x, dtype=int) for x in Is[:3]]) - 1
    J = np.concatenate([np.asarray(x, dtype=int) for x in Js[:3]]) - 1
    V = np.concatenate([np.asarray(x, dtype=complex) for x in Ms[:3]])
    M = sp.csc_matrix((-V, (I, J)), shape=(nsector, nsector))     # Helmholtz.jl:546-549
    if naxis > 0:
        DV = np.array([1.0 / M[k, k] for k in range(naxis)], dtype=complex)
        DM = sp.csc_matrix((DV, (np.arange(naxis), np.arange(naxis))), shape=(nsector, nsector))
        L.push(O.Term(DM, (anti,), (("b",),), "(1-δ(b))", "D"))
    L.push(O.Term(M, (O.pow1,), (("λ",),), "-λ", "__aux__"))
    L.params["b"] = complex(b)
    return L


def bloch_expand(v, B, DOS, naxis, nxsector):
    """src/Bloch.jl:118-143"""
    out = np.zeros(naxis + nxsector * DOS, dtype=complex)
    out[:naxis] = v[:naxis]
    for s in range(DOS):
        out[naxis + s * nxsector:naxis + (s + 1) * nxsector] = v[naxis:naxis + nxsector] * np.exp(2.0j * np.pi / DOS * B * s)
    return out
