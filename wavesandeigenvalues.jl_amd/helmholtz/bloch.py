"""Bloch-periodic operator families on the unit cell of a discrete-rotationally-symmetric domain (config C4).

Restates what the reference does when ``discretize`` meets a mesh with a degree of symmetry (``mesh.dos``):

* ``blochify`` (src/Bloch.jl:4-113) sorts the assembled triplets of every operator by whether the row / the column
  lies on the image boundary (the rotated copy of the reference boundary).  Image DoFs are folded onto their
  reference twins; entries that couple across the seam go to separate matrices that are later multiplied by the phase
  factors exp(±i·b·2π/DOS) (src/Helmholtz.jl:89-91,508-513), b being the Bloch wave number, a parameter of the family.
* DoFs on the symmetry axis (``naxis`` > 0) only carry the b = 0 wave: their entries go to three more matrices that
  are multiplied by the discrete delta filter δ(b) = (1/DOS)·Σ_k exp(2πi·k·b/DOS) (src/Helmholtz.jl:92-98), and a
  diagonal term (1-δ(b))·D pins them for b ≠ 0 (src/Helmholtz.jl:551-568).
* the auxiliary mass term -λ·M is the folded mass matrix WITHOUT phase factors (src/Helmholtz.jl:543-549; the
  reference's own TODO notes that).
* ``bloch_expand`` (src/Bloch.jl:118-143) unfolds a unit-cell vector onto the full ring.

Only P1 point DoFs are handled (the line DoFs of the quadratic/hermitian element orders are outside the P1 inputs this
package produces).  Everything here is host-side input production; the device sees the result as an ordinary multi-term
family whose coefficients depend on (ω, b, ...).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from ..nlevp.algebra import (exp_delay, generate_1_gz, generate_exp_az, generate_gz_hz, generate_Sigma_y_exp_ikx, pow1,
                             pow2)
from ..nlevp.linopfam import LinearOperatorFamily, Term

SUFFIXES = ("", "+", "-", "δ", "δ+", "δ-")


def blochify(A, nsector, naxis=0, axis=True):
    """Split one operator assembled on the extended numbering (image DoFs = indices >= nsector, 0-based) into the
    (base, plus, minus[, axis, axis_plus, axis_minus]) matrices of dimension nsector.  src/Bloch.jl:4-113."""
    A = sp.coo_matrix(A)
    A.sum_duplicates()
    shift = nsector - naxis
    i, j, v = A.row.astype(np.int64), A.col.astype(np.int64), A.data.astype(complex)
    i_img, j_img = i >= nsector, j >= nsector
    i = np.where(i_img, i - shift, i)
    j = np.where(j_img, j - shift, j)
    on_axis = ((i < naxis) | (j < naxis)) if (axis and naxis > 0) else np.zeros(len(i), dtype=bool)
    same = i_img == j_img
    plus = ~i_img & j_img
    minus = i_img & ~j_img
    out = []
    for ax in ((False, True) if naxis > 0 else (False,)):
        for sel in (same, plus, minus):
            m = sel & (on_axis == ax)
            out.append(sp.csr_matrix((v[m], (i[m], j[m])), shape=(nsector, nsector)))
    for M in out:
        M.sum_duplicates()
        M.sort_indices()
    return tuple(out)


def phase_functions(DOS):
    """exp_plus, exp_minus, bloch_filt, anti_bloch_filt, bloch_exp_plus, bloch_exp_minus -- src/Helmholtz.jl:89-98."""
    dphi = 2 * np.pi / DOS

    exp_plus = generate_exp_az(1j * dphi)        # = exp_az(z, Δϕ·i, k), written so that operator files can name it
    exp_minus = generate_exp_az(-1j * dphi)
    y = np.zeros(DOS, dtype=complex)
    y[0] = 1.0 / DOS
    bloch_filt = generate_Sigma_y_exp_ikx(np.fft.fft(y))
    return {
        "exp_plus": exp_plus, "exp_minus": exp_minus, "bloch_filt": bloch_filt,
        "anti_bloch_filt": generate_1_gz(bloch_filt),
        "bloch_exp_plus": generate_gz_hz(bloch_filt, exp_plus),
        "bloch_exp_minus": generate_gz_hz(bloch_filt, exp_minus),
    }


def bloch_terms(terms_ext, nsector, DOS, naxis=0, b="b", flame=True):
    """Term list of the Bloch family in the reference's push order: for every operator its base / plus / minus
    (/ axis) parts (src/Helmholtz.jl:508-513), then D (if there is an axis), then the auxiliary mass term last."""
    pf = phase_functions(DOS)
    extra = [((), ()), ((pf["exp_plus"],), ((b,),)), ((pf["exp_minus"],), ((b,),)),
             ((pf["bloch_filt"],), ((b,),)), ((pf["bloch_exp_plus"],), ((b,),)), ((pf["bloch_exp_minus"],), ((b,),))]
    ops = [("M", (pow2,), (("ω",),), "ω^2"), ("K", (), (), ""), ("C", (pow1, pow1), (("ω",), ("Y",)), "ω*Y")]
    if flame and "Q" in terms_ext:
        ops.append(("Q", (pow1, exp_delay), (("n",), ("ω", "τ")), "n*exp(-iωτ)"))
    out = []
    for name, func, arg, txt in ops:
        for part, (f, a), suf in zip(blochify(terms_ext[name], nsector, naxis), extra, SUFFIXES):
            if part.nnz:
                out.append(Term(part, (*func, *f), (*arg, *a), txt + suf, name))
    Mparts = blochify(terms_ext["M"], nsector, naxis, axis=False)
    Mfold = sp.csr_matrix(Mparts[0] + Mparts[1] + Mparts[2])
    if naxis > 0:
        dv = 1.0 / (-Mfold.diagonal()[:naxis])              # DV = 1/M[idx,idx] with M = -mass  (Helmholtz.jl:549,558-560)
        D = sp.csr_matrix((dv, (np.arange(naxis), np.arange(naxis))), shape=(nsector, nsector), dtype=complex)
        out.append(Term(D, (pf["anti_bloch_filt"],), ((b,),), "(1-δ(b))", "D"))
    out.append(Term(-Mfold, (pow1,), (("λ",),), "-λ", "__aux__"))
    return out


def bloch_family(cell, b=0, device=0, flame=True, b_symbol="b"):
    """Device-backed family of a unit cell produced by annulus.build_unit_cell (or any dict with terms_ext, nsector,
    DOS, params[, naxis]).  ``L.params['b']`` is the Bloch wave number; change it freely between solves -- only the
    scalar coefficients change, the device copy of the matrices and the multigrid hierarchy are reused."""
    L = LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)], device=device)
    L.symmetry_tol = 1e-14          # (the base parts of M, K, C are symmetric to assembly rounding; helmholtz/family.py)
    p = cell["params"]
    L.params["Y"] = complex(p["Y"])
    if flame:
        L.params["n"] = complex(p["n"])
        L.params["τ"] = complex(p["τ"])
    for T in bloch_terms(cell["terms_ext"], cell["nsector"], cell["DOS"], cell.get("naxis", 0), b_symbol, flame):
        L.push(T)
    L.params[b_symbol] = complex(b)
    return L


def seam_terms(L):
    """indices of the seam parts (the "+"/"-" terms) of a Bloch family.  ``L.solver_opts["shape_exclude"] = seam_terms(L)``
    keeps them out of the multigrid shape matrix, so that no aggregate spans the seam: measured at C4 (d = 200 000,
    DOS = 32) this halves the iterations for wave numbers near DOS/2 (b = 16: 53 -> 26, b = 5: 60 -> 50), where the
    solution is close to antiperiodic, and quadruples them near b = 0 (21 -> 93, b = 1: 34 -> 91), where it is smooth
    across the seam -- hence off by default; one hierarchy serves a whole sweep."""
    return [k for k, t in enumerate(L.terms) if t.symbol.endswith(("+", "-")) and t.operator != "__aux__"]


def bloch_expand(v, b, DOS, nxsector=None, naxis=0):
    """Unit-cell vector -> vector on the full ring: sector s carries v·exp(+2πi·b·s/DOS); axis DoFs are copied once.
    src/Bloch.jl:118-143."""
    v = np.asarray(v)
    if nxsector is None:
        nxsector = v.shape[0] - naxis
    out = np.zeros((naxis + nxsector * DOS,) + v.shape[1:], dtype=complex)
    out[:naxis] = v[:naxis]
    for s in range(DOS):
        out[naxis + s * nxsector:naxis + (s + 1) * nxsector] = v[naxis:naxis + nxsector] * np.exp(2j * np.pi / DOS * b * s)
    return out
