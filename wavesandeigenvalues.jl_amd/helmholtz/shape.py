"""Shape-sensitivity helpers around the device kernels: surface bookkeeping, the normalisations of a
shape gradient and the finite-difference cross-check.

Mirrors, for full (non-unit-cell) P1 meshes with 0-based indices:

* ``get_surface_points``   -- src/Meshutils.jl:884-966 (surface points and the triangles / tetrahedra touching each)
* ``get_normal_vectors``   -- src/Meshutils.jl:1030-1071 (outward normals, length = twice the triangle area)
* ``normalize_sensitivity``, ``bound_mass_normalize``, ``normal_sensitivity`` -- src/shape_sensitivity.jl:143-236
* ``forward_finite_differences_shape_sensitivity`` -- src/shape_sensitivity.jl:238-337: the eigenvalue is re-solved on the
  device (``householder``) for the operator family perturbed by the two local re-discretisations
* ``discrete_adjoint_shape_sensitivity`` -- re-exported from ``helmholtz.assemble`` (device kernel)

These are small host-side loops over surface points (a few thousand entries); the operators they need are assembled
on the device (``assemble_p1``) and the eigenvalue problems run through the device solvers.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .assemble import assemble_p1, assemble_p1_flame, discrete_adjoint_shape_sensitivity  # noqa: F401  (re-export)


def boundary_triangles(tets):
    """Faces that belong to exactly one tetrahedron (the reference reads them from the mesh file).  Returns
    (triangles (ntri, 3) with sorted point indices, tri2tet (ntri,)) in lexicographic order of the sorted triple."""
    tt = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    faces = np.concatenate([tt[:, [0, 1, 2]], tt[:, [0, 1, 3]], tt[:, [0, 2, 3]], tt[:, [1, 2, 3]]])
    owner = np.tile(np.arange(tt.shape[0]), 4)
    faces = np.sort(faces, axis=1)
    order = np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))
    faces, owner = faces[order], owner[order]
    same_next = np.zeros(len(faces), dtype=bool)
    same_next[:-1] = (faces[1:] == faces[:-1]).all(axis=1)
    same_prev = np.zeros(len(faces), dtype=bool)
    same_prev[1:] = same_next[:-1]
    keep = ~(same_next | same_prev)
    return faces[keep], owner[keep]


def get_surface_points(triangles, tets):
    """surface_points (sorted point indices that occur in ``triangles``), tri_mask, tet_mask: for every surface point the
    indices of the triangles / tetrahedra that contain it, ascending.  src/Meshutils.jl:884-948."""
    tri = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    tt = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    surface_points = np.unique(tri)
    lut = np.full(int(max(tri.max(initial=-1), tt.max(initial=-1))) + 1, -1, dtype=np.int64)
    lut[surface_points] = np.arange(len(surface_points))

    def link(simplices):
        loc = lut[simplices]
        s_idx, corner = np.nonzero(loc >= 0)
        owner = loc[s_idx, corner]
        order = np.lexsort((s_idx, owner))
        owner, s_idx = owner[order], s_idx[order]
        cuts = np.searchsorted(owner, np.arange(len(surface_points) + 1))
        return [s_idx[cuts[i]:cuts[i + 1]] for i in range(len(surface_points))]

    return surface_points, link(tri), link(tt)


def get_normal_vectors(points, triangles, tets, tri2tet=None):
    """(3, ntri) outward normals of the surface triangles, |n| = 2·area.  The tetrahedron behind each triangle gives
    the inward direction (its fourth point).  src/Meshutils.jl:1030-1071."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    tri = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    tt = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    if tri2tet is None:
        faces, owner = boundary_triangles(tt)
        key = {tuple(f): o for f, o in zip(faces.tolist(), owner.tolist())}
        try:
            tri2tet = np.array([key[tuple(sorted(t))] for t in tri.tolist()], dtype=np.int64)
        except KeyError as e:
            raise ValueError(f"triangle {e.args[0]} is not a boundary face of the tetrahedra") from None
    tet = tt[np.asarray(tri2tet, dtype=np.int64)]
    inside = np.array([[p for p in t if p not in s][0] for t, s in zip(tet.tolist(), tri.tolist())], dtype=np.int64)
    A, B, Cc, D = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]], pts[inside]
    N = np.cross(A - Cc, B - Cc)
    N *= np.sign(np.einsum("ij,ij->i", N, Cc - D))[:, None]
    return N.T.copy()


def scatter_to_points(sens_surface, surface_points, npoints):
    """(3, len(surface_points)) as returned by the sensitivity routines here -> the reference's (3, npoints) layout."""
    out = np.zeros((3, int(npoints)), dtype=complex)
    out[:, np.asarray(surface_points, dtype=np.int64)] = sens_surface
    return out


def normalize_sensitivity(surface_points, normal_vectors, tri_mask, sens):
    """Distribute the point sensitivities ``sens`` (3, npoints) onto the adjacent triangles, weighted with the projected
    triangle areas and divided by the triangle area.  Returns (3, ntri).  src/shape_sensitivity.jl:143-184."""
    nv = np.asarray(normal_vectors, dtype=np.float64)
    A = np.linalg.norm(nv, axis=0) / 2
    out = np.zeros(nv.shape, dtype=complex)
    for crd in range(3):
        V = np.abs(nv[crd]) / 6
        for idx, pnt in enumerate(surface_points):
            tris = np.asarray(tri_mask[idx], dtype=np.int64)
            vol = V[tris].sum()
            if vol == 0:
                continue
            ok = A[tris] > 0
            np.add.at(out[crd], tris[ok], sens[crd, pnt] / A[tris[ok]] * (V[tris[ok]] / vol))
    return out


def bound_mass_normalize(surface_points, normal_vectors, triangles, sens):
    """Solve B·nsens = sens on the surface, B = boundary mass matrix of all surface triangles (P1, weights |n|·(1+δ)/24).
    ``sens`` and the result are (3, npoints).  src/shape_sensitivity.jl:186-228."""
    tri = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    sp_ = np.asarray(surface_points, dtype=np.int64)
    lut = np.full(int(max(tri.max(), sp_.max())) + 1, -1, dtype=np.int64)
    lut[sp_] = np.arange(len(sp_))
    loc = lut[tri]
    if (loc < 0).any():
        raise ValueError("a triangle point is missing from surface_points")
    w = np.linalg.norm(np.asarray(normal_vectors, dtype=np.float64), axis=0)
    Mloc = (np.ones((3, 3)) + np.eye(3)) / 24.0
    vals = (w[:, None, None] * Mloc).ravel()
    B = sp.csc_matrix((vals, (np.repeat(loc, 3, axis=1).ravel(), np.tile(loc, (1, 3)).ravel())), shape=(len(sp_), len(sp_)))
    lu = spla.splu(B)
    out = np.zeros(np.shape(sens), dtype=complex)
    for i in range(3):
        rhs = np.asarray(sens[i, sp_], dtype=complex)
        out[i, sp_] = lu.solve(rhs.real) + 1j * lu.solve(rhs.imag)
    return out


def normal_sensitivity(normal_vectors, normed_sens):
    """Component of the per-triangle gradient along the unit outward normal.  src/shape_sensitivity.jl:230-245
    (Julia's ``dot`` conjugates its first argument; the normals are real)."""
    nv = np.asarray(normal_vectors, dtype=np.float64)
    return np.einsum("ij,ij->j", nv / np.linalg.norm(nv, axis=0), np.asarray(normed_sens))


def _boundary_mass(points, tris, c_tri, n):
    """-i·c·∫φ_aφ_b on the given triangles (src/FEM/FEM.jl:435-441, src/Helmholtz.jl:151-156)."""
    P = points[tris]
    area2 = np.linalg.norm(np.cross(P[:, 0] - P[:, 2], P[:, 1] - P[:, 2]), axis=1)
    Cloc = (np.ones((3, 3)) + np.eye(3)) / 24.0
    v = (-1j * (c_tri * area2)[:, None, None] * Cloc).ravel()
    return sp.csr_matrix((v, (np.repeat(tris, 3, axis=1).ravel(), np.tile(tris, (1, 3)).ravel())), shape=(n, n))


def forward_finite_differences_shape_sensitivity(points, tets, c_tet, surface_points, L, sol, bnd_tris=None, bnd_c=None, h=1e-9,
                                                 device=0, maxiter=5, order=3, nev=3, flame=None):
    """Eigenvalue shift per unit displacement of every point in ``surface_points`` along x, y, z, by re-solving
    (src/shape_sensitivity.jl:238-337, full mesh): the family G = L + (D₊ − D₋), D± the discretisations of the simplices
    touching the point with the point moved by ±h, is solved with ``householder`` from the known eigenvalue and
    sens = (ω_new − ω₀)/(2h).  ``L`` must carry terms with operators "M", "K" and, if ``bnd_tris`` is given, "C".
    ``flame`` (dict as for discrete_adjoint_shape_sensitivity): the flame operator "Q" of the tetrahedra at the point is
    re-assembled on the device too (``assemble_p1_flame`` on the reduced flame domain, volume included -- what the reference's
    ``discretize`` of the reduced mesh does).
    Returns (3, len(surface_points)) complex.  One device family and one eigen-solve per point and coordinate: a
    cross-check for a handful of points, not a production gradient (that is discrete_adjoint_shape_sensitivity)."""
    from ..nlevp.linopfam import LinearOperatorFamily, Term
    from ..nlevp.local_solvers import householder
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    tt = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    cc = np.ones(tt.shape[0]) if c_tet is None else np.asarray(c_tet, dtype=np.float64)
    n = pts.shape[0]
    tri = None if bnd_tris is None else np.asarray(bnd_tris, dtype=np.int64).reshape(-1, 3)
    w0 = complex(sol.params[sol.eigval])
    sens = np.zeros((3, len(surface_points)), dtype=complex)
    for idx, p in enumerate(np.asarray(surface_points, dtype=np.int64)):
        tsel = np.nonzero((tt == p).any(axis=1))[0]
        ssel = np.zeros(0, dtype=np.int64) if tri is None else np.nonzero((tri == p).any(axis=1))[0]
        for crd in range(3):
            D = {}
            for sgn in (+1, -1):
                ph = pts.copy()
                ph[p, crd] += sgn * h
                M, K = assemble_p1(ph, tt[tsel], cc[tsel], device=device)
                D[sgn] = {"M": M, "K": K}
                if len(ssel):
                    D[sgn]["C"] = _boundary_mass(ph, tri[ssel], np.asarray(bnd_c, dtype=np.float64)[ssel], n)
                if flame is not None:
                    fsel = np.intersect1d(np.asarray(flame["flame_tets"], dtype=np.int64), tsel)
                    if len(fsel):
                        D[sgn]["Q"], _ = assemble_p1_flame(ph, tt, fsel, flame["ref_tet"], flame["n_ref"], flame["nglobal_scaled"], device=device)
            G = LinearOperatorFamily([L.eigval, L.auxval], [0.0, complex(np.inf, 0)], device=device)   # shape_sensitivity.jl:311
            for k, val in L.params.items():
                G.params[k] = val
            for T in L.terms:
                A = sp.csr_matrix(T.coeff, dtype=complex)
                if T.symbol != "__aux__" and T.operator != "__aux__" and T.operator in D[+1]:
                    A = sp.csr_matrix(A + D[+1][T.operator] - D[-1][T.operator])
                G.push(Term(A, T.func, T.params, T.symbol, T.operator))
            G.params[L.eigval] = w0
            new_sol, _, _ = householder(G, w0, maxiter=maxiter, output=False, nev=nev, order=order)
            sens[crd, idx] = (complex(new_sol.params[new_sol.eigval]) - w0) / (2 * h)
            G._drop_device()
    return sens


# ------------------------------------------------------------------------------------------------------
# unit cells of discretely rotationally symmetric meshes (mesh.dos.unit; src/shape_sensitivity.jl:27-35,84-128)
# ------------------------------------------------------------------------------------------------------
def get_cylindrics(pnt):
    """columns e_r, e_phi, e_z at ``pnt``  (src/shape_sensitivity.jl:381-390)"""
    X = np.zeros((3, 3))
    X[:, 2] = [0.0, 0.0, 1.0]
    X[:2, 0] = pnt[:2]
    X[:, 0] /= np.linalg.norm(X[:, 0])
    X[:, 1] = np.cross(X[:, 2], X[:, 0])
    return X


def bloch_extend(v, b, nsector, nxbloch, DOS, naxis=0):
    """unit-cell vector -> vector on the cell's EXTENDED numbering (image-plane nodes last): an image node carries its
    reference twin's value times exp(i·b·2π/DOS) -- the map E_b with L_b = E_bᴴ L_ext E_b that ``blochify`` folds into the
    "+"/"-" terms (src/Bloch.jl:4-113, src/Helmholtz.jl:89-91)."""
    v = np.asarray(v, dtype=np.complex128)
    out = np.zeros(nsector + nxbloch, dtype=np.complex128)
    out[:nsector] = v
    out[nsector:] = v[naxis:naxis + nxbloch] * np.exp(2j * np.pi / DOS * complex(b))
    return out


def discrete_adjoint_shape_sensitivity_unit_cell(cell, surface_points, sol, L, b=None, h=1e-9, device=0, flame=True):
    """Shape sensitivity on the unit cell of a Bloch-periodic problem (``mesh.dos.unit``; src/shape_sensitivity.jl:27-35,84-128):
    a surface point is displaced along e_r, e_phi, e_z (``get_cylindrics``), a point of the reference Bloch boundary together
    with its image point (each along ITS cylindrical directions), points on the symmetry axis are skipped; the derivative of the
    Bloch operator L_b(ω) is contracted with the unit-cell eigenvectors.

    Here: with E_b the extension of a unit-cell vector to the cell's extended numbering (``bloch_extend``), L_b = E_bᴴ L_ext E_b
    -- exactly what ``blochify`` folds -- so the sensitivity is -(E_b v_adj)ᴴ (dL_ext/dx) (E_b v): the Cartesian device kernel
    on the cell's own mesh with extended vectors, for the point and its image; the cylindrical components are the directional
    combinations Xᵀ·(Cartesian gradient) (a central difference along a unit direction differs from the combination of the three
    axis differences by O(h²)).  ``cell``: dict of annulus.build_unit_cell; ``L``: its Bloch family; ``b``: Bloch wave number
    (default ``sol.params['b']``; the reference evaluates at b = 1, shape_sensitivity.jl:122-125).
    Returns (3, len(surface_points)): components along e_r, e_phi, e_z."""
    m = cell["info"]["mesh"]
    pts = np.asarray(cell["points"], dtype=np.float64)
    ns, nxb, DOS, naxis = int(cell["nsector"]), int(cell["nxbloch"]), int(cell["DOS"]), int(cell.get("naxis", 0))
    d_ext = pts.shape[0]
    assert d_ext == ns + nxb
    b = complex(sol.params.get("b", 0.0)) if b is None else complex(b)
    w0 = complex(sol.params[sol.eigval])
    v = np.asarray(sol.v, dtype=np.complex128)
    v = v / np.sqrt(np.vdot(v, v))
    saved = (L.active, L.mode, dict(L.params))
    L.active, L.mode = [L.eigval], "all"
    L.params["b"] = b
    try:
        va = np.asarray(sol.v_adj, dtype=np.complex128)
        va = va / np.conj(np.vdot(va, L(w0, 1) @ v))
        cQ = None
        if flame and m["flames"]:
            kq = [i for i, t in enumerate(L.terms) if t.operator == "Q" and t.symbol.endswith(")")]      # the base part n·exp(-iωτ)
            cQ = complex(L.coefficients(w0)[kq[0]]) if kq else None
    finally:
        L.active, L.mode, L.params = saved
    sp_ = np.asarray(surface_points, dtype=np.int64)
    twin = lambda p: d_ext - nxb + (p - naxis)                  # noqa: E731   (shape_sensitivity.jl:88)
    on_bloch = (sp_ >= naxis) & (sp_ < naxis + nxb)
    allp = np.unique(np.concatenate([sp_, twin(sp_[on_bloch])]))
    fl = None
    if cQ is not None:
        f0 = m["flames"][0]
        touch = np.isin(m["tets"][f0["flame_tets"]], np.concatenate([np.arange(naxis, naxis + nxb), np.arange(ns, d_ext)])).any()
        if touch:       # (shape_sensitivity.jl:75-106 moves such a point together with its image for every domain of dscrp, the flame's included)
            raise NotImplementedError("flame tetrahedra on the Bloch boundary: the reduced flame domain of a point and its image point would have "
                                      "to be merged; keep the flame inside the sector (as discretize's unit-cell meshes do) or pass flame=False")
        fl = {**f0, "coeff": cQ}
    Y = complex(cell["params"]["Y"])
    cart = discrete_adjoint_shape_sensitivity(pts, m["tets"], m["c_tet"], allp, sol, L, bnd_tris=m["outlet_tris"], bnd_c=m["outlet_c"], Y=Y, h=h,
                                              device=device, flame=fl,
                                              v_ext=(bloch_extend(v, b, ns, nxb, DOS, naxis), bloch_extend(va, b, ns, nxb, DOS, naxis)))
    pos = {int(p): i for i, p in enumerate(allp)}
    out = np.zeros((3, len(sp_)), dtype=np.complex128)
    for k, p in enumerate(sp_):
        if p < naxis:
            continue                                          # axis points are skipped (shape_sensitivity.jl:93-98)
        out[:, k] = get_cylindrics(pts[p]).T @ cart[:, pos[int(p)]]
        if naxis <= p < naxis + nxb:
            q = twin(p)
            out[:, k] += get_cylindrics(pts[q]).T @ cart[:, pos[int(q)]]
    return out
