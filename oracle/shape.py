"""Oracle restatement of the discrete-adjoint shape sensitivity (test infrastructure, see oracle/__init__.py).

Follows src/shape_sensitivity.jl:16-141 for a full (non-unit-cell) mesh: for every surface point and coordinate the
operator derivative is a central finite difference of two re-discretisations restricted to the simplices that touch
the point, and the eigenvalue sensitivity is -v_adj' (dL/dx) v with v' v = 1, v_adj' L'(w0) v = 1.  The point/simplex
adjacency (the reference's tri_mask / tet_mask from Meshutils) is computed here directly.  Parity of this file is
unpinned by any executed reference output (no tutorial output records a shape gradient)."""
from __future__ import annotations

import copy

import numpy as np

from . import helmholtz_p1 as H


def adjacency(mesh, surface_points):
    """tet_mask / tri_mask: for each surface point the tetrahedra / boundary triangles that contain it."""
    tet_mask, tri_mask = [], []
    for p in surface_points:
        tet_mask.append(np.nonzero((mesh.tetrahedra == p).any(axis=1))[0])
        tri_mask.append(np.nonzero((mesh.triangles == p).any(axis=1))[0])
    return tri_mask, tet_mask


def discrete_adjoint_shape_sensitivity(mesh, dscrp, c_tet, surface_points, tri_mask, tet_mask, L, sol, h=1e-9):
    """shape_sensitivity.jl:16-141 (mesh.dos == 1).  Returns sens (3, npoints) complex, zero outside surface_points."""
    w0 = sol.params[sol.eigval]
    v0 = sol.v / np.sqrt(np.vdot(sol.v, sol.v))
    saved = (L.active, L.mode, dict(L.params))
    L.active, L.mode = [L.eigval], "all"
    v0_adj = sol.v_adj / np.conj(np.vdot(sol.v_adj, L(w0, 1) @ v0))
    L.active, L.mode, L.params = saved[0], saved[1], saved[2]
    sens = np.zeros((3, mesh.points.shape[0]), dtype=complex)
    for idx, p in enumerate(surface_points):
        mh = H.Mesh()
        mh.points = mesh.points.copy()
        mh.triangles, mh.tetrahedra = mesh.triangles, mesh.tetrahedra
        mh.domains = {}
        for dom in dscrp:
            dd = copy.deepcopy(mesh.domains[dom])
            keep = tri_mask[idx] if dd["dimension"] == 2 else tet_mask[idx]
            ks = set(int(k) for k in keep)
            dd["simplices"] = [s for s in dd["simplices"] if int(s) in ks]
            mh.domains[dom] = dd
        base = mesh.points[p].copy()
        local = {dom: val for dom, val in dscrp.items() if mh.domains[dom]["simplices"]}   # an empty domain contributes nothing
        for crd in range(3):
            mh.points[p] = base
            mh.points[p, crd] += h
            Dr = H.discretize_p1(mh, local, c_tet)
            mh.points[p, crd] -= 2 * h
            Dl = H.discretize_p1(mh, local, c_tet)
            for D in (Dr, Dl):
                for k, val in L.params.items():
                    if k in D.params:
                        D.params[k] = val
            Dmat = (Dr(w0) - Dl(w0)) / (2 * h)
            sens[crd, p] = -np.vdot(v0_adj, Dmat @ v0)
        mh.points[p] = base
    return sens



# ---- surface bookkeeping and normalisations (plain loops; Meshutils.jl:884-948,1030-1071; shape_sensitivity.jl:143-245) ----
def get_surface_points(triangles, tetrahedra):
    """Meshutils.jl:884-948 for mesh.dos == 1 (0-based): sorted surface point list, triangles / tetrahedra per point."""
    surface_points = []
    for tri in triangles:
        for p in tri:
            if int(p) not in surface_points:
                surface_points.append(int(p))
    surface_points.sort()
    pos = {p: i for i, p in enumerate(surface_points)}
    tri_mask = [[] for _ in surface_points]
    tet_mask = [[] for _ in surface_points]
    for t, tri in enumerate(triangles):
        for p in tri:
            tri_mask[pos[int(p)]].append(t)
    for t, tet in enumerate(tetrahedra):
        for p in tet:
            if int(p) in pos:
                tet_mask[pos[int(p)]].append(t)
    return surface_points, tri_mask, tet_mask


def get_normal_vectors(points, triangles, tetrahedra, tri2tet):
    """Meshutils.jl:1030-1071."""
    out = np.zeros((3, len(triangles)))
    for i, tri in enumerate(triangles):
        tet = tetrahedra[tri2tet[i]]
        D = [p for p in tet if p not in tri][0]
        A, B, Cc = (points[int(k)] for k in tri)
        N = np.cross(A - Cc, B - Cc)
        N = N * np.sign(np.dot(N, Cc - points[int(D)]))
        out[:, i] = N
    return out


def normalize_sensitivity(surface_points, normal_vectors, tri_mask, sens):
    """shape_sensitivity.jl:143-184."""
    ntri = normal_vectors.shape[1]
    out = np.zeros((3, ntri), dtype=complex)
    for crd in range(3):
        e = np.zeros(3)
        e[crd] = 1.0
        A = [np.linalg.norm(normal_vectors[:, t]) / 2 for t in range(ntri)]
        V = [abs(np.dot(normal_vectors[:, t], e)) / 6 for t in range(ntri)]
        for idx, pnt in enumerate(surface_points):
            tris = tri_mask[idx]
            vol = sum(abs(V[t]) for t in tris)
            if vol == 0:
                continue
            for t in tris:
                if A[t] > 0:
                    out[crd, t] += sens[crd, pnt] / A[t] * (abs(V[t]) / vol)
    return out


def bound_mass_normalize(surface_points, normal_vectors, triangles, sens):
    """shape_sensitivity.jl:186-228 (dense solve)."""
    pos = {int(p): i for i, p in enumerate(surface_points)}
    n = len(surface_points)
    B = np.zeros((n, n))
    Mloc = np.array([[1 / 12, 1 / 24, 1 / 24], [1 / 24, 1 / 12, 1 / 24], [1 / 24, 1 / 24, 1 / 12]])
    for t, tri in enumerate(triangles):
        w = np.linalg.norm(normal_vectors[:, t])
        for a in range(3):
            for b in range(3):
                B[pos[int(tri[a])], pos[int(tri[b])]] += Mloc[a, b] * w
    out = np.zeros(sens.shape, dtype=complex)
    idx = [int(p) for p in surface_points]
    for i in range(3):
        out[i, idx] = np.linalg.solve(B, sens[i, idx])
    return out


def normal_sensitivity(normal_vectors, normed_sens):
    """shape_sensitivity.jl:230-245."""
    out = np.zeros(normal_vectors.shape[1], dtype=complex)
    for t in range(normal_vectors.shape[1]):
        nvec = normal_vectors[:, t]
        out[t] = np.dot(nvec / np.linalg.norm(nvec), normed_sens[:, t])
    return out


# ---- unit cells of discretely rotationally symmetric meshes (shape_sensitivity.jl:27-35, 84-128; Meshutils.jl:946-964) ----
def get_cylindrics(pnt):
    """shape_sensitivity.jl:381-390: columns e_r, e_phi, e_z."""
    X = np.zeros((3, 3))
    X[:, 2] = [0, 0, 1]
    X[:, 0] = pnt
    X[2, 0] = 0.0
    X[:, 0] /= np.linalg.norm(X[:, 0])
    X[:, 1] = np.cross(X[:, 2], X[:, 0])
    return X


def discrete_adjoint_shape_sensitivity_unit(mesh, dscrp, c_tet, surface_points, Lb, sol, nsector, nxbloch, DOS, b, naxis=0, h=1e-9):
    """shape_sensitivity.jl:16-141 for ``mesh.dos.unit``: cylindrical displacement directions, a point of the reference Bloch
    boundary moves together with its image point (masks of the twins merged, Meshutils.jl:946-964), axis points are skipped;
    each re-discretisation of the reduced mesh (extended numbering, image points last) is folded by ``blochify`` into the
    Bloch operator at wave number ``b`` (the reference sets b = 1 there, :122-125; here the caller's).  ``Lb``: the oracle
    Bloch family of the unperturbed cell (normalisation of the adjoint vector).  Returns (3, npoints_ext)."""
    from . import bloch as OB
    w0 = sol.params[sol.eigval]
    v0 = sol.v / np.sqrt(np.vdot(sol.v, sol.v))
    saved = (Lb.active, Lb.mode, dict(Lb.params))
    Lb.active, Lb.mode = [Lb.eigval], "all"
    Lb.params["b"] = complex(b)
    v0_adj = sol.v_adj / np.conj(np.vdot(sol.v_adj, Lb(w0, 1) @ v0))
    Lb.active, Lb.mode, Lb.params = saved[0], saved[1], saved[2]
    npts = mesh.points.shape[0]
    sens = np.zeros((3, npts), dtype=complex)
    Yv = complex(dscrp["Outlet"][1][1]) if "Outlet" in dscrp else 1e15
    fl = dscrp.get("Flame")
    nv, tv = (complex(fl[1][7]), complex(fl[1][8])) if fl is not None else (0.0, 0.0)

    def folded(pts_h, local):
        mh.points = pts_h
        D = H.discretize_p1(mh, local, c_tet)
        ext = {k: sp_zero for k in ("M", "K", "C", "Q")}
        for t in D.terms:
            if t.operator in ext:
                ext[t.operator] = t.coeff
        Lf = OB.bloch_family(ext, nsector, DOS, naxis, Y=Yv, n=nv, tau=tv, b=b)
        for k, val in Lb.params.items():
            if k in Lf.params and k not in ("ω", "λ"):
                Lf.params[k] = val
        Lf.params["b"] = complex(b)
        return Lf(w0)

    import scipy.sparse as sps
    sp_zero = sps.csc_matrix((npts, npts), dtype=complex)
    for p in surface_points:
        if p < naxis:
            continue
        bloch = naxis <= p < naxis + nxbloch
        pb = npts - nxbloch + (p - naxis)
        moved = [p, pb] if bloch else [p]
        tets = np.nonzero(np.isin(mesh.tetrahedra, moved).any(axis=1))[0]
        tris = np.nonzero(np.isin(mesh.triangles, moved).any(axis=1))[0]
        mh = H.Mesh()
        mh.triangles, mh.tetrahedra = mesh.triangles, mesh.tetrahedra
        mh.domains = {}
        for dom in dscrp:
            dd = copy.deepcopy(mesh.domains[dom])
            ks = set(int(k) for k in (tris if dd["dimension"] == 2 else tets))
            dd["simplices"] = [s for s in dd["simplices"] if int(s) in ks]
            mh.domains[dom] = dd
        local = {dom: val for dom, val in dscrp.items() if mh.domains[dom]["simplices"]}
        for crd in range(3):
            pr, pl = mesh.points.copy(), mesh.points.copy()
            for q in moved:
                X = get_cylindrics(mesh.points[q])
                pr[q] += h * X[:, crd]
                pl[q] -= h * X[:, crd]
            Dmat = (folded(pr, local) - folded(pl, local)) / (2 * h)
            sens[crd, p] = -np.vdot(v0_adj, Dmat @ v0)
    return sens
