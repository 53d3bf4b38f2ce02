"""Oracle restatement of the P1 subset of Helmholtz.discretize (test infrastructure, see oracle/__init__.py).

Used ONLY to regenerate the reference's tutorial problems (Rijke tube) as fixtures, i.e. to
produce the *inputs* of the hot path.  Follows
  src/Meshutils.jl:92-165,272-401,516-548,757-816,1079-1100   (Mesh, read_msh4, link, size, locate, field)
  src/Mesh/sorter.jl:8-169                                    (ordered simplex lists)
  src/Helmholtz.jl:54-81,120-211,232-345,405-525,528-581      (discretize)
  src/FEM/FEM.jl:2-31,435-441,704-710,1745-1766,2429-2448     (P1 element kernels)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .nlevp import LinearOperatorFamily, Term, exp_delay, pow1, pow2


class Mesh:
    pass


def read_msh4(fname, scale=1.0):
    """Meshutils.jl:272-401 (gmsh 4.1 ASCII) + Meshutils.jl:119-154 (dedupe + sorted order)."""
    with open(fname) as f:
        lines = f.read().split("\n")
    pos = 0
    tag2dom = {}
    domains = {}
    ent2dom = [dict(), dict(), dict(), dict()]
    points = None
    triangles, tetrahedra = [], []
    while pos < len(lines):
        line = lines[pos].strip()
        pos += 1
        if not line.startswith("$"):
            continue
        fld = line[1:]
        if fld == "PhysicalNames":
            n = int(lines[pos]); pos += 1
            for _ in range(n):
                dim, tag, dom = lines[pos].split()[:3]; pos += 1
                dom = dom[1:-1]
                tag2dom[tag] = dom
                domains[dom] = {"dimension": int(dim), "simplices": []}
        elif fld == "Entities":
            cnt = [int(x) for x in lines[pos].split()]; pos += 1
            for d in range(4):
                for _ in range(cnt[d]):
                    s = lines[pos].split(); pos += 1
                    k = 4 if d == 0 else 7          # 0-based index of numPhysicalTags
                    nt = int(s[k])
                    ent2dom[d][s[0]] = [tag2dom[t] for t in s[k + 1:k + 1 + nt]]
        elif fld == "Nodes":
            nb, nn = [int(x) for x in lines[pos].split()][:2]; pos += 1
            points = np.zeros((nn, 3))
            for _ in range(nb):
                nib = int(lines[pos].split()[3]); pos += 1
                tags = [int(lines[pos + j]) for j in range(nib)]; pos += nib
                for j in range(nib):
                    points[tags[j] - 1] = [float(x) for x in lines[pos + j].split()[:3]]
                pos += nib
        elif fld == "Elements":
            nb = int(lines[pos].split()[0]); pos += 1
            for _ in range(nb):
                s = lines[pos].split(); pos += 1
                edim, etag, etype, nib = int(s[0]), s[1], int(s[2]), int(s[3])
                for j in range(nib):
                    nodes = [int(x) for x in lines[pos + j].split()[1:]]
                    if etype == 2:
                        triangles.append(nodes)
                        for dom in ent2dom[edim][etag]:
                            domains[dom]["simplices"].append(len(triangles) - 1)
                    elif etype == 4:
                        tetrahedra.append(nodes)
                        for dom in ent2dom[edim][etag]:
                            domains[dom]["simplices"].append(len(tetrahedra) - 1)
                pos += nib

    def uniq_sorted(simplices):
        # sorter.jl:8-31 ordering: ascending in the key "nodes sorted descending, lexicographic";
        # node order inside a simplex is kept as in the file (first occurrence wins).
        keyed = {}
        for s in simplices:
            k = tuple(sorted(s, reverse=True))
            if k not in keyed:
                keyed[k] = s
        keys = sorted(keyed)
        index = {k: i for i, k in enumerate(keys)}
        return [keyed[k] for k in keys], index

    utri, tri_index = uniq_sorted(triangles)
    utet, tet_index = uniq_sorted(tetrahedra)
    for dom, d in domains.items():
        src = triangles if d["dimension"] == 2 else tetrahedra if d["dimension"] == 3 else None
        idx = tri_index if d["dimension"] == 2 else tet_index
        if src is None:
            d["simplices"] = []
            continue
        new = []
        for i in d["simplices"]:
            j = idx[tuple(sorted(src[i], reverse=True))]
            if j not in new:
                new.append(j)
        d["simplices"] = new
    m = Mesh()
    m.points = points * scale                              # (N,3)
    m.triangles = np.array(utri, dtype=np.int64) - 1       # 0-based
    m.tetrahedra = np.array(utet, dtype=np.int64) - 1
    m.domains = domains
    return m


def link_triangles_to_tetrahedra(mesh):
    """Meshutils.jl:516-548: for each boundary triangle the (last visited) tet owning that face."""
    tri_index = {tuple(sorted(t)): i for i, t in enumerate(mesh.triangles.tolist())}
    tri2tet = np.zeros(len(mesh.triangles), dtype=np.int64)
    for it, tet in enumerate(mesh.tetrahedra.tolist()):
        for f in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)):
            k = tuple(sorted(tet[a] for a in f))
            if k in tri_index:
                tri2tet[tri_index[k]] = it
    return tri2tet


def generate_field(mesh, func):
    """Meshutils.jl:1079-1086 (order=:const): value at the centroid of every tetrahedron."""
    ctr = mesh.points[mesh.tetrahedra].mean(axis=1)
    return np.array([func(*c) for c in ctr])


def compute_size(mesh, dom):
    """Meshutils.jl:757-767 (3-D domains)."""
    V = 0.0
    for it in mesh.domains[dom]["simplices"]:
        X = mesh.points[mesh.tetrahedra[it]]
        V += abs(np.linalg.det((X[:3] - X[3]).T)) / 6
    return V


def find_tetrahedron_containing_point(mesh, point):
    """Meshutils.jl:800-816: first tet (in list order) with all barycentric coordinates in [0,1]."""
    point = np.asarray(point, dtype=float)
    for it, tet in enumerate(mesh.tetrahedra):
        X = mesh.points[tet]
        J = (X[:3] - X[3]).T
        xi = np.linalg.solve(J, point - X[3])
        xi = np.append(xi, 1 - xi.sum())
        if np.all((0 <= xi) & (xi <= 1)):
            return it
    return -1


def _coo_tet(X):
    """FEM.jl:9-20 CooTrafo for a tetrahedron: J=[x1-x4,x2-x4,x3-x4], inverse, determinant."""
    J = (X[:3] - X[3]).T
    return J, np.linalg.inv(J), np.linalg.det(J)


_M_TET = (np.ones((4, 4)) + np.eye(4)) / 120.0          # FEM.jl:704-710
_M_TRI = (np.ones((3, 3)) + np.eye(3)) / 24.0           # FEM.jl:435-441


def _stiff_p1(Jinv, det):
    """FEM.jl:1745-1766: (grad phi_a . grad phi_b) |det J| / 6."""
    G = np.vstack([Jinv, -Jinv.sum(axis=0)])            # rows = grad phi_a
    return G @ G.T * (abs(det) / 6.0)


def discretize_p1(mesh, dscrp, c_tet):
    """Helmholtz.jl:54-581, order=:lin, non-Bloch, mass_weighting=true.

    dscrp: ordered dict  domain -> (type, data)  with types 'interior', 'admittance', 'flame'.
    Returns an oracle LinearOperatorFamily with terms in dscrp order and the aux term last.
    """
    N = mesh.points.shape[0]
    tri2tet = link_triangles_to_tetrahedra(mesh)
    c_tri = c_tet[tri2tet]
    L = LinearOperatorFamily(["ω", "λ"], [0.0, complex(np.inf, 0)])

    def tet_coo(tets, kernel):
        I, J, V = [], [], []
        for it in tets:
            tet = mesh.tetrahedra[it]
            vv = kernel(it, mesh.points[tet])
            ii = np.repeat(tet, 4).reshape(4, 4)
            I.append(ii.ravel()); J.append(ii.T.ravel()); V.append(vv.ravel())
        return np.concatenate(I), np.concatenate(J), np.concatenate(V)

    def mass_kernel(it, X):
        return _M_TET * abs(_coo_tet(X)[2])

    def stiff_kernel(it, X):
        _, Jinv, det = _coo_tet(X)
        return -c_tet[it] ** 2 * _stiff_p1(Jinv, det)       # Helmholtz.jl:120-124

    for domain, (typ, data) in dscrp.items():
        simplices = mesh.domains[domain]["simplices"]
        if typ == "interior":
            I, J, V = tet_coo(simplices, mass_kernel)
            L.push(Term(sp.csc_matrix((V.astype(complex), (I, J)), shape=(N, N)), (pow2,), (("ω",),), "ω^2", "M"))
            I, J, V = tet_coo(simplices, stiff_kernel)
            L.push(Term(sp.csc_matrix((V.astype(complex), (I, J)), shape=(N, N)), (), (), "", "K"))
        elif typ == "admittance":
            sym, val = data
            if sym not in L.params:
                L.params[sym] = complex(val)
            I, J, V = [], [], []
            for it in simplices:
                tri = mesh.triangles[it]
                X = mesh.points[tri]
                detJ = np.linalg.norm(np.cross(X[0] - X[2], X[1] - X[2]))   # FEM.jl:9-20 (unit normal column)
                vv = c_tri[it] * _M_TRI * detJ                              # Helmholtz.jl:151-156, FEM.jl:435-441
                ii = np.repeat(tri, 3).reshape(3, 3)
                I.append(ii.ravel()); J.append(ii.T.ravel()); V.append(vv.ravel())
            V = -1j * np.concatenate(V)                                     # Helmholtz.jl:459
            L.push(Term(sp.csc_matrix((V, (np.concatenate(I), np.concatenate(J))), shape=(N, N)),
                        (pow1, pow1), (("ω",), (sym,)), "ω*" + sym, "C"))
        elif typ == "flame":
            gamma, rho, nglobal, x_ref, n_ref, n_sym, tau_sym, n_val, tau_val = data
            nlocal = (gamma - 1) / rho * nglobal / compute_size(mesh, domain)   # Helmholtz.jl:325
            L.params.setdefault(n_sym, complex(n_val))
            L.params.setdefault(tau_sym, complex(tau_val))
            ref_idx = find_tetrahedron_containing_point(mesh, x_ref)
            I, S = [], []
            for it in simplices:
                tet = mesh.tetrahedra[it]
                det = _coo_tet(mesh.points[tet])[2]
                S.append(np.full(4, abs(det) / 24.0))                           # FEM.jl:2429-2431
                I.append(tet)
            I = np.concatenate(I); S = np.concatenate(S)
            tet = mesh.tetrahedra[ref_idx]
            Jinv = _coo_tet(mesh.points[tet])[1]
            Mg = np.vstack([np.eye(3), -np.ones((1, 3))])
            G = -nlocal * (Mg @ Jinv @ np.asarray(n_ref, dtype=float))          # FEM.jl:2442-2448, Helmholtz.jl:482
            II = np.repeat(I, 4); JJ = np.tile(tet, len(I)); VV = np.outer(S, G).ravel()   # Helmholtz.jl:19-33
            L.push(Term(sp.csc_matrix((VV.astype(complex), (II, JJ)), shape=(N, N)),
                        (pow1, exp_delay), ((n_sym,), ("ω", tau_sym)), f"{n_sym}*exp(-iω{tau_sym})", "Q"))
        else:
            raise ValueError(typ)
    # aux / mass-weighting term, Helmholtz.jl:528-574
    I, J, V = tet_coo(range(len(mesh.tetrahedra)), mass_kernel)
    L.push(Term(sp.csc_matrix((-V.astype(complex), (I, J)), shape=(N, N)), (pow1,), (("λ",),), "-λ", "__aux__"))
    return L


def rijke_tube(msh_path, n=0.01, tau=0.001):
    """The Rijke-tube set-up of tutorials 01/04 (docs/src/tutorial_04_perturbation_theory.md:29-48)."""
    mesh = read_msh4(msh_path, scale=0.001)
    gamma, rho, Tu, Tb, P0 = 1.4, 1.225, 300.0, 1200.0, 101325.0
    A = np.pi * 0.025 ** 2
    Q02U0 = P0 * (Tb / Tu - 1) * A * gamma / (gamma - 1)
    R = 287.05
    c = generate_field(mesh, lambda x, y, z: np.sqrt(gamma * R * Tu) if z < 0.0 else np.sqrt(gamma * R * Tb))
    dscrp = {
        "Interior": ("interior", ()),
        "Outlet": ("admittance", ("Y", 1e15)),
        "Flame": ("flame", (gamma, rho, Q02U0, [0.0, 0.0, -0.00101], [0.0, 0.0, 1.0], "n", "τ", n, tau)),
    }
    return mesh, discretize_p1(mesh, dscrp, c)
