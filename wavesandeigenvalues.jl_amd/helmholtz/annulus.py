"""Synthetic annular-combustor Helmholtz problem (P1 tetrahedra), the input producer for configs C2/C3.

This stands where the reference's ``Helmholtz.discretize(mesh, dscrp, c)`` (src/Helmholtz.jl:54-581) stands:
it produces the matrix terms of

    L(ω) = ω²·M + K + ω·Y·C + n·exp(-iωτ)·Q          (+ the auxiliary term -λ·M, src/Helmholtz.jl:571-574)

with the same element formulas (P1 mass / stiffness / boundary mass / volume source / gradient source:
src/FEM/FEM.jl:435-441,704-710,1745-1766,2429-2448) on a deterministic structured mesh, vectorised with
numpy.  Geometry: annulus r∈[0.1,0.2] m, height 0.5 m; every hexahedron of the (θ,z,r) grid is
cut into 6 tetrahedra (Kuhn), nodes are numbered lexicographically in (θ,z,r); c=347 m/s for z<0.3 else
850 m/s; outlet admittance on the top face; 12 flame slabs (z∈[0.25,0.30], central half of each 30° sector)
sharing one n-τ flame response, each with its own reference point just upstream (z=0.24).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

R_IN, R_OUT, HEIGHT = 0.1, 0.2, 0.5
N_SECTOR = 12
Z_JUMP = 0.3
C_COLD, C_HOT = 347.0, 850.0
FLAME_Z0, FLAME_Z1, REF_Z = 0.25, 0.30, 0.24

# preset grids (nθ, nz, nr) -> d = nθ·nz·nr
PRESETS = {
    "tiny": (24, 12, 4),        # 1 152 DoF   (unit tests)
    "small": (48, 26, 7),       # 8 736 DoF
    "20k": (72, 36, 8),         # 20 736 DoF
    "C2": (160, 78, 16),        # 199 680 DoF (BASELINE.json configs[1])
    "C5": (224, 106, 21),       # 498 624 DoF (configs[4])
    "C3": (288, 128, 27),       # 995 328 DoF (configs[2])
    "C4": (20, 200, 50),        # 200 000 DoF: ONE sector (unit cell) of a 32-fold symmetric ring (configs[3])
}

# the 6 Kuhn tetrahedra of the unit cube, as corner indices (bit0=θ, bit1=z, bit2=r)
_KUHN = []
for perm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)):
    v = 0
    tet = [0]
    for ax in perm:
        v |= 1 << ax
        tet.append(v)
    _KUHN.append(tet)
_KUHN = np.array(_KUHN)


def _mesh(nth, nz, nr, sector_of=None):
    """Full ring (θ periodic, nth planes) or, with sector_of=DOS, ONE sector of a DOS-fold symmetric ring: nth cells,
    nth+1 planes, the last plane being the rotated image of the first (the unit cell of a Bloch computation)."""
    if sector_of is None:
        th = np.arange(nth) * (2 * np.pi / nth)
        nplanes, ncells = nth, nth
    else:
        th = np.arange(nth + 1) * (2 * np.pi / sector_of / nth)
        nplanes, ncells = nth + 1, nth
    z = np.linspace(0.0, HEIGHT, nz)
    r = np.linspace(R_IN, R_OUT, nr)
    TH, Z, RR = np.meshgrid(th, z, r, indexing="ij")
    pts = np.stack([RR * np.cos(TH), RR * np.sin(TH), Z], axis=-1).reshape(-1, 3)

    def nid(i, j, k):
        return ((i % nplanes) * nz + j) * nr + k

    I, J, K = np.meshgrid(np.arange(ncells), np.arange(nz - 1), np.arange(nr - 1), indexing="ij")
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    corners = np.stack([nid(I + (c & 1), J + ((c >> 1) & 1), K + ((c >> 2) & 1)) for c in range(8)], axis=1)
    tets = corners[:, _KUHN].reshape(-1, 4)
    return pts, tets, (th, z, r)


def _coo_to_csr(I, J, V, d):
    A = sp.coo_matrix((V, (I, J)), shape=(d, d)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def build(preset="C2", Y=1e15, n=1.0, tau=1e-3, grid=None, n_sector=N_SECTOR, ref_offset="cartesian"):
    """Return dict(d, terms={M,K,C,Q} scipy CSR complex128, params, points, info).

    ref_offset: how the flame reference points are nudged off the grid planes -- "cartesian" (+1e-7 in x, y, z; the
    historical C2/C3 inputs) or "polar" (+1e-7 in θ, r, z: identical in every sector, which makes a grid with
    nθ divisible by n_sector exactly n_sector-fold symmetric -- the comparator for the Bloch unit cell)."""
    nth, nz, nr = grid if grid is not None else PRESETS[preset]
    pts, tets, _ = _mesh(nth, nz, nr)
    terms, info = _assemble(pts, tets, nz, n_sector, range(n_sector), ref_offset)
    info["grid"] = (nth, nz, nr)
    return {
        "d": pts.shape[0],
        "terms": terms,
        "params": {"Y": complex(Y), "n": complex(n), "τ": complex(tau)},
        "points": pts,
        "info": info,
    }


def _assemble(pts, tets, nz, n_sector, flame_sectors, ref_offset):
    """Element loops of discretize (src/Helmholtz.jl:405-525) for the fixed annulus description, vectorised."""
    d = pts.shape[0]
    X = pts[tets]                                            # (nt,4,3)
    Jm = np.transpose(X[:, :3, :] - X[:, 3:4, :], (0, 2, 1))  # columns x_a - x_4  (FEM.jl:9-20)
    det = np.linalg.det(Jm)
    Jinv = np.linalg.inv(Jm)
    adet = np.abs(det)
    ctr = X.mean(axis=1)
    c_tet = np.where(ctr[:, 2] < Z_JUMP, C_COLD, C_HOT)

    ii = np.repeat(tets, 4, axis=1).ravel()
    jj = np.tile(tets, (1, 4)).ravel()
    Mloc = (np.ones((4, 4)) + np.eye(4)) / 120.0              # FEM.jl:704-710
    Mv = (adet[:, None, None] * Mloc).ravel()
    G = np.concatenate([Jinv, -Jinv.sum(axis=1, keepdims=True)], axis=1)   # rows = ∇φ_a (FEM.jl:1745-1766)
    Kv = (-(c_tet ** 2 * adet / 6.0)[:, None, None] * (G @ np.transpose(G, (0, 2, 1)))).ravel()   # Helmholtz.jl:120-124
    M = _coo_to_csr(ii, jj, Mv.astype(complex), d)
    K = _coo_to_csr(ii, jj, Kv.astype(complex), d)

    # outlet: tet faces lying in the top plane (boundary mass, FEM.jl:435-441; Helmholtz.jl:151-156,459)
    top = np.isclose(pts[:, 2], HEIGHT)
    faces = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    tri_nodes = tets[:, faces]                                # (nt,4,3)
    on_top = top[tri_nodes].all(axis=2)
    t_idx, f_idx = np.nonzero(on_top)
    tri = tri_nodes[t_idx, f_idx]                             # (ntri,3)
    P = pts[tri]
    area2 = np.linalg.norm(np.cross(P[:, 0] - P[:, 2], P[:, 1] - P[:, 2]), axis=1)
    Cloc = (np.ones((3, 3)) + np.eye(3)) / 24.0
    Cv = (-1j * (c_tet[t_idx] * area2)[:, None, None] * Cloc).ravel()
    C = _coo_to_csr(np.repeat(tri, 3, axis=1).ravel(), np.tile(tri, (1, 3)).ravel(), Cv, d)

    # flames: Q = Σ_f S_f ⊗ g_f   (Helmholtz.jl:292-344,464-487; FEM.jl:2429-2448)
    gamma, rho, Tu, Tb, P0 = 1.4, 1.225, 300.0, 1200.0, 101325.0
    A_f = np.pi * (R_OUT ** 2 - R_IN ** 2) / n_sector
    Q02U0 = P0 * (Tb / Tu - 1) * A_f * gamma / (gamma - 1)
    ang = np.mod(np.arctan2(ctr[:, 1], ctr[:, 0]), 2 * np.pi)
    sector = np.floor(ang / (2 * np.pi / n_sector)).astype(int)
    frac = ang / (2 * np.pi / n_sector) - sector
    in_flame = (ctr[:, 2] > FLAME_Z0) & (ctr[:, 2] < FLAME_Z1) & (frac > 0.25) & (frac < 0.75)
    QI, QJ, QV = [], [], []
    flames = []                                               # per flame: its tetrahedra, reference tetrahedron, (γ-1)/ρ·Q02U0
    n_ref = np.array([0.0, 0.0, 1.0])
    r_mid = 0.5 * (R_IN + R_OUT)
    for f in flame_sectors:
        sel = np.nonzero(in_flame & (sector == f))[0]
        if len(sel) == 0:
            continue
        vol = adet[sel].sum() / 6.0
        nlocal = (gamma - 1) / rho * Q02U0 / vol               # Helmholtz.jl:325
        S_nodes = tets[sel].ravel()
        S_vals = np.repeat(adet[sel] / 24.0, 4)                # FEM.jl:2429-2431
        a0 = (f + 0.5) * 2 * np.pi / n_sector
        if ref_offset == "polar":
            x_ref = np.array([(r_mid + 1e-7) * np.cos(a0 + 1e-7), (r_mid + 1e-7) * np.sin(a0 + 1e-7), REF_Z + 1e-7])
        else:
            x_ref = np.array([r_mid * np.cos(a0), r_mid * np.sin(a0), REF_Z]) + 1e-7
        # first tet (list order) containing x_ref (Meshutils.jl:800-816), searched among nearby tets only
        near = np.nonzero(np.linalg.norm(ctr - x_ref, axis=1) < 4 * HEIGHT / nz)[0]
        ref = -1
        for it in near:
            xi = np.linalg.solve(Jm[it], x_ref - X[it, 3])
            xi = np.append(xi, 1 - xi.sum())
            if np.all((xi >= 0) & (xi <= 1)):
                ref = it
                break
        assert ref >= 0, "reference point not found"
        flames.append({"flame_tets": sel.astype(np.int32), "ref_tet": int(ref), "n_ref": n_ref.copy(), "x_ref": x_ref.copy(),
                       "nglobal_scaled": float((gamma - 1) / rho * Q02U0), "volume": float(vol)})
        g = -nlocal * (G[ref] @ n_ref)                         # FEM.jl:2442-2448, Helmholtz.jl:482
        QI.append(np.repeat(S_nodes, 4)); QJ.append(np.tile(tets[ref], len(S_nodes)))
        QV.append(np.outer(S_vals, g).ravel())
    Q = _coo_to_csr(np.concatenate(QI), np.concatenate(QJ), np.concatenate(QV).astype(complex), d)

    return ({"M": M, "K": K, "C": C, "Q": Q},
            {"ntets": len(tets), "ntri_outlet": len(tri), "nflame_tets": int(in_flame.sum()),
             # the mesh description behind the matrices (shape sensitivities re-discretise simplices): tetrahedra, speed of sound,
             # outlet triangles with the speed of sound of the tetrahedron behind each, flames
             "mesh": {"tets": tets.astype(np.int32), "c_tet": c_tet, "outlet_tris": tri.astype(np.int32), "outlet_c": c_tet[t_idx],
                      "flames": flames}})


def build_unit_cell(grid=(8, 26, 7), DOS=N_SECTOR, Y=1e15, n=1.0, tau=1e-3):
    """One sector of the DOS-fold symmetric annulus, discretised on its own mesh (nθc cells, nθc+1 planes).  The
    matrices are returned on the EXTENDED numbering (image plane included, d_ext = (nθc+1)·nz·nr, image nodes last,
    reference-plane nodes first -- the layout src/Bloch.jl:4-8 assumes with naxis = 0); helmholtz/bloch.py folds them
    into the Bloch terms.  Same element formulas and flame model as build()."""
    nthc, nz, nr = grid
    pts, tets, _ = _mesh(nthc, nz, nr, sector_of=DOS)
    terms, info = _assemble(pts, tets, nz, DOS, [0], "polar")
    info["grid"] = grid
    return {
        "d_ext": pts.shape[0], "nsector": nthc * nz * nr, "nxbloch": nz * nr, "DOS": DOS,
        "terms_ext": terms,
        "params": {"Y": complex(Y), "n": complex(n), "τ": complex(tau)},
        "points": pts,
        "info": info,
    }
