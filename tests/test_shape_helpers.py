"""Host-side shape-sensitivity helpers (helmholtz/shape.py) against the oracle's loop restatements, on the tutorial mesh
(tests/golden/rijke_mesh.npz).  No GPU needed: these functions are bookkeeping and small surface solves."""
import os

import numpy as np

from oracle import fixtures as F
from oracle import shape as OSH
from wae_amd.helmholtz import shape as SH


def _mesh():
    m = np.load(os.path.join(F.GOLDEN_DIR, "rijke_mesh.npz"))
    return m["points"], m["tetrahedra"].astype(np.int64)


def test_boundary_triangles_close_the_surface():
    pts, tets = _mesh()
    tri, tri2tet = SH.boundary_triangles(tets)
    # Euler characteristic of a closed surface of genus 0: V - E + F = 2
    edges = np.unique(np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [0, 2]], tri[:, [1, 2]]]), axis=1), axis=0)
    assert len(np.unique(tri)) - len(edges) + len(tri) == 2
    for t, k in zip(tri[:50], tri2tet[:50]):
        assert set(t) <= set(tets[k])
    # outward normals: the closed-surface integral of n vanishes, and (1/6)·Σ x·n = volume = Σ tet volumes
    nv = SH.get_normal_vectors(pts, tri, tets, tri2tet)
    assert np.abs(nv.sum(axis=1)).max() < 1e-12 * np.abs(nv).sum()
    ctr = pts[tri].mean(axis=1)
    X = pts[tets]
    vol = np.abs(np.linalg.det(X[:, :3] - X[:, 3:4])).sum() / 6
    assert abs(np.einsum("ij,ji->", ctr, nv) / 6 - vol) < 1e-10 * vol


def test_surface_points_and_normals_match_oracle():
    pts, tets = _mesh()
    tri, tri2tet = SH.boundary_triangles(tets)
    sp_, tri_mask, tet_mask = SH.get_surface_points(tri, tets)
    sp_o, tri_o, tet_o = OSH.get_surface_points(tri, tets)
    assert list(sp_) == sp_o
    assert all(list(a) == b for a, b in zip(tri_mask, tri_o))
    assert all(list(a) == b for a, b in zip(tet_mask, tet_o))
    nv = SH.get_normal_vectors(pts, tri, tets)                     # tri2tet found by the function itself
    assert np.array_equal(nv, OSH.get_normal_vectors(pts, tri, tets, tri2tet))


def test_normalisations_match_oracle():
    pts, tets = _mesh()
    tri, tri2tet = SH.boundary_triangles(tets)
    sp_, tri_mask, _ = SH.get_surface_points(tri, tets)
    nv = SH.get_normal_vectors(pts, tri, tets, tri2tet)
    rng = np.random.default_rng(5)
    s_surf = rng.standard_normal((3, len(sp_))) + 1j * rng.standard_normal((3, len(sp_)))
    sens = SH.scatter_to_points(s_surf, sp_, len(pts))
    a = SH.normalize_sensitivity(sp_, nv, tri_mask, sens)
    b = OSH.normalize_sensitivity(list(sp_), nv, [list(t) for t in tri_mask], sens)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    c = SH.bound_mass_normalize(sp_, nv, tri, sens)
    d = OSH.bound_mass_normalize(list(sp_), nv, tri, sens)
    assert np.abs(c - d).max() <= 1e-9 * np.abs(d).max()
    assert np.abs(c[:, np.setdiff1d(np.arange(len(pts)), sp_)]).max() == 0
    e = SH.normal_sensitivity(nv, a)
    f = OSH.normal_sensitivity(nv, a)
    assert np.abs(e - f).max() <= 1e-13 * np.abs(f).max()
    # a uniform normal displacement field: the point gradient g_p = Σ_adjacent (area/3)·n̂ normalises back to ~n̂ per triangle
    area = np.linalg.norm(nv, axis=0) / 2
    g = np.zeros((3, len(pts)))
    for t, tr in enumerate(tri):
        for p in tr:
            g[:, p] += nv[:, t] / 2 / 3
    ns = SH.normal_sensitivity(nv, SH.bound_mass_normalize(sp_, nv, tri, g.astype(complex))[:, tri].mean(axis=2))
    flat = area > 0
    assert np.median(np.abs(ns[flat] - 1.0)) < 0.05
