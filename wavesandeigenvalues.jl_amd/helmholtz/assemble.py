"""P1 mass and stiffness matrices assembled on the device (wae_p1_assemble, include/waehip.h) -- the element loops of
``discretize`` for the "interior" domain (src/Helmholtz.jl:405-441; kernels src/FEM/FEM.jl:704-710,1745-1766)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from .. import _lib


def assemble_p1(points, tets, c_tet=None, device=0, dtype=np.complex128):
    """points (npoints, 3), tets (ntets, 4) 0-based, c_tet (ntets,) speed of sound per tetrahedron (None = 1).
    Returns (M, K) as scipy CSR matrices sharing one pattern; K = -c² ∫ ∇φ_a·∇φ_b as in the reference (Helmholtz.jl:120-124)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    tt = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    cc = None if c_tet is None else np.ascontiguousarray(c_tet, dtype=np.float64)
    L = _lib.lib()
    h = C.c_void_p()
    dp = C.POINTER(C.c_double)
    _lib.check(L.wae_p1_assemble(int(device), pts.shape[0], pts.ctypes.data_as(dp), tt.shape[0], tt.ctypes.data_as(C.POINTER(C.c_int32)),
                                 None if cc is None else cc.ctypes.data_as(dp), C.byref(h)))
    try:
        n, nnz = C.c_int64(0), C.c_int64(0)
        _lib.check(L.wae_p1_info(h, C.byref(n), C.byref(nnz)))
        rowptr = np.zeros(n.value + 1, dtype=np.int32)
        col = np.zeros(nnz.value, dtype=np.int32)
        m = np.zeros(nnz.value, dtype=np.float64)
        k = np.zeros(nnz.value, dtype=np.float64)
        _lib.check(L.wae_p1_get(h, rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                                m.ctypes.data_as(dp), k.ctypes.data_as(dp)))
    finally:
        L.wae_p1_free(h)
    shape = (n.value, n.value)
    return (sp.csr_matrix((m.astype(dtype), col, rowptr), shape=shape), sp.csr_matrix((k.astype(dtype), col.copy(), rowptr.copy()), shape=shape))


def _take_csr(L, h, dtype):
    """copy a P1 handle out as a scipy CSR matrix (values = the `mass` array) and free it"""
    dp = C.POINTER(C.c_double)
    try:
        n, nnz = C.c_int64(0), C.c_int64(0)
        _lib.check(L.wae_p1_info(h, C.byref(n), C.byref(nnz)))
        rowptr = np.zeros(n.value + 1, dtype=np.int32)
        col = np.zeros(nnz.value, dtype=np.int32)
        v = np.zeros(nnz.value, dtype=np.float64)
        _lib.check(L.wae_p1_get(h, rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)), v.ctypes.data_as(dp), None))
    finally:
        L.wae_p1_free(h)
    return sp.csr_matrix((v.astype(dtype), col, rowptr), shape=(n.value, n.value))


def assemble_p1_boundary(points, tris, c_tri=None, device=0):
    """Boundary mass term of an admittance boundary on the device (wae_p1_assemble_boundary):
    C = -i·c·|e1×e2|·(1+δ_ab)/24 per boundary triangle (src/Helmholtz.jl:443-463, src/FEM/FEM.jl:435-441).  Returns C (complex CSR)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    tt = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    cc = None if c_tri is None else np.ascontiguousarray(c_tri, dtype=np.float64)
    L = _lib.lib()
    h = C.c_void_p()
    dp = C.POINTER(C.c_double)
    _lib.check(L.wae_p1_assemble_boundary(int(device), pts.shape[0], pts.ctypes.data_as(dp), tt.shape[0], tt.ctypes.data_as(C.POINTER(C.c_int32)),
                                          None if cc is None else cc.ctypes.data_as(dp), C.byref(h)))
    return -1j * _take_csr(L, h, np.complex128)


def assemble_p1_flame(points, tets, flame_tets, ref_tet, n_ref, nglobal_scaled, device=0):
    """Flame operator Q = Σ_flame S ⊗ g on the device (wae_p1_assemble_flame; src/Helmholtz.jl:292-344,464-487):
    ``nglobal_scaled`` = (γ-1)/ρ·Q02U0, the library divides by the flame volume it sums itself.  Returns (Q, V_flame)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    tt = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    fl = np.ascontiguousarray(flame_tets, dtype=np.int32)
    nr = np.ascontiguousarray(n_ref, dtype=np.float64)
    L = _lib.lib()
    h = C.c_void_p()
    vol = C.c_double(0.0)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    _lib.check(L.wae_p1_assemble_flame(int(device), pts.shape[0], pts.ctypes.data_as(dp), tt.shape[0], tt.ctypes.data_as(ip), len(fl), fl.ctypes.data_as(ip),
                                       int(ref_tet), nr.ctypes.data_as(dp), float(nglobal_scaled), C.byref(h), C.byref(vol)))
    return _take_csr(L, h, np.complex128), vol.value


def discrete_adjoint_shape_sensitivity(points, tets, c_tet, surface_points, sol, L, bnd_tris=None, bnd_c=None, Y=None, h=1e-9,
                                       device=0, flame=None, v_ext=None):
    """sens = discrete_adjoint_shape_sensitivity(...)   (src/shape_sensitivity.jl:16-141, full mesh, P1)

    Sensitivity of the eigenvalue ``sol.params[sol.eigval]`` to a displacement of every point in ``surface_points`` along
    x, y, z: -v_adj' (dL/dx) v with v'v = 1 and v_adj' L'(ω) v = 1 (the normalisation uses ``L``, the device-backed
    family).  The interior operators M, K (all tetrahedra touching the point) and, if given, the admittance boundary
    ω·Y·C (``bnd_tris``: boundary triangles, ``bnd_c``: speed of sound at each, ``Y``) take part, and -- round 3 -- the flame
    term: ``flame`` = dict(flame_tets, ref_tet, n_ref, nglobal_scaled[, coeff]) as produced by ``flame_description`` / the
    fixtures (``coeff``: the flame term's scalar n·exp(-iωτ) at ω; default: read from ``L``'s term with operator "Q").  As
    in the reference the flame domain is re-discretised REDUCED to the tetrahedra at the point, its volume included
    (``nlocal = nglobal_scaled / V_reduced``, Helmholtz.jl:325 on the reduced mesh of shape_sensitivity.jl:62-80).
    ``v_ext`` = (v, v_adj) already normalised and given on ``points`` (the unit-cell route extends the sector vectors to the
    image points and passes them here).  Returns a complex array (3, len(surface_points))."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    tt = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
    cc = None if c_tet is None else np.ascontiguousarray(c_tet, dtype=np.float64)
    sp_ = np.asarray(surface_points, dtype=np.int64)
    w0 = complex(sol.params[sol.eigval])
    if v_ext is not None:
        v, va = (np.asarray(x, dtype=np.complex128) for x in v_ext)
    else:
        v = np.asarray(sol.v, dtype=np.complex128)
        v = v / np.sqrt(np.vdot(v, v))
        saved = (L.active, L.mode, dict(L.params))
        L.active, L.mode = [L.eigval], "all"
        try:
            va = np.asarray(sol.v_adj, dtype=np.complex128)
            va = va / np.conj(np.vdot(va, L(w0, 1) @ v))
        finally:
            L.active, L.mode, L.params = saved
    v, va = np.ascontiguousarray(v), np.ascontiguousarray(va)
    # (point, simplex) adjacency pairs, in surface-point order
    lut = np.full(pts.shape[0], -1, dtype=np.int64)
    lut[sp_] = np.arange(len(sp_))
    loc_t = lut[tt]
    it, ia = np.nonzero(loc_t >= 0)
    pair_tet = it.astype(np.int32)
    pair_pt_t = tt[it, ia].astype(np.int32)
    own_t = loc_t[it, ia]
    out_t = np.zeros((len(pair_tet), 3), dtype=np.complex128)
    tri = None
    npair_s = 0
    pair_tri = pair_pt_s = own_s = None
    out_s = np.zeros((0, 3), dtype=np.complex128)
    if bnd_tris is not None and len(bnd_tris):
        tri = np.ascontiguousarray(bnd_tris, dtype=np.int32).reshape(-1, 3)
        loc_s = lut[tri]
        js, ja = np.nonzero(loc_s >= 0)
        pair_tri, pair_pt_s, own_s = js.astype(np.int32), tri[js, ja].astype(np.int32), loc_s[js, ja]
        npair_s = len(pair_tri)
        out_s = np.zeros((npair_s, 3), dtype=np.complex128)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    om = np.array([w0.real, w0.imag])
    wy = complex(w0 * (Y if Y is not None else 0.0))
    omy = np.array([wy.real, wy.imag])
    bc = np.ascontiguousarray(bnd_c, dtype=np.float64) if npair_s else None

    def P(a, t):
        return None if a is None else a.ctypes.data_as(t)
    _lib.check(_lib.lib().wae_p1_shape_sensitivity(
        int(device), pts.shape[0], P(pts, dp), P(tt, ip), P(cc, dp), len(pair_tet), P(pair_pt_t, ip), P(pair_tet, ip),
        P(tri, ip) if npair_s else None, P(bc, dp), npair_s, P(pair_pt_s, ip) if npair_s else None, P(pair_tri, ip) if npair_s else None,
        tt.shape[0], 0 if tri is None else tri.shape[0], P(om, dp), P(omy, dp),
        v.view(np.float64).ctypes.data_as(dp), va.view(np.float64).ctypes.data_as(dp), float(h),
        out_t.view(np.float64).ctypes.data_as(dp) if len(pair_tet) else None, out_s.view(np.float64).ctypes.data_as(dp) if npair_s else None))
    sens = np.zeros((3, len(sp_)), dtype=np.complex128)
    np.add.at(sens.T, own_t, out_t)                                   # per point, in pair order: deterministic
    if npair_s:
        np.add.at(sens.T, own_s, out_s)
    if flame is not None:
        sens += _flame_shape_part(pts, tt, sp_, lut, v, va, w0, L, flame, h, device)
    return sens


def _flame_shape_part(pts, tt, sp_, lut, v, va, w0, L, flame, h, device):
    """-v_adj' n e^{-iωτ} (Q₊ - Q₋)/(2h) v per surface point and coordinate (wae_p1_shape_sensitivity_flame + the per-point sums)."""
    fl = np.asarray(flame["flame_tets"], dtype=np.int64)
    ref = int(flame["ref_tet"])
    nr = np.ascontiguousarray(flame["n_ref"], dtype=np.float64)
    coeff = flame.get("coeff")
    if coeff is None:
        k = [i for i, t in enumerate(L.terms) if t.operator == "Q"]
        assert len(k) == 1, "the family needs exactly one term with operator 'Q' (or pass flame['coeff'])"
        saved = (L.active, L.mode, dict(L.params))
        L.active, L.mode = [L.eigval], "all"
        try:
            coeff = complex(L.coefficients(w0)[k[0]])
        finally:
            L.active, L.mode, L.params = saved
    loc = lut[tt[fl]]                                                 # (nflame, 4): position of each node in surface_points, or -1
    it, ia = np.nonzero(loc >= 0)
    pair_tet = fl[it].astype(np.int32)
    pair_pt = tt[fl[it], ia].astype(np.int32)
    own = loc[it, ia]
    rsel = np.nonzero(lut[tt[ref]] >= 0)[0]
    pair_pt_r = tt[ref][rsel].astype(np.int32)
    own_r = lut[tt[ref]][rsel]
    npair, npr = len(pair_tet), len(pair_pt_r)
    det_pm = np.zeros((npair, 3, 2))
    ssum = np.zeros(npair, dtype=np.complex128)
    g_pm = np.zeros((npr, 3, 2), dtype=np.complex128)
    g0 = np.zeros(1, dtype=np.complex128)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)

    def P(a, t):
        return a.ctypes.data_as(t) if a.size else None
    _lib.check(_lib.lib().wae_p1_shape_sensitivity_flame(
        int(device), pts.shape[0], pts.ctypes.data_as(dp), tt.shape[0], tt.ctypes.data_as(ip), npair, P(pair_pt, ip), P(pair_tet, ip), ref, npr,
        P(pair_pt_r, ip), nr.ctypes.data_as(dp), v.view(np.float64).ctypes.data_as(dp), va.view(np.float64).ctypes.data_as(dp), float(h),
        P(det_pm, dp), P(ssum.view(np.float64), dp), P(g_pm.view(np.float64), dp), g0.view(np.float64).ctypes.data_as(dp)))
    ns = len(sp_)
    a_pm = np.zeros((ns, 3, 2), dtype=np.complex128)                  # v_adj' S± per point and coordinate
    V_pm = np.zeros((ns, 3, 2))                                       # volume of the point's flame tetrahedra
    np.add.at(a_pm, own, det_pm / 24.0 * ssum[:, None, None])
    np.add.at(V_pm, own, det_pm / 6.0)
    b_pm = np.full((ns, 3, 2), g0[0], dtype=np.complex128)            # sum_b grad(phi_b).n_ref v_b on the reference tetrahedron
    b_pm[own_r] = g_pm
    out = np.zeros((3, ns), dtype=np.complex128)
    has = V_pm[:, 0, 0] > 0                                           # points without a flame tetrahedron: empty domain, no term
    nl = np.zeros_like(V_pm)
    nl[has] = float(flame["nglobal_scaled"]) / V_pm[has]
    q = a_pm * (-nl * b_pm)                                           # v_adj' Q± v  (g = -nlocal grad.n_ref)
    out[:, has] = (-coeff * (q[has, :, 0] - q[has, :, 1]) / (2 * h)).T
    return out
