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
