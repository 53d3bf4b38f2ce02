"""ctypes binding of libwaehip.so (include/waehip.h).  No CPU fallback: importing works without a GPU
(so that the ABI can be inspected), but every compute entry point needs the HIP library and a device."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAE_LIB_PATH") or os.path.join(_HERE, "csrc", "libwaehip.so")     # (WAE_LIB_PATH: an A/B build of the library)

WAE_OK, WAE_WARN_MAXITER, WAE_WARN_STAGNATION = 0, 1, 2
WAE_ERR_INVALID, WAE_ERR_BREAKDOWN, WAE_ERR_EIGS, WAE_ERR_NAN, WAE_ERR_HIP = -1, -2, -3, -4, -5
OP_N, OP_T, OP_C = 0, 1, 2
CSC, CSR = 0, 1


class SolveInfo(C.Structure):
    _fields_ = [("iters_max", C.c_int32), ("iters_total", C.c_int32), ("n_unconverged", C.c_int32),
                ("levels", C.c_int32), ("relres_max", C.c_double), ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class WaeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libwaehip error {code}: {msg}")
        self.code = code


_lib = None

# every symbol include/waehip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "wae_last_error", "wae_device_count", "wae_version", "wae_family_create", "wae_family_create_opts", "wae_family_destroy",
    "wae_family_info", "wae_family_spmv_bytes", "wae_spmv_sum", "wae_spmv_sum_cols", "wae_spmv_sum_multi", "wae_solver_setup",
    "wae_solve", "wae_solve_guess", "wae_beyn_moments", "wae_beyn_moments_mgpu", "wae_beyn_moments_rb", "wae_rb_export", "wae_rb_import", "wae_eig_residuals", "wae_arnoldi_shiftinvert", "wae_arnoldi_shiftinvert_batch", "wae_perturb", "wae_slot_write", "wae_slot_read", "wae_slot_axpby", "wae_slot_forms", "wae_arnoldi_shiftinvert_slots", "wae_arnoldi_ritz_to_slot", "wae_perturb_slots", "wae_p1_assemble", "wae_p1_assemble_boundary", "wae_p1_assemble_flame", "wae_p1_info", "wae_p1_get", "wae_p1_free", "wae_p1_shape_sensitivity", "wae_p1_shape_sensitivity_flame", "wae_bench_spmv", "wae_bench_spmv_level", "wae_bench_triad", "wae_debug_spmv",
]


def lib():
    """Load libwaehip.so (built by __graft_entry__.build() / csrc/Makefile); raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vpp = C.POINTER(C.c_void_p)
    L.wae_last_error.restype = C.c_char_p
    L.wae_version.restype = C.c_char_p
    L.wae_device_count.argtypes = [C.POINTER(C.c_int)]
    L.wae_family_create.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    vpp, vpp, vpp, C.c_int32]
    L.wae_family_create_opts.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         vpp, vpp, vpp, C.c_int32, dp, C.c_int32]
    L.wae_family_destroy.argtypes = [C.c_void_p]
    L.wae_family_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.wae_family_spmv_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32]
    L.wae_family_spmv_bytes.restype = C.c_int64
    L.wae_spmv_sum.argtypes = [C.c_void_p, dp, dp, dp, C.c_int32, C.c_int32]
    L.wae_spmv_sum_cols.argtypes = [C.c_void_p, dp, C.c_int32, dp, dp, C.c_int32, C.c_int32]
    L.wae_spmv_sum_multi.argtypes = [C.c_void_p, dp, dp, dp]
    L.wae_solver_setup.argtypes = [C.c_void_p, dp, dp, C.c_int32]
    L.wae_solve.argtypes = [C.c_void_p, dp, C.c_int32, dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                            C.POINTER(SolveInfo)]
    L.wae_solve_guess.argtypes = [C.c_void_p, dp, C.c_int32, dp, dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                  C.POINTER(SolveInfo)]
    L.wae_beyn_moments.argtypes = [C.c_void_p, C.c_int32, dp, dp, dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                   dp, C.c_uint64, C.POINTER(SolveInfo)]
    L.wae_beyn_moments_rb.argtypes = [C.c_void_p, C.c_int32, dp, dp, dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_uint64, dp, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(SolveInfo)]
    L.wae_beyn_moments_mgpu.argtypes = [vpp, C.c_int32, C.c_int32, dp, dp, dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, dp,
                                        C.POINTER(SolveInfo)]
    ip = C.POINTER(C.c_int32)
    L.wae_rb_export.argtypes = [C.c_void_p, ip, ip, ip, ip, dp, dp]
    L.wae_rb_import.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, ip, dp, dp]
    L.wae_eig_residuals.argtypes = [C.c_void_p, C.c_int32, dp, dp, C.c_uint64, dp]
    L.wae_arnoldi_shiftinvert.argtypes = [C.c_void_p, dp, dp, C.c_int32, dp, C.c_int32, C.c_double, C.c_int32, dp, dp,
                                          C.POINTER(SolveInfo)]
    L.wae_arnoldi_shiftinvert_batch.argtypes = [C.c_void_p, C.c_int32, dp, dp, C.c_int32, dp, C.c_int32, C.c_double, C.c_int32, C.c_double,
                                                dp, dp, C.POINTER(SolveInfo)]
    L.wae_perturb.argtypes = [C.c_void_p, dp, C.c_int32, dp, dp, C.c_int32, dp, C.c_double, C.c_int32, dp, dp,
                              C.POINTER(SolveInfo)]
    ip = C.POINTER(C.c_int32)
    L.wae_slot_write.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp]
    L.wae_slot_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp]
    L.wae_slot_axpby.argtypes = [C.c_void_p, C.c_int32, C.c_int32, ip, C.c_int32, ip, dp, dp, C.c_int32]
    L.wae_slot_forms.argtypes = [C.c_void_p, C.c_int32, dp, C.c_int32, C.c_int32, ip, C.c_int32, ip, dp]
    L.wae_arnoldi_shiftinvert_slots.argtypes = [C.c_void_p, C.c_int32, dp, dp, C.c_int32, C.c_int32, ip, C.c_int32, C.c_double, C.c_int32, C.c_double,
                                                dp, C.POINTER(SolveInfo)]
    L.wae_arnoldi_ritz_to_slot.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp, C.c_int32, ip, C.c_int32]
    L.wae_perturb_slots.argtypes = [C.c_void_p, dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, C.c_double, C.c_int32, dp, dp,
                                    C.POINTER(SolveInfo)]
    L.wae_p1_assemble.argtypes = [C.c_int32, C.c_int64, dp, C.c_int64, C.POINTER(C.c_int32), dp, C.POINTER(C.c_void_p)]
    L.wae_p1_assemble_boundary.argtypes = [C.c_int32, C.c_int64, dp, C.c_int64, C.POINTER(C.c_int32), dp, C.POINTER(C.c_void_p)]
    L.wae_p1_assemble_flame.argtypes = [C.c_int32, C.c_int64, dp, C.c_int64, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int32), C.c_int32, dp,
                                        C.c_double, C.POINTER(C.c_void_p), dp]
    L.wae_p1_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wae_p1_get.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp]
    L.wae_p1_free.argtypes = [C.c_void_p]
    i32p = C.POINTER(C.c_int32)
    L.wae_p1_shape_sensitivity.argtypes = [C.c_int32, C.c_int64, dp, i32p, dp, C.c_int64, i32p, i32p, i32p, dp, C.c_int64, i32p, i32p,
                                           C.c_int64, C.c_int64, dp, dp, dp, dp, C.c_double, dp, dp]
    L.wae_p1_shape_sensitivity_flame.argtypes = [C.c_int32, C.c_int64, dp, C.c_int64, i32p, C.c_int64, i32p, i32p, C.c_int32, C.c_int64, i32p, dp, dp, dp,
                                                 C.c_double, dp, dp, dp, dp]
    L.wae_bench_spmv.argtypes = [C.c_void_p, dp, C.c_int32, C.c_int32, dp]
    L.wae_bench_triad.argtypes = [C.c_int32, C.c_int64, C.c_int32, dp]
    L.wae_bench_spmv_level.argtypes = [C.c_void_p, dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, C.POINTER(C.c_int64)]
    L.wae_debug_spmv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32, dp, dp, dp, dp, C.c_int32, C.c_int32, C.c_double,
                                 C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    _lib = L
    return L


def check(code, warn_ok=True):
    if code < 0 or (code > 0 and not warn_ok):
        raise WaeError(code, lib().wae_last_error().decode(errors="replace"))
    return code


def zptr(a):
    """pointer to the interleaved (re,im) doubles of a complex128 array (must be contiguous)."""
    assert a.dtype == np.complex128
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count():
    n = C.c_int(0)
    check(lib().wae_device_count(C.byref(n)))
    return n.value
