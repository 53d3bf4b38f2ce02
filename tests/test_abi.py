"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/waehip.h declares (no compute without a GPU), and the product path fails loudly without a device."""
import os
import re

import numpy as np
import pytest

import wae_amd
from wae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "waehip.h")).read()
    declared = set(re.findall(r"\b(wae_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("wae_family")   # the opaque type
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "waehip.h")).read()
    assert "torch" not in hdr.replace("no torch", "") and "at::" not in hdr and "hipStream_t" not in hdr


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import scipy.sparse as sp
    from wae_amd.nlevp import LinearOperatorFamily, Term, pow1
    L = LinearOperatorFamily()
    L.push(Term(sp.identity(4, dtype=complex, format="csr"), (pow1,), (("λ",),), "λ", "I"))
    with pytest.raises(_lib.WaeError):
        L(1.0) @ np.ones(4, dtype=complex)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "wavesandeigenvalues.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def _build_c_caller(out, link):
    """gcc (C99, pedantic) on tests/abi/abi_caller.c against include/waehip.h; link=True also links libwaehip.so"""
    import subprocess
    src = os.path.join(ROOT, "tests", "abi", "abi_caller.c")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), src, "-o", out]
    libdir = os.path.dirname(_lib.LIB_PATH)
    if link:
        cmd += ["-L", libdir, "-lwaehip", "-Wl,-rpath," + libdir]
    else:
        cmd[1:1] = ["-c"]
    subprocess.check_call(cmd)
    return out


def test_header_is_valid_c99_and_the_info_struct_layout_matches_ctypes(tmp_path):
    """the header is consumed by C callers (Julia's ccall follows the C layout rules): it must compile as C, and
    wae_solve_info / the code constants must be what the ctypes mirror (and julia/WAEHip.jl) assume"""
    import ctypes as C
    import subprocess
    _build_c_caller(str(tmp_path / "abi_caller.o"), link=False)
    exe = _build_c_caller(str(tmp_path / "abi_caller"), link=True)
    out = subprocess.check_output([exe, "layout"], text=True).split("\n")
    tok = out[0].split()
    lay = {tok[i]: int(tok[i + 1]) for i in range(0, len(tok), 2)}
    assert lay["sizeof"] == C.sizeof(_lib.SolveInfo)
    for name, _ in _lib.SolveInfo._fields_:
        assert lay[name] == getattr(_lib.SolveInfo, name).offset, name
    codes = [int(x) for x in out[1].split()[1:9]]
    assert codes == [_lib.WAE_OK, _lib.WAE_WARN_MAXITER, _lib.WAE_WARN_STAGNATION, _lib.WAE_ERR_INVALID, _lib.WAE_ERR_BREAKDOWN,
                     _lib.WAE_ERR_EIGS, _lib.WAE_ERR_NAN, _lib.WAE_ERR_HIP]
    t2 = out[1].split()
    assert [int(x) for x in t2[10:13]] == [_lib.OP_N, _lib.OP_T, _lib.OP_C] and [int(x) for x in t2[14:16]] == [_lib.CSC, _lib.CSR]
    # the Julia glue declares the same struct: four Int32 then two Float64
    jl = open(os.path.join(ROOT, "julia", "WAEHip.jl"), encoding="utf-8").read()
    m = re.search(r"struct SolveInfo(.*?)end", jl, re.S)
    assert m and re.findall(r"::(\w+)", m.group(1)) == ["Int32"] * 4 + ["Float64"] * 2
