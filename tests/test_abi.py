"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/waehip.h declares (no compute without a GPU), and the product path fails loudly without a device."""
import os
import re
import shutil
import subprocess

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


def _header_prototypes():
    """name -> list of C parameter type strings, from include/waehip.h"""
    hdr = open(os.path.join(ROOT, "include", "waehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char \*)\s*\*?\s*(wae_[a-z_0-9]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        protos[m.group(1)] = [] if args == ["void"] else args
    return protos


def _julia_ccalls():
    """(name, [julia argument types]) of every ccall in julia/WAEHip.jl (signature tuples given inline or through `sig`)"""
    jl = open(os.path.join(ROOT, "julia", "WAEHip.jl"), encoding="utf-8").read()
    sigs = {m.group(1): m.group(2) for m in re.finditer(r"\n\s*(sig)\s*=\s*\((.*?)\)\n", jl, flags=re.S)}
    out = []
    for m in re.finditer(r"ccall\(\(:(wae_[a-z_0-9]+), libwaehip\),\s*(\w+),\s*(\(|sig\b)", jl):
        name = m.group(1)
        if m.group(3) == "sig":
            body = sigs["sig"]
        else:
            i = m.end()
            depth, j = 1, i
            while depth:
                depth += {"(": 1, ")": -1}.get(jl[j], 0)
                j += 1
            body = jl[i:j - 1]
        types, depth, cur = [], 0, ""
        for ch in body:
            if ch in "{(":
                depth += 1
            if ch in "})":
                depth -= 1
            if ch == "," and depth == 0:
                types.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            types.append(cur.strip())
        out.append((name, types))
    return out


def test_julia_ccall_signatures_match_the_header():
    """julia/WAEHip.jl cannot be executed here (no julia binary): its ccall signatures are checked mechanically against
    include/waehip.h -- same symbol, same number of arguments, pointer where the header has a pointer, same integer / float
    width where it has a scalar."""
    protos = _header_prototypes()
    calls = _julia_ccalls()
    assert len(calls) >= 30
    seen = set()
    for name, types in calls:
        assert name in protos, name
        cargs = protos[name]
        assert len(types) == len(cargs), (name, types, cargs)
        for jt, ct in zip(types, cargs):
            if "*" in ct:
                assert jt.startswith(("Ptr{", "Ref{")) or jt == "Cstring", (name, jt, ct)
            elif ct.startswith("int64_t"):
                assert jt == "Int64", (name, jt, ct)
            elif ct.startswith("int32_t"):
                assert jt in ("Int32", "Cint"), (name, jt, ct)
            elif ct.startswith("uint64_t"):
                assert jt == "UInt64", (name, jt, ct)
            elif ct.startswith("double"):
                assert jt == "Float64", (name, jt, ct)
            else:
                raise AssertionError((name, jt, ct))
        seen.add(name)
    for must in ("wae_family_create", "wae_spmv_sum", "wae_solver_setup", "wae_solve", "wae_solve_guess", "wae_beyn_moments",
                 "wae_beyn_moments_rb", "wae_beyn_moments_mgpu", "wae_eig_residuals", "wae_arnoldi_shiftinvert_batch", "wae_perturb",
                 "wae_slot_write", "wae_slot_read", "wae_slot_axpby", "wae_slot_forms", "wae_arnoldi_shiftinvert_slots", "wae_arnoldi_ritz_to_slot",
                 "wae_perturb_slots"):
        assert must in seen, must
    # the file is at least bracket-balanced (a cheap stand-in for a parser)
    jl = open(os.path.join(ROOT, "julia", "WAEHip.jl"), encoding="utf-8").read()
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"#=.*?=#", "", jl, flags=re.S))
    code = "\n".join(ln.split("#")[0] for ln in code.split("\n"))
    for a, b in ("()", "[]", "{}"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))


def test_every_product_export_is_bound_from_julia():
    """The host the north star names is Julia: every entry of include/waehip.h that is not a measurement helper (wae_bench_*) or the
    test hook (wae_debug_*) must have a `ccall` in julia/WAEHip.jl -- a feature that exists only in the Python mirror is not a
    drop-in."""
    protos = _header_prototypes()
    bound = {name for name, _ in _julia_ccalls()}
    need = {n for n in protos if not n.startswith(("wae_bench_", "wae_debug_"))}
    assert len(need) >= 30
    assert not sorted(need - bound), sorted(need - bound)
    assert set(protos) == set(_lib.EXPORTS)
    # ... and the batched refinement + the solve driver that uses it exist on the Julia side
    jl = open(os.path.join(ROOT, "julia", "WAEHip.jl"), encoding="utf-8").read()
    for fn in ("function householder_many(", "function solve_batched(", "function eigs_many(", "function conjugate_span_start(",
               "function assemble_p1(", "function assemble_p1_boundary(", "function assemble_p1_flame(",
               "function discrete_adjoint_shape_sensitivity_p1(", "function discrete_adjoint_shape_sensitivity_p1_flame(",
               "function spmv_cols(", "function spmv_multi(", "function rb_export(", "function rb_import(",
               # the device-resident Newton-type refinement (round 4) and the host-memory form it is checked against
               "function householder_many_host(", "function eigs_many_slots(", "function eigval_series_slots(", "function slot_write(",
               "function slot_read(", "function slot_axpby(", "function slot_forms(", "function arnoldi_slots(", "function ritz_to_slot("):
        assert fn in jl, fn


def test_fused_galerkin_product_is_bit_identical(tmp_path):
    """tests/abi/amg_check.cpp (host only): `galerkin_pair` -- the triple products R K P and R M P of the multigrid set-up formed in one
    traversal of the shared pattern -- against two separate `galerkin` calls, bit for bit, for several thread counts; and the one-pass
    shape matrix of `amg_setup` against its two-step form: the same hierarchy."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path / "amg_check")
    csrc = os.path.join(ROOT, "wavesandeigenvalues.jl_amd", "csrc")
    cmd = ["hipcc", "-O1", "-std=c++17", "-I", csrc, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "amg_check.cpp"),
           os.path.join(csrc, "amg.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "amg_check ok 24" in r.stdout, r.stdout + r.stderr
