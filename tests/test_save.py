"""Interchange files: the reference's text formats for LinearOperatorFamily / Solution
(src/NLEVP/LinOpFam.jl:196-294, src/NLEVP/save.jl:2-135, src/NLEVP/toml.jl:10-63) and the binary container."""
import numpy as np
import pytest
import scipy.sparse as sp

import wae_amd  # noqa: F401
from oracle import fixtures as F
from wae_amd.helmholtz import annulus
from wae_amd.helmholtz.bloch import bloch_family
from wae_amd.helmholtz.family import helmholtz_family
from wae_amd.nlevp import Solution, pade_
from wae_amd.nlevp.save import load_family, parse_julia, read_sol, read_toml, save

RNG = np.random.default_rng(21)


def _same_family(A, B, zs=(2 * np.pi * (300 + 10j),)):
    assert len(A.terms) == len(B.terms)
    assert A.eigval == B.eigval and A.auxval == B.auxval
    for k in A.params:
        a, b = A.params[k], B.params[k]
        assert a == b or (np.isnan(a.real) and np.isnan(b.real)) or (np.isinf(a.real) and np.isinf(b.real))
    for ta, tb in zip(A.terms, B.terms):
        assert ta.symbol == tb.symbol and ta.operator == tb.operator and ta.params == tb.params
        assert abs(sp.csr_matrix(ta.coeff) - sp.csr_matrix(tb.coeff)).max() == 0
    for z in zs:
        for args in ((z,), (z, 1), (z, 2)):
            ca, cb = A.coefficients(*args), B.coefficients(*args)
            for x, y in zip(ca, cb):
                assert (x is None and y is None) or abs(complex(x or 0) - complex(y or 0)) <= 1e-14 * max(1.0, abs(complex(x or 0)))


def test_julia_literals():
    assert parse_julia("1.0 + 2.0im") == 1 + 2j
    assert parse_julia("-1.5e-3-2.0im") == -1.5e-3 - 2j
    assert parse_julia("Inf + 0.0im") == complex(np.inf, 0)
    v = parse_julia("NaN + NaN*im")
    assert np.isnan(v.real) and np.isnan(v.imag)
    assert parse_julia("UInt32[0x00000001, 0x0000000a]") == [1, 10]
    assert parse_julia("Complex{Float64}[1.0+0.0im,0.5-2.0im,]") == [1 + 0j, 0.5 - 2j]
    # negative zero: the reference prints "+" whenever imag(v) >= 0, which holds for -0.0 (save.jl:73-78, LinOpFam.jl:283-287),
    # so negated terms (-M, -Q) and conj'd real vectors routinely produce "+-0.0im"
    v = parse_julia("1.0+-0.0im")
    assert v == 1 + 0j and np.signbit(v.imag)
    assert parse_julia("Complex{Float64}[-1.0+-0.0im,0.0+-2.5im,2.0+0.0im,]") == [-1 + 0j, -2.5j, 2 + 0j]
    assert parse_julia("((:ω,), (:ω, :τ))") == (("ω",), ("ω", "τ"))
    assert parse_julia("()") == ()
    assert parse_julia('"n*exp(-iωτ)"') == "n*exp(-iωτ)"
    f = parse_julia("(pow2,generate_exp_az(0.0+0.5im),)")
    assert f[0] == "pow2" and f[1][0] == "generate_exp_az" and f[1][1] == (0.5j,)
    assert parse_julia("[(:ω,1.0 + 0.0im),(:λ,Inf + 0.0im),]") == [("ω", 1 + 0j), ("λ", complex(np.inf, 0))]


def test_reference_written_family_file_loads(tmp_path):
    """a file laid out line by line as LinOpFam.jl:236-294 writes it (UInt32 index arrays print as hex in Julia)"""
    txt = """# LinearOperatorFamily version 0
#2021-03-01T10:00:00.000
#+ω^2*M+K
params=[(:ω,0.0 + 0.0im),
(:λ,Inf + 0.0im),
(:τ,0.001 + 0.0im),
]
eigval=:ω
auxval=:λ
[terms]
\t[terms.1]
\tfunctions=(pow2,)
\tsymbol="ω^2"
\tparams=((:ω,),)
\toperator="M"
\tsize=[3,3]
\t\t[terms.1.sparse_matrix]
\t\tI=UInt32[0x00000001, 0x00000002, 0x00000003]
\t\tJ=UInt32[0x00000001, 0x00000002, 0x00000003]
\t\tV=Complex{Float64}[1.0+0.0im,2.0+0.0im,3.0+0.0im,]

\t[terms.2]
\tfunctions=(pow1,exp_delay,)
\tsymbol="n*exp(-iωτ)"
\tparams=((:n,), (:ω, :τ))
\toperator="Q"
\tsize=[3,3]
\t\t[terms.2.sparse_matrix]
\t\tI=[1, 3]
\t\tJ=[2, 1]
\t\tV=Complex{Float64}[0.5-1.5im,-2.0+0.0im,]

\t[terms.3]
\tfunctions=()
\tsymbol=""
\tparams=()
\toperator="K"
\tsize=[3,3]
\t\t[terms.3.sparse_matrix]
\t\tI=[1, 2]
\t\tJ=[1, 1]
\t\tV=Complex{Float64}[-1.0+0.0im,4.0+0.0im,]

"""
    p = tmp_path / "ref_family.toml"
    p.write_text(txt, encoding="utf-8")
    L = load_family(str(p))
    assert [t.operator for t in L.terms] == ["M", "Q", "K"] and L.eigval == "ω" and L.auxval == "λ"
    assert np.isnan(L.params["n"].real) and L.params["τ"] == 0.001          # n was never given a value (push!, :333-338)
    L.params["n"] = 2.0
    z = 3.0 + 1.0j
    c = L.coefficients(z)
    assert abs(c[0] - z * z) < 1e-15 and abs(c[1] - 2.0 * np.exp(-1j * z * 0.001)) < 1e-15 and c[2] == 1
    A = sum(ck * t.coeff for ck, t in zip(c, L.terms)).toarray()
    assert abs(A[0, 1] - c[1] * (0.5 - 1.5j)) < 1e-15 and abs(A[1, 0] - 4.0) < 1e-15 and abs(A[2, 2] - 3 * z * z) < 1e-14


def test_family_round_trip_text_and_binary(tmp_path):
    L = helmholtz_family(F.rijke_terms(), n=0.7, tau=1.3e-3)
    for binary in (False, True):
        p = str(tmp_path / f"rijke_{binary}")
        save(p, L, binary=binary)
        _same_family(L, load_family(p))
    # a Bloch family: closures written as constructor expressions (generate_exp_az(…), generate_gz_hz(…, …))
    cell = annulus.build_unit_cell(grid=(3, 6, 3), DOS=8, tau=2e-4)
    cell["naxis"] = 0
    Lb = bloch_family(cell, b=3)
    for binary in (False, True):
        p = str(tmp_path / f"bloch_{binary}")
        save(p, Lb, binary=binary)
        Lr = load_family(p)
        assert Lr.params["b"] == 3
        _same_family(Lb, Lr, zs=(2 * np.pi * (300 + 10j), 100.0))
    head = open(str(tmp_path / "bloch_False"), encoding="utf-8").read(400)
    assert head.startswith("# LinearOperatorFamily version 0\n#") and "params=[(:ω," in head


def test_unknown_function_is_refused_not_evaluated(tmp_path):
    p = tmp_path / "bad.toml"
    p.write_text('params=[(:ω,0.0 + 0.0im),\n]\neigval=:ω\nauxval=:ω\n[terms]\n\t[terms.1]\n\tfunctions=(run_me,)\n\tsymbol=""\n'
                 '\tparams=((:ω,),)\n\toperator="M"\n\tsize=[1,1]\n\t\t[terms.1.sparse_matrix]\n\t\tI=[1]\n\t\tJ=[1]\n'
                 '\t\tV=Complex{Float64}[1.0+0.0im,]\n', encoding="utf-8")
    with pytest.raises(KeyError):
        load_family(str(p))
    L = load_family(str(p), functions={"run_me": lambda z, k=0: 7.0})
    assert L.coefficients(1.0)[0] == 7.0


def test_solution_round_trip(tmp_path):
    d = 7
    params = {"ω": 1700.0 + 35.5j, "λ": 1e-13 - 2e-14j, "τ": 0.001 + 0j, "Y": complex(np.inf, 0)}
    sol = Solution(params, RNG.standard_normal(d) + 1j * RNG.standard_normal(d), RNG.standard_normal(d) - 1j * RNG.standard_normal(d), "ω", "λ")
    sol.eigval_pert["τ/Taylor"] = RNG.standard_normal(6) + 1j * RNG.standard_normal(6)
    sol.v_pert["τ/Taylor"] = [RNG.standard_normal(d) + 1j * RNG.standard_normal(d) for _ in range(6)]
    pade_(sol, "τ", 2, 2, vector=True)
    p = str(tmp_path / "sol.toml")
    save(p, sol)
    back = read_sol(p)
    assert back.eigval == "ω" and set(back.params) == set(params)
    for k, v in params.items():
        assert back.params[k] == v
    assert np.array_equal(back.v, sol.v) and np.array_equal(back.v_adj, sol.v_adj)
    assert set(back.eigval_pert) == {"τ/Taylor", "τ/[2/2]"} and set(back.v_pert) == {"τ/Taylor", "τ/[2/2]"}
    assert np.array_equal(back.eigval_pert["τ/Taylor"], sol.eigval_pert["τ/Taylor"])
    for a, b in zip(back.eigval_pert["τ/[2/2]"], sol.eigval_pert["τ/[2/2]"]):
        assert np.array_equal(a, b)
    for a, b in zip(back.v_pert["τ/Taylor"], sol.v_pert["τ/Taylor"]):
        assert np.array_equal(a, b)
    for A, B in zip(back.v_pert["τ/[2/2]"], sol.v_pert["τ/[2/2]"]):
        assert np.array_equal(np.asarray(A), np.asarray(B))
    # the loaded object evaluates like the original (Padé approximant of the eigenvalue, LinOpFam.jl:684-699)
    assert back("τ", 0.0012, 2, 2) == sol("τ", 0.0012, 2, 2)
    # file layout of save.jl:2-20, including its stray "]" after v
    lines = open(p, encoding="utf-8").read().split("\n")
    assert lines[0] == "# Solution version 0" and lines[2].startswith("params=[(:ω,1700.0 + 35.5im),")
    k = next(i for i, ln in enumerate(lines) if ln.startswith("v="))
    assert lines[k + 1] == "]" and lines[k + 2].startswith("v_adj=[")
    assert isinstance(read_toml(p)["/eigval_pert"]["/τ/[2/2]"]["den"], list)


def test_binary_container_as_the_julia_writer_lays_it_out(tmp_path):
    """julia/WAEHip.jl save_family_bin: 1-based CSC arrays, Infinity/NaN tokens in the JSON header, 8-byte alignment"""
    import struct
    A = sp.csc_matrix(np.array([[1.0, 0, 2.0j], [0, 3.0, 0], [4.0, 0, 5.0]]))
    head = ('{"version": 1, "eigval": "ω", "auxval": "λ", "active": ["ω"], "mode": "all", "params": {"ω": [0.0, 0.0], '
            '"λ": [Infinity, 0.0], "τ": [NaN, NaN]}, "terms": [{"symbol": "ω^2", "operator": "M", "functions": ["pow2"], '
            '"params": [["ω"]], "m": 3, "n": 3, "nnz": 5, "base": 1}, {"symbol": "", "operator": "K", "functions": [], '
            '"params": [], "m": 3, "n": 3, "nnz": 5, "base": 1}]}').encode("utf-8")
    p = tmp_path / "jl.waefam"
    with open(p, "wb") as f:
        f.write(b"WAEFAM1\n" + struct.pack("<Q", len(head)) + head)
        for scale in (1.0, -2.0):
            for arr, dt in ((A.indptr + 1, np.int64), (A.indices + 1, np.int64), (A.data * scale, np.complex128)):
                f.write(b"\0" * (-f.tell() % 8))
                f.write(np.asarray(arr, dtype=dt).tobytes())
    from wae_amd.nlevp import LinearOperatorFamily
    L = LinearOperatorFamily(str(p))
    assert np.isinf(L.params["λ"].real) and np.isnan(L.params["τ"].real) and L.eigval == "ω"
    assert abs(L.terms[0].coeff - A).max() == 0 and abs(L.terms[1].coeff + 2 * A).max() == 0
    assert L.coefficients(2.0) [0] == 4.0


def test_solution_file_with_negative_zero_imaginary_parts_loads(tmp_path):
    """a Solution file as save.jl:2-20,71-82 writes it for a conj'd real vector: every entry reads ``x+-0.0im``"""
    txt = """# Solution version 0
#2021-03-01T10:00:00.000
params=[(:ω,1700.0 + 35.5im),
(:λ,0.0 + 0.0im),
]
eigval=:ω
v=[1.0+-0.0im,-2.0+-0.0im,0.5+0.25im,]
]
v_adj=[1.0+0.0im,-2.0+0.0im,0.5-0.25im,]
"""
    p = tmp_path / "neg0.toml"
    p.write_text(txt, encoding="utf-8")
    sol = read_sol(str(p))
    assert np.array_equal(sol.v, np.array([1, -2, 0.5 + 0.25j])) and np.all(np.signbit(sol.v.imag[:2]))
    assert np.array_equal(sol.v_adj, np.array([1, -2, 0.5 - 0.25j]))
