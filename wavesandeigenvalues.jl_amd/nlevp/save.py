"""Operator and solution interchange.

Two formats:

* the reference's own text files -- ``save(fname, L::LinearOperatorFamily)`` (src/NLEVP/LinOpFam.jl:236-294, read back
  by ``LinearOperatorFamily(fname)`` :196-225) and ``save(fname, sol::Solution)`` / ``read_sol`` (src/NLEVP/save.jl:2-135)
  on top of the "Julia-enriched TOML" reader src/NLEVP/toml.jl:10-63.  The reference reads them by ``eval``-ing every
  right-hand side as Julia code; here a small recursive-descent parser accepts the literal subset those writers emit
  (numbers, complex numbers ``a+bim``, ``:symbols``, "strings", tuples, (typed) arrays, function names and constructor
  calls such as ``generate_exp_az(0.0+0.26im)``) and NOTHING is evaluated.  Files written here load in the reference
  (results go back as ``Solution`` files), files written by the reference load here.
* a binary container for large operators (a 1M-DoF family is ~1 GB of text): magic ``WAEFAM1\\n``, a length-prefixed JSON
  header (params, eigval/auxval/active/mode, per term: symbol, operator, function specs, parameter tuples, shape, nnz,
  index base) followed by the raw CSC arrays of every term (colptr int64, rowval int64, nzval complex128), 8-byte aligned.
  ``julia/WAEHip.jl`` writes the same container from a Julia ``LinearOperatorFamily``.

Coefficient functions are stored by NAME (or constructor expression, see algebra._tag): only functions of
nlevp/algebra.py (= src/NLEVP/algebra.jl) or ones passed in ``functions=`` can be loaded.
"""
from __future__ import annotations

import datetime
import json
import re
import struct

import numpy as np
import scipy.sparse as sp

from . import algebra
from .linopfam import LinearOperatorFamily, Solution, Term

MAGIC = b"WAEFAM1\n"


# ------------------------------------------------------------------------------------------------------
# Julia literals
# ------------------------------------------------------------------------------------------------------
def _jl_float(x):
    x = float(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    return repr(x)


def _jl_complex(v, spaced=False):
    """``$(real(v))+$(imag(v))im`` as vector_write does (save.jl:71-82); spaced=True mimics ``$value`` of a Complex."""
    v = complex(v)
    re_, im_ = _jl_float(v.real), _jl_float(abs(v.imag) if not np.isnan(v.imag) else v.imag)
    neg = (v.imag < 0) or (v.imag == 0 and np.signbit(v.imag))
    if im_ in ("NaN", "Inf"):
        im_ += "*"
    if spaced:
        return f"{re_} {'-' if neg else '+'} {im_}im"
    return f"{re_}{'-' if neg else '+'}{im_}im"


def _jl_vector(V):
    return "[" + "".join(_jl_complex(v) + "," for v in np.asarray(V).ravel()) + "]"


def _jl_value(x):
    if isinstance(x, (list, tuple, np.ndarray)):
        return "[" + ", ".join(_jl_value(v) for v in x) + "]"
    if isinstance(x, (complex, np.complexfloating)):
        return _jl_complex(x)
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    return _jl_float(x)


def function_expr(f):
    """Julia expression that names / reconstructs a coefficient function."""
    spec = algebra.spec_of(f)
    if spec is None:
        raise ValueError(f"coefficient function {f!r} has no name: build it with the generators of nlevp.algebra "
                         "or tag it with algebra._tag(f, 'name', args...)")
    name, args = spec[0], spec[1:]
    if not args:
        return name
    return name + "(" + ", ".join(function_expr(a) if callable(a) else _jl_value(a) for a in args) + ")"


class Ident(str):
    """a bare identifier in a file (a function name)"""


class Call(tuple):
    """(name, args) of a constructor call in a file"""


_REAL = re.compile(r"[+-]?\s*[+-]?\s*(?:0x[0-9a-fA-F]+|Inf|NaN|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")
_IDENT = re.compile(r"[^\W\d][\w!]*", re.UNICODE)


class _Parser:
    def __init__(self, text):
        self.s = text
        self.i = 0

    def ws(self):
        while self.i < len(self.s) and self.s[self.i] in " \t\r\n":
            self.i += 1

    def peek(self):
        self.ws()
        return self.s[self.i] if self.i < len(self.s) else ""

    def expect(self, ch):
        if self.peek() != ch:
            raise ValueError(f"expected {ch!r} at {self.i}: ...{self.s[max(0, self.i - 20):self.i + 20]!r}")
        self.i += 1

    def sequence(self, close):
        out = []
        while True:
            if self.peek() == close:
                self.i += 1
                return out
            out.append(self.value())
            if self.peek() in ",;":
                self.i += 1

    def number(self):
        total_re, total_im, is_c, is_f = 0.0, 0.0, False, False
        first = True
        while True:
            self.ws()
            m = _REAL.match(self.s, self.i)
            if not m or (not first and m.group(0).lstrip()[0] not in "+-"):
                break
            tok = m.group(0).replace(" ", "")
            j = m.end()
            body = tok.lstrip("+-")
            # the reference's writers print ``$(real)+$(imag)im`` whenever imag(v) >= 0, which is true for -0.0 as well
            # (save.jl:73-78, LinOpFam.jl:283-287): "1.0+-0.0im" -- a binary sign followed by the number's own sign
            sign = -1.0 if tok[:len(tok) - len(body)].count("-") % 2 else 1.0
            if body.startswith("0x"):
                val = float(int(body, 16))
            elif body == "Inf":
                val, is_f = float("inf"), True
            elif body == "NaN":
                val, is_f = float("nan"), True
            else:
                val = float(body)
                is_f = is_f or any(c in body for c in ".eE")
            k = j
            if self.s.startswith("*", k):
                k += 1
            if self.s.startswith("im", k) and not (k + 2 < len(self.s) and (self.s[k + 2].isalnum() or self.s[k + 2] == "_")):
                total_im = sign * val if not is_c else total_im + sign * val     # (keeps the sign of a lone -0.0)
                is_c = True
                j = k + 2
            else:
                total_re += sign * val
            self.i = j
            first = False
        if first:
            raise ValueError(f"cannot parse a value at {self.i}: {self.s[self.i:self.i + 30]!r}")
        if is_c:
            return complex(total_re, total_im)
        return total_re if is_f else int(total_re)

    def value(self):
        c = self.peek()
        if c == "[":
            self.i += 1
            return self.sequence("]")
        if c == "(":
            self.i += 1
            return tuple(self.sequence(")"))
        if c == '"':
            j = self.s.index('"', self.i + 1)
            out = self.s[self.i + 1:j]
            self.i = j + 1
            return out
        if c == ":":
            self.i += 1
            m = _IDENT.match(self.s, self.i)
            if not m:
                raise ValueError(f"bad symbol at {self.i}")
            self.i = m.end()
            return m.group(0)
        m = _IDENT.match(self.s, self.i)
        if m and m.group(0) not in ("Inf", "NaN", "im"):
            name = m.group(0)
            self.i = m.end()
            if self.s.startswith("{", self.i):                  # type parameters: Complex{Float64}[...]
                depth = 0
                while True:
                    depth += {"{": 1, "}": -1}.get(self.s[self.i], 0)
                    self.i += 1
                    if depth == 0:
                        break
            if self.s.startswith("[", self.i):                  # typed array: UInt32[...], Any[...]
                self.i += 1
                return self.sequence("]")
            if self.s.startswith("(", self.i):                  # constructor call / Symbol("τ/Taylor")
                self.i += 1
                args = self.sequence(")")
                if name == "Symbol" and len(args) == 1:
                    return args[0]
                return Call((name, tuple(args)))
            return Ident(name)
        return self.number()


def parse_julia(text):
    p = _Parser(text)
    v = p.value()
    p.ws()
    if p.i != len(p.s):
        raise ValueError(f"trailing text after value: {p.s[p.i:p.i + 30]!r}")
    return v


def read_toml(fname):
    """The reader src/NLEVP/toml.jl:10-63, with the literal parser above in place of ``eval``.  Tags become nested dicts
    whose keys carry the reference's "/" prefix."""
    D, entry, tagged = {}, None, False
    data, var, multi = "", "", False
    with open(fname, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line or line[0] == "#":
                continue
            if not multi and line[0] == "[":
                entry, tagged = D, True
                for tag in line[1:-1].split("."):
                    entry = entry.setdefault("/" + tag, {})
            elif not multi and line[0].isalpha():
                k = line.index("=")
                var, data = line[:k].strip(), line[k + 1:]
                multi = data.endswith(",")
            elif multi:
                data += line
                multi = data.endswith(",")
            else:
                continue                               # stray line (save.jl:17 writes a lone "]"): the reference skips it too
            if not multi and data != "":
                (entry if tagged else D)[var] = parse_julia(data)
                data = ""
    return D


# ------------------------------------------------------------------------------------------------------
# Solution  (save.jl:2-67, read_sol :88-135)
# ------------------------------------------------------------------------------------------------------
def _stamp():
    return datetime.datetime.now(datetime.timezone.utc).replace(tzinfo=None).isoformat(timespec="milliseconds")


def _write_params(f, params):
    f.write("params=[")
    for key, value in params.items():
        f.write(f"(:{key},{_jl_complex(value, spaced=True)}),\n")
    f.write("]\n")


def save_solution(fname, sol):
    with open(fname, "w", encoding="utf-8") as f:
        f.write("# Solution version 0\n#" + _stamp() + "\n")
        _write_params(f, sol.params)
        f.write(f"eigval=:{sol.eigval}\n")
        f.write("v=" + _jl_vector(sol.v) + "\n]\n")                    # the stray "]" is the reference's (save.jl:17)
        f.write("v_adj=" + _jl_vector(sol.v_adj) + "\n")
        f.write("[eigval_pert]\n")
        for key, value in sol.eigval_pert.items():
            f.write(f"\t[eigval_pert.{key}]\n")
            if isinstance(value, tuple):
                f.write("\t\tnum=" + _jl_vector(value[0]) + "\n\t\tden=" + _jl_vector(value[1]) + "\n")
            else:
                f.write("\t\tnum=" + _jl_vector(value) + "\n")
        f.write("[v_pert]\n")
        for key, value in sol.v_pert.items():
            f.write(f"\t[v_pert.{key}]\n")
            parts = (("num", value[0]), ("den", value[1])) if isinstance(value, tuple) else (("num", value),)
            for nm, vecs in parts:
                f.write(f"\t\t[v_pert.{key}.{nm}]\n")
                for idx, val in enumerate(vecs, start=1):
                    f.write(f"\t\t\t[v_pert.{key}.{nm}.{idx}]\n\t\t\tv=" + _jl_vector(val) + "\n")


def read_sol(fname):
    D = read_toml(fname)
    params = {sym: complex(val) for sym, val in D["params"]}
    sol = Solution(params, np.asarray(D["v"], dtype=complex), np.asarray(D["v_adj"], dtype=complex), D["eigval"])
    for key, value in D.get("/eigval_pert", {}).items():
        num = np.asarray(value["num"], dtype=complex)
        sol.eigval_pert[key[1:]] = (num, np.asarray(value["den"], dtype=complex)) if "den" in value else num
    for key, value in D.get("/v_pert", {}).items():
        def vecs(block):
            return [np.asarray(block[f"/{i}"]["v"], dtype=complex) for i in range(1, len(block) + 1)]
        num = vecs(value["/num"])
        sol.v_pert[key[1:]] = (num, vecs(value["/den"])) if "/den" in value else num
    return sol


# ------------------------------------------------------------------------------------------------------
# LinearOperatorFamily, text  (LinOpFam.jl:196-294)
# ------------------------------------------------------------------------------------------------------
def _jl_params_tuple(params):
    def one(t):
        return "(" + ", ".join(":" + p for p in t) + ("," if len(t) == 1 else "") + ")"
    return "(" + ", ".join(one(t) for t in params) + ("," if len(params) == 1 else "") + ")"


def save_family(fname, L):
    eq = "".join("+" + ((t.symbol + "*") if t.symbol else "") + t.operator for t in L.terms if not t.operator.startswith("_"))
    with open(fname, "w", encoding="utf-8") as f:
        f.write("# LinearOperatorFamily version 0\n#" + _stamp() + "\n#" + eq + "\n")
        _write_params(f, L.params)
        f.write(f"eigval=:{L.eigval}\nauxval=:{L.auxval}\n[terms]\n")
        for idx, term in enumerate(L.terms, start=1):
            A = sp.coo_matrix(sp.csc_matrix(term.coeff))                # column-major order, like findnz
            f.write(f"\t[terms.{idx}]\n\tfunctions=(" + "".join(function_expr(fn) + "," for fn in term.func) + ")\n")
            f.write(f"\tsymbol=\"{term.symbol}\"\n\tparams={_jl_params_tuple(term.params)}\n\toperator=\"{term.operator}\"\n")
            f.write(f"\tsize=[{A.shape[0]},{A.shape[1]}]\n\t\t[terms.{idx}.sparse_matrix]\n")
            f.write("\t\tI=[" + ", ".join(str(int(i) + 1) for i in A.row) + "]\n")
            f.write("\t\tJ=[" + ", ".join(str(int(j) + 1) for j in A.col) + "]\n")
            f.write("\t\tV=Complex{Float64}" + _jl_vector(A.data) + "\n\n")


def resolve_function(x, functions=None):
    """Ident / Call from a file -> coefficient function.  Only names of nlevp.algebra or of ``functions`` are known."""
    table = {"generate_Σy_exp_ikx": algebra.generate_Sigma_y_exp_ikx}
    for nm in ("pow0", "pow1", "pow2", "exp_delay", "tau_delay", "pow_a", "generate_exp_az", "exp_pm", "generate_z_g_z",
               "generate_gz_hz", "generate_1_gz"):
        table[nm] = getattr(algebra, nm)
    table.update(functions or {})
    if isinstance(x, Call):
        name, args = x
        if name not in table:
            raise KeyError(f"unknown coefficient-function constructor {name!r}")
        return table[name](*[resolve_function(a, functions) if isinstance(a, (Ident, Call)) else a for a in args])
    if isinstance(x, Ident):
        if x not in table:
            raise KeyError(f"unknown coefficient function {str(x)!r}: pass functions={{{str(x)!r}: f}}")
        return table[x]
    raise TypeError(f"not a function reference: {x!r}")


def load_family(fname, functions=None, device=0):
    with open(fname, "rb") as f:
        if f.read(len(MAGIC)) == MAGIC:
            return load_family_bin(fname, functions, device)
    D = read_toml(fname)
    names = [p for p, _ in D["params"]]
    L = LinearOperatorFamily(names, [complex(v) for _, v in D["params"]], device=device)
    L.eigval, L.auxval, L.active = D["eigval"], D["auxval"], [D["eigval"]]
    terms = D["/terms"]
    for idx in range(1, len(terms) + 1):
        t = terms[f"/{idx}"]
        m, n = t["size"]
        sm = t["/sparse_matrix"]
        A = sp.csr_matrix((np.asarray(sm["V"], dtype=complex), (np.asarray(sm["I"], dtype=np.int64) - 1,
                                                                np.asarray(sm["J"], dtype=np.int64) - 1)), shape=(m, n))
        L.push(Term(A, tuple(resolve_function(fn, functions) for fn in t["functions"]),
                    tuple(tuple(p) for p in t["params"]), t["symbol"], t["operator"]))
    for p, v in D["params"]:                           # push! registers unknown parameters as NaN: restore the values
        L.params[p] = complex(v)
    return L


# ------------------------------------------------------------------------------------------------------
# LinearOperatorFamily, binary
# ------------------------------------------------------------------------------------------------------
def _pad8(f):
    f.write(b"\0" * (-f.tell() % 8))


def save_family_bin(fname, L):
    head = {"version": 1, "eigval": L.eigval, "auxval": L.auxval, "active": list(L.active), "mode": L.mode,
            "params": {k: [complex(v).real, complex(v).imag] for k, v in L.params.items()}, "terms": []}
    mats = []
    for t in L.terms:
        A = sp.csc_matrix(t.coeff)
        A.sort_indices()
        mats.append(A)
        head["terms"].append({"symbol": t.symbol, "operator": t.operator, "functions": [function_expr(fn) for fn in t.func],
                              "params": [list(p) for p in t.params], "m": A.shape[0], "n": A.shape[1], "nnz": int(A.nnz),
                              "base": 0})
    # non-finite parameter values (λ = Inf) are not JSON: encode as strings
    blob = json.dumps(head, ensure_ascii=False, allow_nan=True).encode("utf-8")
    with open(fname, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(blob)))
        f.write(blob)
        for A in mats:
            for arr, dt in ((A.indptr, np.int64), (A.indices, np.int64), (A.data, np.complex128)):
                _pad8(f)
                f.write(np.ascontiguousarray(arr, dtype=dt).tobytes())


def load_family_bin(fname, functions=None, device=0):
    with open(fname, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError("not a WAEFAM1 file")
        (n,) = struct.unpack("<Q", f.read(8))
        head = json.loads(f.read(n).decode("utf-8"))
        names = list(head["params"])
        L = LinearOperatorFamily(names, [complex(*head["params"][k]) for k in names], device=device)
        L.eigval, L.auxval = head["eigval"], head["auxval"]
        for t in head["terms"]:
            out = []
            for cnt, dt in ((t["n"] + 1, np.int64), (t["nnz"], np.int64), (t["nnz"], np.complex128)):
                f.seek(-f.tell() % 8, 1)
                out.append(np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dtype=dt))
            ptr, idx, val = out
            A = sp.csc_matrix((val, idx - t["base"], ptr - t["base"]), shape=(t["m"], t["n"])).tocsr()
            L.push(Term(A, tuple(resolve_function(parse_julia(fn), functions) for fn in t["functions"]),
                        tuple(tuple(p) for p in t["params"]), t["symbol"], t["operator"]))
        for k in names:
            L.params[k] = complex(*head["params"][k])
        L.active, L.mode = list(head.get("active", [L.eigval])), head.get("mode", "all")
    return L


def save(fname, obj, binary=False):
    """``save(fname, L)`` / ``save(fname, sol)`` of the reference (LinOpFam.jl:236, save.jl:2)."""
    if isinstance(obj, Solution):
        return save_solution(fname, obj)
    if isinstance(obj, LinearOperatorFamily):
        return save_family_bin(fname, obj) if binary else save_family(fname, obj)
    raise TypeError(f"cannot save {type(obj).__name__}")
