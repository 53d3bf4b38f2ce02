"""Oracle restatement of the NLEVP operator layer (test infrastructure, see oracle/__init__.py).

Follows the reference files
  src/NLEVP/algebra.jl      (scalar coefficient functions f(z..., k...) = k-th derivative)
  src/NLEVP/LinOpFam.jl     (Term, LinearOperatorFamily, Solution, Pade helpers)
Matrices are scipy CSC (the reference uses Julia SparseMatrixCSC) or dense ndarrays.
"""
from __future__ import annotations

import copy
import math
from math import comb, factorial

import numpy as np
import scipy.sparse as sp

NaN = complex(float("nan"), float("nan"))


# ----------------------------------------------------------------------------------------------
# algebra.jl
# ----------------------------------------------------------------------------------------------
def pow0(z, k=0):
    """algebra.jl:4-12"""
    if k == 0:
        return 1.0 + 0j
    return 0j if k > 0 else NaN


def pow1(z, k=0):
    """algebra.jl:16-26"""
    if k == 0:
        return complex(z)
    if k == 1:
        return 1.0 + 0j
    return 0j if k > 1 else NaN


def pow2(z, k=0):
    """algebra.jl:30-42"""
    z = complex(z)
    if k == 0:
        return z * z
    if k == 1:
        return 2 * z
    if k == 2:
        return 2.0 + 0j
    return 0j if k > 2 else NaN


def pow_(z, k, a):
    """algebra.jl:46-76  k-th derivative of z^a"""
    z = complex(z)
    if isinstance(a, (int, np.integer)) and k > a > 0:
        return 0j
    if k >= 0:
        f = 1
        i = a
        for _ in range(k):
            f *= i
            i -= 1
        if f == 0:
            return 0j
        return f * z ** (a - k)
    return NaN


def pow_a(a):
    """algebra.jl:78-107"""
    def f(z, k=0):
        return pow_(z, k, a)
    f.__name__ = f"pow_a({a})"
    return f


def exp_az(z, a, k):
    """algebra.jl:129-135"""
    return a ** k * np.exp(a * z)


def generate_exp_az(a):
    """algebra.jl:110-127"""
    def f(z, k):
        return a ** k * np.exp(a * z) if k >= 0 else NaN
    return f


def exp_delay(omega, tau, m, n):
    """algebra.jl:138-147:  d^m/dω^m d^n/dτ^n exp(-iωτ)"""
    a = -1.0j
    omega = complex(omega)
    tau = complex(tau)
    f = 0j
    for i in range(n + 1):
        f += comb(n, i) * pow_(tau, i, m) * (a * omega) ** (n - i)
    f *= a ** m * np.exp(a * omega * tau)
    return complex(f)


def exp_pm(s):
    """algebra.jl:215-227  (Bloch phase factors exp(±i b 2π/N) are built from this)"""
    a = s * 1.0j

    def f(omega, tau, m, n):
        r = 0j
        for i in range(n + 1):
            r += comb(n, i) * pow_(tau, i, m) * (a * omega) ** (n - i)
        return complex(r * a ** m * np.exp(a * omega * tau))
    return f


def generate_z_g_z(g):
    """algebra.jl:169-179"""
    def z_g_z(z, n):
        if n == 0:
            return z * g(z, 0)
        return z * g(z, n) + n * g(z, n - 1)
    return z_g_z


def generate_gz_hz(g, h):
    """algebra.jl:290-299"""
    def func(z, k):
        return sum(comb(k, i) * h(z, k - i) * g(z, i) for i in range(k + 1))
    return func


def generate_1_gz(g):
    """algebra.jl:301-310"""
    def func(z, k):
        return 1 - g(z, k) if k == 0 else -g(z, k)
    return func


def generate_Sigma_y_exp_ikx(y):
    """algebra.jl:276-288"""
    N = len(y)

    def f(z, n):
        s = 0j
        for k, yk in enumerate(y):
            s += (k ** n if (k or n == 0) else 0) * yk * np.exp(2j * np.pi * k / N * z)
        return s * (2j * np.pi / N) ** n
    return f


# ----------------------------------------------------------------------------------------------
# LinOpFam.jl
# ----------------------------------------------------------------------------------------------
class Term:
    """LinOpFam.jl:16-35.  coeff * prod_j func_j(params_j..., derivs_j...)"""

    def __init__(self, coeff, func, params, symbol="", operator=""):
        self.coeff = coeff
        self.func = tuple(func)
        self.params = tuple(tuple(p) for p in params)
        self.symbol = symbol
        self.operator = operator
        varlist = []
        for par in self.params:
            for var in par:
                if var not in varlist:
                    varlist.append(var)
        self.varlist = varlist

    def scalar(self, d):
        """LinOpFam.jl:466-477 (the scalar part): d maps symbol -> (value, deriv order)."""
        c = 1.0 + 0j
        for func, pars in zip(self.func, self.params):
            args = [d[p][0] for p in pars]
            dargs = [d[p][1] for p in pars]
            c *= func(*args, *dargs)
        return complex(c)

    def __call__(self, d):
        """LinOpFam.jl:466-479"""
        return self.scalar(d) * self.coeff


class Solution:
    """LinOpFam.jl:95-112"""

    def __init__(self, params, v, v_adj, eigval, auxval=""):
        self.params = copy.deepcopy(params)
        self.v = v
        self.v_adj = v_adj
        self.eigval = eigval
        self.eigval_pert = {}
        self.v_pert = {}
        self.auxval = auxval

    def __call__(self, param, eps, L=0, M=0):
        """LinOpFam.jl:684-699 (eigenvalue only)"""
        key = f"{param}/[{L}/{M}]"
        if key not in self.eigval_pert:
            self.eigval_pert[key] = pade(self.eigval_pert[f"{param}/Taylor"], L, M)
        a, b = self.eigval_pert[key]
        de = eps - self.params[param]
        return polyval(a, de) / polyval(b, de)


class LinearOperatorFamily:
    """LinOpFam.jl:131-186 and functor :482-529"""

    def __init__(self, params=("λ",), values=None):
        params = list(params)
        if values is None:
            values = [NaN for _ in params]
        self.terms = []
        self.eigval = params[0]
        self.auxval = params[-1] if len(params) > 1 else ""
        self.active = [self.eigval]
        self.params = {p: complex(v) for p, v in zip(params, values)}
        self.mode = "all"

    def push(self, T):
        """LinOpFam.jl:305-346"""
        for idx, term in enumerate(self.terms):
            if term.func == T.func and term.params == T.params:
                coeff = term.coeff + T.coeff
                nrm = abs(coeff).sum() if sp.issparse(coeff) else np.abs(coeff).sum()
                if nrm == 0:
                    del self.terms[idx]
                else:
                    self.terms[idx] = Term(coeff, term.func, term.params, term.symbol, term.operator)
                return
        for pars in T.params:
            for par in pars:
                if par not in self.params:
                    self.params[par] = NaN
        self.terms.append(T)

    def size(self):
        """LinOpFam.jl:385-393"""
        return self.terms[0].coeff.shape[0] if self.terms else 0

    def coefficients(self, *args, oplist=(), in_or_ex=False):
        """Scalar part of the functor, LinOpFam.jl:482-526: returns [c_k or None (term skipped)]."""
        nact = len(self.active)
        if self.mode == "all":
            for var, val in zip(self.active, args):
                self.params[var] = complex(val)
        if self.mode == "all" and len(args) == nact:
            derivs = [0] * nact
        else:
            derivs = [int(a) for a in args[len(args) - nact:]]
        deriv_dict = dict(zip(self.active, derivs))
        out = []
        for term in self.terms:
            if ((not in_or_ex and term.operator in oplist) or (in_or_ex and term.operator not in oplist)
                    or (self.mode != "householder" and term.operator == "__aux__")):
                out.append(None)
                continue
            if any(d > 0 and var not in term.varlist for var, d in zip(self.active, derivs)):
                out.append(None)
                continue
            d = {var: (self.params[var], deriv_dict.get(var, 0)) for var in term.varlist}
            out.append(term.scalar(d))
        if self.mode in ("compact", "householder"):
            div = 1.0
            for a in args[len(args) - nact:]:
                div *= float(factorial(int(a)))
            out = [None if c is None else c / div for c in out]
        return out

    def __call__(self, *args, oplist=(), in_or_ex=False):
        """LinOpFam.jl:482-529: materialise L(args) = sum_k c_k A_k."""
        cs = self.coefficients(*args, oplist=oplist, in_or_ex=in_or_ex)
        A0 = self.terms[0].coeff
        if sp.issparse(A0):
            acc = sp.csc_matrix(A0.shape, dtype=complex)
        else:
            acc = np.zeros(A0.shape, dtype=complex)
        for c, term in zip(cs, self.terms):
            if c is None:
                continue
            acc = acc + c * term.coeff
        if sp.issparse(acc):
            acc = sp.csc_matrix(acc)
        return acc


def polyval(p, z):
    """LinOpFam.jl:723-730 (Horner)"""
    f = p[-1]
    for i in range(len(p) - 2, -1, -1):
        f = f * z + p[i]
    return f


def pade(w, L, M):
    """LinOpFam.jl:622-642"""
    w = np.asarray(w, dtype=complex)
    b = np.array([1.0 + 0j])
    if M > 0:
        A = np.zeros((M, M), dtype=complex)
        for i in range(1, M + 1):
            for j in range(1, M + 1):
                if L + i - j >= 0:
                    A[i - 1, j - 1] = w[L + i - j]
        b = np.concatenate([[1.0 + 0j], np.linalg.solve(A, -w[L + 1:L + M + 1])])
    a = np.zeros(L + 1, dtype=complex)
    for l in range(L + 1):
        for m in range(l + 1):
            if m <= M:
                a[l] += w[l - m] * b[m]
    return a, b


def pade_vector(sol, param, L, M):
    """pade!(..., vector=true)  LinOpFam.jl:653-678"""
    tkey = f"{param}/Taylor"
    V = np.array(sol.v_pert[tkey][:L + M + 1])
    d = V.shape[1]
    A = np.zeros((L + 1, d), dtype=complex)
    B = np.zeros((M + 1, d), dtype=complex)
    for i in range(d):
        A[:, i], B[:, i] = pade(V[:, i], L, M)
    return A, B


def estimate_pol(w):
    """LinOpFam.jl:736-747"""
    w = np.asarray(w, dtype=complex)
    N = len(w)
    de = np.zeros(N - 2, dtype=complex)
    k = np.zeros(N - 2, dtype=complex)
    for j in range(2, N):                 # Julia j = 2..N-1 (1-based)
        i = j - 1
        denom = (i + 1) * w[j] * w[j - 2] - i * w[j - 1] ** 2
        de[i - 1] = w[j - 1] * w[j - 2] / denom
        k[i - 1] = (i ** 2 - 1) * w[j] * w[j - 2] - (i * w[j - 1]) ** 2
    return de, k


def conv_radius(a):
    """LinOpFam.jl:754-761"""
    a = np.asarray(a)
    return np.abs(a[:-1] / a[1:])


def poly_roots(p):
    """Householder.jl:195-203 (companion-matrix eigenvalues)"""
    p = np.asarray(p, dtype=complex)
    N = len(p) - 1
    C = np.zeros((N, N), dtype=complex)
    for i in range(1, N):
        C[i, i - 1] = 1
    C[:, N - 1] = -p[:N] / p[N]
    return np.linalg.eigvals(C)
