"""Scalar coefficient functions f(z..., k...) = k-th derivative, host side (they stay on the host in the
reference too: src/NLEVP/algebra.jl).  The device only ever receives the evaluated scalars c_k."""
from __future__ import annotations

import cmath
from math import comb

NaN = complex(float("nan"), float("nan"))


def _tag(f, *spec):
    """Attach the constructor expression (name, arguments...) a coefficient function was made by: what
    nlevp/save.py writes into operator files instead of an anonymous closure (the reference writes `$func`,
    LinOpFam.jl:262-265, which for a closure is an unloadable name)."""
    f.__wae_spec__ = spec
    return f


def spec_of(f):
    return getattr(f, "__wae_spec__", None)


def pow0(z, k=0):
    """algebra.jl:4-12"""
    return (1.0 + 0j) if k == 0 else (0j if k > 0 else NaN)


_tag(pow0, "pow0")


def pow1(z, k=0):
    """algebra.jl:16-26"""
    if k == 0:
        return complex(z)
    return (1.0 + 0j) if k == 1 else (0j if k > 1 else NaN)


_tag(pow1, "pow1")


def pow2(z, k=0):
    """algebra.jl:30-42"""
    z = complex(z)
    if k < 0:
        return NaN
    return (z * z, 2 * z, 2.0 + 0j)[k] if k <= 2 else 0j


_tag(pow2, "pow2")


def pow_(z, k, a):
    """algebra.jl:46-76: d^k/dz^k z^a (integer or general exponent a)"""
    if k < 0:
        return NaN
    if isinstance(a, int) and k > a > 0:
        return 0j
    f, i = 1, a
    for _ in range(k):
        f *= i
        i -= 1
    return 0j if f == 0 else f * complex(z) ** (a - k)


def pow_a(a):
    """algebra.jl:78-107"""
    def f(z, k=0):
        return pow_(z, k, a)
    return _tag(f, "pow_a", a)


def exp_az(z, a, k):
    """algebra.jl:129-135"""
    return a ** k * cmath.exp(a * z)


def generate_exp_az(a):
    """algebra.jl:110-127"""
    def f(z, k=0):
        return a ** k * cmath.exp(a * z) if k >= 0 else NaN
    return _tag(f, "generate_exp_az", a)


def _exp_delay_a(a):
    def f(omega, tau, m, n):
        omega, tau = complex(omega), complex(tau)
        s = 0j
        for i in range(n + 1):
            s += comb(n, i) * pow_(tau, i, m) * (a * omega) ** (n - i)
        return s * a ** m * cmath.exp(a * omega * tau)
    return f


exp_delay = _tag(_exp_delay_a(-1.0j), "exp_delay")      # algebra.jl:138-147: d^m/dω^m d^n/dτ^n exp(-iωτ)
tau_delay = exp_delay                # algebra.jl:181


def exp_pm(s):
    """algebra.jl:215-227 (Bloch phase factors)"""
    return _tag(_exp_delay_a(s * 1.0j), "exp_pm", s)


def generate_z_g_z(g):
    """algebra.jl:169-179"""
    def z_g_z(z, n):
        return z * g(z, 0) if n == 0 else z * g(z, n) + n * g(z, n - 1)
    return _tag(z_g_z, "generate_z_g_z", g)


def generate_gz_hz(g, h):
    """algebra.jl:290-299"""
    def func(z, k):
        return sum(comb(k, i) * h(z, k - i) * g(z, i) for i in range(k + 1))
    return _tag(func, "generate_gz_hz", g, h)


def generate_1_gz(g):
    """algebra.jl:301-310"""
    def func(z, k):
        return 1 - g(z, k) if k == 0 else -g(z, k)
    return _tag(func, "generate_1_gz", g)


def generate_Sigma_y_exp_ikx(y):
    """algebra.jl:276-288"""
    N = len(y)

    def f(z, n=0):
        s = 0j
        for k, yk in enumerate(y):
            s += (1 if n == 0 else k ** n) * yk * cmath.exp(2j * cmath.pi * k / N * z)
        return s * (2j * cmath.pi / N) ** n
    return _tag(f, "generate_Σy_exp_ikx", [complex(v) for v in y])
