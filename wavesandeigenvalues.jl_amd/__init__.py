"""MI355X-native NLEVP hot path for WavesAndEigenvalues.jl: host-side mirror of the reference's operator /
solver interface over libwaehip.so (hand-written HIP kernels for gfx950, C ABI in include/waehip.h).

The directory name contains a dot, so import it through the top-level shim ``wae_amd`` (repo root):

    import wae_amd
    from wae_amd.nlevp import LinearOperatorFamily, Term, beyn, householder, mslp, perturb_fast_
"""
from . import _lib  # noqa: F401
from . import nlevp  # noqa: F401
from . import helmholtz  # noqa: F401

__all__ = ["_lib", "nlevp", "helmholtz"]
