"""Producers of LinearOperatorFamily objects for the device hot path (the slot Helmholtz.discretize fills)."""
from . import annulus  # noqa: F401
from .family import annulus_family, helmholtz_family  # noqa: F401
