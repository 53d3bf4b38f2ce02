#!/usr/bin/env python3
"""A/B of the fine-level SpMV kernels: python dev/spmv_ab.py PRESET [r ...]  (env WAE_SPMV_TILE / WAE_REORDER select the variant)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd  # noqa
from wae_amd import _lib
if os.environ.get("WAE_AB_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ["WAE_AB_LIB"])
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C2"
rs = [int(x) for x in sys.argv[2:]] or [64]
L, pb = annulus_family(preset, tau=2e-4)
fam = L.device()
cz = L.coefficients(2 * np.pi * (500 + 20j))
mask = [1 if c != 0 else 0 for c in cz]
for r in rs:
    ms = fam.bench_spmv(cz, r=r, reps=50)
    by = fam.spmv_bytes(r=r, mask=mask)
    print(f"{preset} r={r}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s algorithmic  (TILE={os.environ.get('WAE_SPMV_TILE','1')} REORDER={os.environ.get('WAE_REORDER','1')})", flush=True)
