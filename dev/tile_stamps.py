#!/usr/bin/env python3
"""phase timeline of one workgroup of spmv_tile_kernel (diagnostic build dev/ab/libwaehip_stamps.so, -DWAE_TILE_STAMPS)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import wae_amd
from wae_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ.get("WAE_AB_LIB", "libwaehip_stamps.so"))
from wae_amd.helmholtz.family import annulus_family
preset = sys.argv[1] if len(sys.argv) > 1 else "C3"
L, pb = annulus_family(preset, tau=2e-4)
fam = L.device()
cz = L.coefficients(2 * np.pi * (500 + 20j))
ms = fam.bench_spmv(cz, r=64, reps=20)
print("us per launch", ms * 1e3)
out = (C.c_ulonglong * 512)()
_lib.lib().wae_debug_tile_stamps(out)
t = np.array(list(out), dtype=np.int64).reshape(64, 8)[:8, :7]
print("cycles (100 MHz ticks?) per phase, chunks 0..7: [spc+wait window | issue next DMA | compute | barrier(+DMA drain) | stage | epilogue]")
print(np.diff(t, axis=1))
print("chunk start to next chunk start:", np.diff(t[:, 0]))
