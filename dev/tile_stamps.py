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
out = (C.c_ulonglong * 520)()
_lib.lib().wae_debug_tile_stamps(out)
full = np.array(list(out), dtype=np.int64).reshape(65, 8)
raw = full[:8]
print("ticks: kernel entry -> prologue done %d; whole workgroup %d (%.1f us)" % (full[8, 1] - full[8, 0], full[8, 2] - full[8, 0], (full[8, 2] - full[8, 0]) / 2.377e3))
t = raw[:, :7]
dt_real = (raw[7, 7] - raw[0, 7]) / 100e6
print("s_memtime ticks per second: %.4g  (chunks 0..7 span %.2f us)" % ((raw[7, 0] - raw[0, 0]) / dt_real, dt_real * 1e6))
print("ticks per phase, chunks 0..7: [barrier (window landed) | - | compute + next-window pieces | lane-pair combine | loads + results | DMA wait + stores]")
print(np.diff(t, axis=1))
print("chunk start to next chunk start:", np.diff(t[:, 0]))

wl = (C.c_ulonglong * 4096)()
_lib.lib().wae_debug_tile_wglog(wl)
w = np.array(list(wl), dtype=np.int64).reshape(1024, 4)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
start = (w[:, 0] - t0) / 100.0
end = (w[:, 1] - t0) / 100.0
xcc = (w[:, 3] >> 32) & 0xf
hw = w[:, 3] & 0xffffffff
print("workgroups %d: start %.1f..%.1f us, end %.1f..%.1f us (median %.1f), chunks per workgroup %d..%d" % (len(w), start.min(), start.max(), end.min(), end.max(), np.median(end), w[:, 2].min(), w[:, 2].max()))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCC %d: %3d workgroups, blockIdx&7 in %s, end %.1f..%.1f us, us per chunk %.2f..%.2f, distinct (SE,CU) ids %d" % (x, m.sum(), sorted(set((np.nonzero(m)[0] & 7).tolist())), end[m].min(), end[m].max(), ((end[m] - start[m]) / w[m, 2]).min(), ((end[m] - start[m]) / w[m, 2]).max(), len(set((hw[m] >> 8 & 0xff).tolist()))))
