import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wae_amd
from wae_amd.helmholtz.family import annulus_family
d, l = 199680, 16
rng = np.random.default_rng(0)
B0 = rng.standard_normal((d, l)) + 1j * rng.standard_normal((d, l))
for name, f in (("svd", lambda: np.linalg.svd(B0, full_matrices=False)), ("qr", lambda: np.linalg.qr(B0)),
                ("gram", lambda: B0.conj().T @ B0), ("gemm", lambda: B0 @ np.ones((l, l), dtype=complex))):
    t = time.time(); f(); print(name, time.time() - t, flush=True)
import torch
Bt = torch.from_numpy(B0).cuda()
torch.cuda.synchronize()
for name, f in (("torch svd", lambda: torch.linalg.svd(Bt, full_matrices=False)), ("torch qr", lambda: torch.linalg.qr(Bt))):
    try:
        f(); torch.cuda.synchronize(); t = time.time(); f(); torch.cuda.synchronize(); print(name, time.time() - t, flush=True)
    except Exception as e:
        print(name, "failed", repr(e)[:200])
L, pb = annulus_family("C2", tau=2e-4)
fam = L.device()
cz = L.coefficients(3000 + 10j)
P = B0[:, :8].copy()
for r in (1, 8, 32):
    X = np.asfortranarray(np.tile(P[:, :1], (1, r)))
    fam.spmv(cz, X)
    t = time.time(); fam.spmv(cz, X); print("spmv host call r", r, time.time() - t, flush=True)
    C = np.tile(cz, (r, 1))
    t = time.time(); fam.spmv(C, X); print("spmv cols host call r", r, time.time() - t, flush=True)
