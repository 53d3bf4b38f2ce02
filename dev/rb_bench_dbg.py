#!/usr/bin/env python3
"""Development check: eigenpair residuals after the snapshot-projection Beyn pass (bench flow)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import wae_amd  # noqa
import bench
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp.distributed import beyn_moments_distributed_rb, moments2eigs_device
from wae_amd.nlevp.beyn import inpoly
from wae_amd.nlevp import compute_moment_matrices

L, pb = annulus_family("C2", tau=2e-4)
d = pb["d"]
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
G = np.array(bench.GAMMA_HZ) * 2 * np.pi
V = np.random.default_rng(7).standard_normal((d, 16)) + 0j
for rb in (0, 32):
    if rb:
        buf, info = beyn_moments_distributed_rb(L, G, V, 1, 32, rb)
    else:
        buf = torch.zeros(d * 16 * 2 * 2, dtype=torch.float64, device="cuda:0")
        compute_moment_matrices(L, G, V, K=1, N=32, out_dev=buf.data_ptr(), rb=0)
    Om, Pd, S = moments2eigs_device(buf, (d, 16, 2))
    mask = np.array([inpoly(w, G) for w in Om], dtype=bool)
    print("rb", rb, "Om/2pi", np.round(Om / 2 / np.pi, 3))
    Om = Om[mask]
    P = np.asfortranarray(Pd[:, torch.from_numpy(mask).to(Pd.device)].cpu().numpy())
    r = fam.eig_residuals(np.array([L.coefficients(w) for w in Om]), P=P)
    print("rb", rb, "inside", np.round(Om / 2 / np.pi, 3), "res", r, flush=True)
    # host cross-check of the first eigenpair's residual, and of the moments themselves
    T = pb["terms"]
    w = Om[0]; v = P[:, 0]
    cs = [w * w, 1.0, w * 1e15, np.exp(-1j * w * 2e-4)]
    parts = [c_ * (T[k] @ v) for c_, k in zip(cs, "MKCQ")]
    num = np.linalg.norm(sum(parts)); den = sum(np.linalg.norm(p_) for p_ in parts)
    print("   host residual of pair 0:", num / den, " rows: penalty part", np.linalg.norm(sum(parts)[T["C"].diagonal() != 0]) / den, flush=True)
    A = buf.cpu().numpy().view(np.complex128).reshape((d, 16, 2), order="F")
    if rb == 0:
        Aref = A.copy()
    else:
        E = A - Aref
        print("   moment error: max %.2e (rel %.2e); per column max" % (np.abs(E).max(), np.abs(E).max() / np.abs(Aref).max()), np.round(np.log10(np.abs(E[:, :, 0]).max(axis=0) / np.abs(Aref[:, :, 0]).max(axis=0)), 1))
        pen = T["C"].diagonal() != 0
        print("   moment error on penalty rows rel to their own size: %.2e ; interior rows: %.2e" % (np.abs(E[pen]).max() / np.abs(Aref[pen]).max(), np.abs(E[~pen]).max() / np.abs(Aref[~pen]).max()))
