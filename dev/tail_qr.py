#!/usr/bin/env python3
"""Development check: cost of the tall-skinny factorisation in the Beyn tail (d x 16 complex128 on the GPU)."""
import time, torch
d, l = 199680, 16
torch.manual_seed(0)
B0 = torch.randn(d, l, dtype=torch.complex128, device="cuda")
B1 = torch.randn(d, l, dtype=torch.complex128, device="cuda")
def T(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print("qr reduced      %.2f ms" % T(lambda: torch.linalg.qr(B0)))
print("qr mode=r       %.2f ms" % T(lambda: torch.linalg.qr(B0, mode="r")))
print("gram B0^H B0    %.2f ms" % T(lambda: B0.conj().T @ B0))
print("B0 @ (16x16)    %.2f ms" % T(lambda: B0 @ torch.eye(l, dtype=torch.complex128, device="cuda")))
print("U^H B1 (16xd d x16) %.2f ms" % T(lambda: B0.conj().T @ B1))
print("svd 16x16       %.2f ms" % T(lambda: torch.linalg.svd(B0[:16, :])))
