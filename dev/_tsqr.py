import time, torch
torch.manual_seed(0)
dev = "cuda:0"
rows, cols = 2 * 995328, 16
A = torch.randn(rows, 8, dtype=torch.complex128, device=dev)
A = torch.cat([A, A @ torch.randn(8, 8, dtype=torch.complex128, device=dev) * 1e-11 + 1e-11 * torch.randn(rows, 8, dtype=torch.complex128, device=dev)], dim=1)
def plain(A):
    Q, R = torch.linalg.qr(A); return Q, R
def tsqr(A, blk=2048):
    n = A.shape[0]; nb = n // blk; nfull = nb * blk
    Q1, R1 = torch.linalg.qr(A[:nfull].reshape(nb, blk, A.shape[1]))
    Rs = R1.reshape(nb * A.shape[1], A.shape[1])
    if nfull < n:
        Qt, Rt = torch.linalg.qr(A[nfull:]); Rs = torch.cat([Rs, Rt])
    Q2, R = torch.linalg.qr(Rs)
    Q = torch.bmm(Q1, Q2[:nb * A.shape[1]].reshape(nb, A.shape[1], A.shape[1])).reshape(nfull, A.shape[1])
    if nfull < n: Q = torch.cat([Q, Qt @ Q2[nb * A.shape[1]:]])
    return Q, R
for name, f in (("plain", plain), ("tsqr2048", lambda A: tsqr(A, 2048)), ("tsqr512", lambda A: tsqr(A, 512)), ("tsqr8192", lambda A: tsqr(A, 8192))):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); Q, R = f(A); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    err = (Q @ R - A).abs().max().item() / A.abs().max().item()
    orth = (Q.conj().T @ Q - torch.eye(cols, dtype=Q.dtype, device=dev)).abs().max().item()
    print(name, "%.1f ms" % (dt * 1e3), "recon %.1e orth %.1e" % (err, orth), flush=True)
