"""Multi-GPU Beyn: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The quadrature points of the contour are independent (beyn.jl:126-134 is a plain sum), so they are dealt
round-robin to the ranks; every rank holds a full replica of the family (0.07 GB at 200k DoF, 0.4 GB at 1M DoF --
nothing against 288 GB of HBM) and accumulates a partial moment tensor d x l x 2K in its own HBM; the only
exchange step of the whole path is ONE sum all-reduce of that tensor.  The Hankel SVD / eigen tail runs on the host.
"""
from __future__ import annotations

import numpy as np

from .beyn import compute_moment_matrices, gauss_points, initialize_V, moments2eigs, pos_test


def shard_points(zs, ws, rank, world):
    """round-robin shard of the quadrature points (balances the per-edge conditioning differences)"""
    return zs[rank::world], ws[rank::world]


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def rank_world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def allreduce_sum_(t):
    """in-place sum over ranks of a torch tensor (no-op for a single process)"""
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(t)
    return t


class RankFailure(RuntimeError):
    """raised on the ranks that were fine when another rank's share of the work failed"""


def _guard(fn, *a, **kw):
    """run a per-rank piece of work; returns (result, exception or None) instead of raising, so that the rank still reaches
    `fail_together` -- a rank that raised and left would let the others block in the next collective until the process-group
    timeout"""
    try:
        return fn(*a, **kw), None
    except Exception as e:          # noqa: BLE001  (whatever it was, the other ranks must hear of it)
        return None, e


def fail_together(err, what):
    """Every rank calls this between its share of the work and the next collective: one MAX all-reduce of a status word; if
    any rank failed (e.g. an inner solve stalled at a quadrature point near a pole: DeviceFamily raises for contour
    integrals), ALL ranks raise now -- the failing rank its own exception, the others a RankFailure."""
    d = _dist()
    bad = 1 if err is not None else 0
    if d is not None and d.get_world_size() > 1:
        import torch
        t = torch.tensor([bad], dtype=torch.int32)
        if d.get_backend() == "nccl":
            t = t.cuda()
        d.all_reduce(t, op=d.ReduceOp.MAX)
        bad = int(t.item())
    if err is not None:
        raise err
    if bad:
        raise RankFailure(f"{what}: another rank failed; no rank continues into the collective")


def device_moment_fn(L, G, V, K, N):
    """default producer of a rank's partial moments: the HIP path writing straight into a torch CUDA buffer."""
    import torch

    def fn(zs, ws):
        d, l = V.shape
        dev = torch.device("cuda", L.device_id)
        buf = torch.zeros(d * l * 2 * K * 2, dtype=torch.float64, device=dev)
        compute_moment_matrices(L, G, V, K=K, N=N, points=(zs, ws), out_dev=buf.data_ptr())
        return buf
    return fn


def beyn_moments_distributed(G, N, shape, moment_fn):
    """Sharded moments: ``moment_fn(z_shard, w_shard)`` -> flat float64 torch tensor (interleaved re/im of the
    column-major d x l x 2K tensor); returns the all-reduced tensor as a numpy complex array on every rank."""
    rank, world = rank_world()
    zs, ws = gauss_points(G, N)
    zr, wr = shard_points(zs, ws, rank, world)
    buf, err = _guard(moment_fn, zr, wr)
    fail_together(err, "beyn_moments_distributed")
    allreduce_sum_(buf)
    return buf.cpu().numpy().view(np.complex128).reshape(shape, order="F")


def _allgather_dev(local, world):
    """all-gather equal-sized device tensors into one flat device tensor (rank-major); through the host under gloo"""
    import torch
    dd = _dist()
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if dd.get_backend() == "nccl":
        dd.all_gather_into_tensor(out, local)                # RCCL over xGMI, device to device
    else:                                                    # gloo rehearsal on one box
        parts = [torch.empty(local.numel(), dtype=local.dtype) for _ in range(world)]
        dd.all_gather(parts, local.cpu())
        out.copy_(torch.cat(parts))
    return out


def _alltoall_dev(send, world):
    """all-to-all of equal pieces of a flat device tensor (piece r goes to rank r; the result holds the pieces of ranks 0, 1, ...);
    through an all-gather on the host under gloo (which has no all-to-all)"""
    import torch
    dd = _dist()
    if dd.get_backend() == "nccl":
        out = torch.empty_like(send)
        dd.all_to_all_single(out, send)                      # RCCL over xGMI, device to device
        return out
    n = send.numel() // world
    parts = [torch.empty(send.numel(), dtype=send.dtype) for _ in range(world)]
    dd.all_gather(parts, send.cpu())
    rank = dd.get_rank()
    return torch.cat([p[rank * n:(rank + 1) * n] for p in parts]).to(send.device)


def beyn_moments_distributed_rb(L, G, V, K, N, S, timings=None, extra=None, zmap=None):
    """Sharded moments with snapshot-projection initial guesses (wae_beyn_moments_rb).  Two exchange steps.

    (1) Snapshot phase, split by PROBE COLUMN: every rank solves all S snapshot points for its l/world columns.  The
    per-column bases are independent, so each rank's progressive snapshot phase is exactly the single-GPU one for its
    columns and leaves a finished (orthonormal, projected) basis; the bases are exchanged -- all-gather of the basis
    vectors (S x d x l complex in all) and of the small projected terms -- and installed with wae_rb_import: no rank
    repeats another rank's orthogonalisation.  (2) The remaining points are dealt round-robin and start from the
    projection; the partial moments (snapshot contributions in each rank's own columns included) are summed with one
    all-reduce.  When l is not divisible by the world size the snapshot POINTS are split instead and every rank
    rebuilds the basis from the gathered raw snapshots (mode 1).
    zmap = (c, rho) (optional): the moments are taken in the variable z' = (z - c)/rho, A_p = sum_j w_j z'_j^p L(z_j)^{-1} V -- the
    systems solved are the same, only the powers that weigh them change.  Beyn's method is invariant under this affine map (the
    eigenvalues of the Hankel pencil are (omega - c)/rho; the caller maps them back); with raw powers of z ~ 2 pi 500 the K = 2
    Hankel matrix has singular values of 1e12 and the eigenvectors lose a digit.
    Returns the flat float64 CUDA tensor of the column-major d x l x 2K moments (on every rank) and solver statistics."""
    import time

    import torch
    rank, world = rank_world()
    V = np.asfortranarray(np.asarray(V, dtype=np.complex128))   # once: the library takes column-major (a C-ordered d x l
    d, l = V.shape                                              # matrix costs a 10-ms strided copy per call at d = 2e5)
    dev = torch.device("cuda", L.device_id)
    zs, ws = gauss_points(G, N)
    from .beyn import coefficient_table, snapshot_split, spread_order
    fam = L.ensure_solver()
    ct = coefficient_table(L, zs)                    # (the systems: true z)
    if zmap is not None:                             # (the powers: mapped z)
        zs = (np.asarray(zs, dtype=np.complex128) - complex(zmap[0])) / complex(zmap[1])
    buf = torch.zeros(d * l * 2 * K * 2, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)        # the library works on its own stream: torch's fill must have landed before it accumulates
    kw = dict(K=K, tol=L.solver_tol, maxit=L.solver_maxit, out_dev=buf.data_ptr())
    by_column = world > 1 and l % world == 0
    # how the snapshot phase is shared out when the probe columns divide over the ranks (WAE_SNAPSHOT_SPLIT):
    #   "columns" (default): every rank solves all S points for its l/world columns progressively (mode 0);
    #   "hybrid": POINTS for the solves -- a rank's S/world snapshot points x all l columns are full-width batches from zero
    #       guesses (mode 3) --, one all-to-all of the raw solutions, COLUMNS for the basis -- every rank orthonormalises and
    #       projects all S snapshots of its l/world columns (mode 4) -- then the exchange of the finished bases as below.
    # Measured rank shares at 1M unknowns (dev/c3_rank_share.py, profiles/r03_rank_share.json; S = 40, l = 8): columns 0.90 / 0.54 /
    # 0.35 s at 2 / 4 / 8 ranks, hybrid 1.38 / 0.63 / 0.36 s: the progressive guesses save more iterations (759 against 1 135
    # column-iterations at 8 ranks) than the narrow batches cost, and the slice basis is another 0.08-0.13 s.  Hence the default.
    import os
    split = os.environ.get("WAE_SNAPSHOT_SPLIT", "columns")
    hybrid = by_column and split == "hybrid" and int(S) >= world
    t0 = time.perf_counter()
    if by_column or world == 1:
        S = min(int(S), len(zs))
        if hybrid:
            S = (S // world) * world                              # equal shares (the all-to-all wants equal sizes)
        idx, rest = snapshot_split(len(zs), S)
        idx = spread_order(idx)
        S = len(idx)
        ls = l // world
        c0 = rank * ls
        cap = S + (0 if (extra is None or world > 1) else int(extra))
        if hybrid:
            per = S // world
            mine = idx[rank::world]
            raw = torch.empty(per * d * l * 2, dtype=torch.float64, device=dev)
            _, err = _guard(fam.beyn_moments_rb, zs[mine], ws[mine], ct[mine], V, 3, per, Q_dev=raw.data_ptr(), **kw)
            fail_together(err, "snapshot phase (solves)")
            i0 = dict(fam.last_info)
            torch.cuda.synchronize(dev)
            send = raw.view(per, d, world, ls, 2).permute(2, 0, 1, 3, 4).contiguous()      # [destination rank][s][row][c_local]
            del raw
            local = _alltoall_dev(send.view(-1), world)                                    # [source rank][s][row][c_local] = S x d x ls
            del send
            order = np.concatenate([idx[r::world] for r in range(world)])                   # snapshot point behind every slot
            _, err = _guard(fam.beyn_moments_rb, zs[order], ws[order], ct[order], V[:, c0:c0 + ls], 4, S, slot0=S, Q_dev=local.data_ptr(),
                            accumulate=True, l_total=l, col0=c0, **kw)
            fail_together(err, "snapshot phase (basis)")
        else:
            local = torch.empty(cap * d * ls * 2, dtype=torch.float64, device=dev)
            _, err = _guard(fam.beyn_moments_rb, zs[idx], ws[idx], ct[idx], V[:, c0:c0 + ls], 0, cap, Q_dev=local.data_ptr(), l_total=l, col0=c0, **kw)
            fail_together(err, "snapshot phase")
            i0 = dict(fam.last_info)
        snap_cols = S * ls
        t1 = time.perf_counter()
        if world > 1:
            kact, Hk, g = fam.rb_export()
            torch.cuda.synchronize(dev)
            slabs = _allgather_dev(local, world)                                     # [rank][s][row][c_local]
            store = slabs.view(world, S, d, ls, 2).permute(1, 2, 0, 3, 4).contiguous()  # [s][row][c]: columns back in order
            del slabs
            small = torch.from_numpy(np.concatenate([Hk.ravel(), g.ravel()]).view(np.float64).copy()).to(dev)
            allsm = _allgather_dev(small, world).cpu().numpy().view(np.complex128).reshape(world, -1)
            nH = Hk.size
            Hk_all = np.concatenate([allsm[r, :nH].reshape(Hk.shape) for r in range(world)], axis=3)
            g_all = np.concatenate([allsm[r, nH:].reshape(g.shape) for r in range(world)], axis=1)
            fam.rb_import(store.data_ptr(), kact, Hk_all, g_all)
            torch.cuda.synchronize(dev)
        else:
            store = local
        t2 = time.perf_counter()
        mine2 = rest[rank::world]
        if world > 1:
            _, err = _guard(fam.beyn_moments_rb, zs[mine2], ws[mine2], ct[mine2], V, 2, S, Q_dev=store.data_ptr(), accumulate=True, **kw)
        else:        # same handle, same probe matrix: it is still in HBM
            _, err = _guard(fam.beyn_moments_rb, zs[mine2], ws[mine2], ct[mine2], None, 2, cap, Q_dev=store.data_ptr(), accumulate=True, l_total=l, **kw)
    else:
        S = min(int(S), len(zs))
        S = max(world, (S // world) * world)                      # equal snapshot shares (all_gather wants equal sizes)
        idx, rest = snapshot_split(len(zs), S)
        idx = spread_order(idx)
        S = len(idx)
        mine = idx[rank::world]
        per = len(mine)
        local = torch.empty(per * d * l * 2, dtype=torch.float64, device=dev)
        _, err = _guard(fam.beyn_moments_rb, zs[mine], ws[mine], ct[mine], V, 0, per, Q_dev=local.data_ptr(), **kw)
        fail_together(err, "snapshot phase")
        i0 = dict(fam.last_info)
        snap_cols = per * l
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        store = _allgather_dev(local, world)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        mine2 = rest[rank::world]
        _, err = _guard(fam.beyn_moments_rb, zs[mine2], ws[mine2], ct[mine2], V, 1, S, slot0=S, Q_dev=store.data_ptr(), accumulate=True, **kw)
    fail_together(err, "projected phase")
    i1 = dict(fam.last_info)
    t3 = time.perf_counter()
    allreduce_sum_(buf)
    if world > 1:
        torch.cuda.synchronize(dev)
    t4 = time.perf_counter()
    if timings is not None:
        for k, v in (("snapshots", t1 - t0), ("allgather", t2 - t1), ("projected", t3 - t2), ("allreduce", t4 - t3)):
            timings[k] = timings.get(k, 0.0) + v
    info = {"iters_max": max(i0["iters_max"], i1["iters_max"]), "iters_total": i0["iters_total"] + i1["iters_total"],
            "n_unconverged": i0["n_unconverged"] + i1["n_unconverged"], "levels": i1["levels"],
            "relres_max": max(i0["relres_max"], i1["relres_max"]), "seconds": i0["seconds"] + i1["seconds"],
            "snapshot_iters": i0["iters_total"], "projected_iters": i1["iters_total"], "snapshots": S,
            "snapshot_columns": snap_cols, "projected_columns": len(mine2) * l,
            "snapshot_split": ("hybrid" if hybrid else "columns") if by_column else ("points" if world > 1 else "none")}
    return buf, info


def beyn_distributed(L, G, l=5, K=1, N=16, V=None, pos_test_=True):
    """Ω, P, Σ of `beyn` (beyn.jl:34-110) with the quadrature sharded over the process group."""
    d = L.size()
    if V is None:
        V = initialize_V(d, l)
    A = beyn_moments_distributed(G, N, (d, V.shape[1], 2 * K), device_moment_fn(L, G, V, K, N))
    Om, P, S = moments2eigs(A, return_sigma=True)
    if pos_test_:
        Om, P = pos_test(Om, P, G)
    return Om, P, S


def _tall_gram(A, B, rows=512):
    """A^H B for tall-skinny A, B (d x l): rocBLAS runs the l x l x d product as one long-K GEMM at ~10 GFLOP/s (21 ms for
    d = 2e5, l = 16); as a batch of (l x rows)(rows x l) products plus a sum it takes a fraction of a millisecond."""
    import torch
    d = A.shape[0]
    nfull = (d // rows) * rows
    out = torch.zeros((A.shape[1], B.shape[1]), dtype=A.dtype, device=A.device)
    if nfull:
        Ab = A[:nfull].reshape(d // rows, rows, A.shape[1])
        Bb = B[:nfull].reshape(d // rows, rows, B.shape[1])
        out = torch.bmm(Ab.conj().transpose(1, 2), Bb).sum(dim=0)
    if nfull < d:
        out = out + A[nfull:].conj().T @ B[nfull:]
    return out


def warm_up_dense_linalg(device, rows=65536, cols=8, K=1):
    """One small run of the Hankel factorisation of `moments2eigs_device` on random moments of the caller's (l, K) shape: creates the
    handles and loads the kernels of the dense linear-algebra libraries torch calls there (rocSOLVER / rocBLAS / the host LAPACK:
    ~0.3 s once per PROCESS).  Library initialisation, not solver work: bench.py calls it before it starts the clock of the cold
    solver call.  K = 1: the QR path; K > 1: the Gram path, as bench.py takes them."""
    import torch
    buf = torch.randn(rows * cols * 2 * K * 2, dtype=torch.float64, device=device)
    moments2eigs_device(buf, (rows, cols, 2 * K), gram_rel_tol=1e-6 if K > 1 else 0.0)
    torch.cuda.synchronize(device)


def _svd_by_gram(B0, rel_tol):
    """Thin SVD of a tall-skinny matrix from Gram matrices, by deflation in stages so that EVERY group of singular values comes
    out to full relative accuracy when the matrix is numerically rank-deficient (a Hankel matrix of contour moments: the
    eigenvalues inside the contour, then quadrature / solver noise ten orders below).  A Gram matrix resolves singular values
    down to ~1e-8 of its largest only, so one stage takes the directions within 1e-6 of the largest singular value of what is
    left (condition number of that group <= 1e6: the Gram matrix loses nothing that matters), projects them out, and the next
    stage starts from the remainder -- whose largest singular value is again accurate -- until that is below rel_tol times the
    largest of all.  With the moments of the benchmark (gap of 1e9 after the eigenvalue group, rel_tol = 1e-6) that is one
    stage plus the Gram matrix of the remainder.  Returns (U of the kept group, S of it, Wh of it, all singular values).
    10 ms where rocSOLVER's Householder QR of the 2M x 16 matrix takes 120."""
    import torch
    n = B0.shape[1]
    stage_span = 1e-6
    rest, blocks, s_top = B0, [], None
    S = None
    for _ in range(8):                                                  # 16 decades / 6 per stage: 3 suffice in double precision
        lam, W = torch.linalg.eigh(_tall_gram(rest, rest).cpu())
        lam, W = lam.flip(0).clamp_min(0.0), W.flip(1)
        S = lam.sqrt()
        if s_top is None:
            s_top = float(S[0])
        nkept = sum(b.shape[1] for b in blocks)
        if float(S[0]) <= rel_tol * s_top or float(S[0]) == 0.0 or nkept >= n:
            break
        k = min(int((S > max(rel_tol * s_top, stage_span * float(S[0]))).sum()), n - nkept)
        U = (rest @ W[:, :k].to(B0.device)) / S[:k].to(B0.device).to(B0.dtype)
        for Ub in blocks:                                               # (later stages: rounding left along the earlier blocks)
            U = U - Ub @ _tall_gram(Ub, U)
        # second pass of the same construction on U itself (Cholesky-QR2 idea): orthonormal to rounding
        l2, W2 = torch.linalg.eigh(_tall_gram(U, U).cpu())
        U = U @ ((W2 / l2.sqrt().to(W2.dtype)) @ W2.conj().T).to(B0.device)          # U G2^{-1/2}
        blocks.append(U)
        rest = rest - U @ _tall_gram(U, rest)
    if not blocks:
        raise ValueError("_svd_by_gram: the matrix is zero")
    U = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)
    k = U.shape[1]
    # the singular triplets of the kept group, exactly: B0 = U (U^H B0) + rest
    Uc, Sc, Whc = torch.linalg.svd(_tall_gram(U, B0).cpu(), full_matrices=False)
    U = U @ Uc.to(B0.device)
    return U, Sc.to(B0.device), Whc.to(B0.device), torch.cat([Sc, S[:n - k]])


def moments2eigs_device(buf, shape, tol_sigma=0.0, gram_rel_tol=0.0):
    """`moments2eigs` (beyn.jl:289-323) with the tall-skinny part kept on the GPU (torch.linalg.svd on the moment
    buffer that the all-reduce already left in HBM); only the (lK x lK) eigenproblem runs on the host.
    buf: flat float64 CUDA tensor holding the column-major d x l x 2K complex moments.  Returns (Ω, P_device, Σ).
    gram_rel_tol > 0: the SVD through the Gram matrix (`_svd_by_gram`), keeping the singular directions above gram_rel_tol·σ₁ --
    the reference's `tol` option (beyn.jl:92-95) with a relative threshold; Σ still lists every singular value."""
    import torch
    d, l, K2 = shape
    K = K2 // 2
    A = torch.view_as_complex(buf.view(-1, 2)).view(K2, l, d).permute(2, 1, 0)       # (d, l, 2K) strided view
    B0 = torch.cat([torch.cat([A[:, :, i + j] for j in range(K)], dim=1) for i in range(K)], dim=0)
    B1 = torch.cat([torch.cat([A[:, :, i + j + 1] for j in range(K)], dim=1) for i in range(K)], dim=0)
    if gram_rel_tol > 0.0:
        U, S, Wh, Sall = _svd_by_gram(B0, gram_rel_tol)
        small = (_tall_gram(U, B1.contiguous()) @ Wh.conj().T) / S.to(U.dtype)
        Om, Pt = np.linalg.eig(small.cpu().numpy())
        P = U[:d, :] @ torch.from_numpy(Pt).to(U.device)
        return Om, P, Sall.cpu().numpy()
    if B0.shape[0] > 8 * B0.shape[1]:
        # tall-skinny: thin QR, then the SVD of the small triangular factor (the same factorisation up to rounding;
        # rocSOLVER's Jacobi SVD of the d x l matrix itself took 65 ms at d = 2e5, this takes a few)
        Qf, Rf = torch.linalg.qr(B0)
        Ur, S, Wh = torch.linalg.svd(Rf)
        U = Qf @ Ur
    else:
        U, S, Wh = torch.linalg.svd(B0, full_matrices=False)
    if tol_sigma > 0:
        m = S > tol_sigma
        U, S, Wh = U[:, m], S[m], Wh[m, :]
    small = (_tall_gram(U, B1.contiguous()) @ Wh.conj().T) / S.to(U.dtype)
    Om, Pt = np.linalg.eig(small.cpu().numpy())
    P = U[:d, :] @ torch.from_numpy(Pt).to(U.device)
    return Om, P, S.cpu().numpy()


# ------------------------------------------------------------------------------------------------------
# independent units: Bloch wave numbers, start values of the local solvers -- no data-path collective,
# one gather of the results at the end
# ------------------------------------------------------------------------------------------------------
def shard_items(n_items, rank, world):
    """indices of the units this rank owns (round-robin)"""
    return list(range(rank, n_items, world))


def gather_rows(local_rows, n_items, width):
    """Every rank contributes the rows it owns of an (n_items x width) complex table; returns the complete table on
    every rank.  One collective (sum all-reduce of a table that is zero outside the owned rows -- the shards may be
    ragged, which all_gather would not take); NaN marks a failed unit and survives the sum."""
    import torch
    tab = np.zeros((n_items, width), dtype=np.complex128)
    for k, row in local_rows.items():
        tab[k, :] = np.asarray(row, dtype=np.complex128)
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return tab
    t = torch.from_numpy(tab.view(np.float64))
    if d.get_backend() == "nccl":
        t = t.cuda()
    d.all_reduce(t)
    return t.cpu().numpy().view(np.complex128).reshape(n_items, width)


def sweep_distributed(items, unit_fn, width):
    """Run ``unit_fn(item) -> sequence of `width` complex numbers`` for the items this rank owns; gather the table.
    Returns (table (len(items) x width), {index: whatever unit_fn returned as second value}) -- the second values
    (Solutions, eigenvectors) stay on the rank that computed them."""
    rank, world = rank_world()
    rows, keep = {}, {}
    for k in shard_items(len(items), rank, world):
        res = unit_fn(items[k])
        if isinstance(res, tuple):
            rows[k], keep[k] = res
        else:
            rows[k] = res
    return gather_rows(rows, len(items), width), keep


def refine_distributed(L, starts, method=None, **kw):
    """One start value per rank (Householder/mslp from e.g. the Beyn estimates): returns the table
    [ω, iterations, flag] per start on every rank and the owned Solutions."""
    from .local_solvers import householder
    method = method or householder

    def unit(z0):
        sol, n, flag = method(L, z0, **kw)
        return [sol.params[L.eigval], n, flag], sol
    return sweep_distributed(list(starts), unit, 3)


def bloch_sweep_distributed(L, bs, starts, method=None, b_symbol="b", **kw):
    """Bloch sweep (config C4): one Bloch wave number per rank at a time, no communication until the final gather.
    starts: start values shared by all b, or a dict b -> start values (e.g. the Beyn estimates of that b).
    Returns (table (len(bs) x 3·nstart): [ω, iterations, flag] per start, {index of b: [Solution, ...]})."""
    from .local_solvers import mslp
    method = method or mslp
    nstart = max(len(starts[b]) for b in bs) if isinstance(starts, dict) else len(starts)

    def unit(b):
        L.params[b_symbol] = complex(b)
        row = np.full(3 * nstart, complex(np.nan, np.nan))
        sols = []
        for q, z0 in enumerate(starts[b] if isinstance(starts, dict) else starts):
            sol, n, flag = method(L, z0, **kw)
            row[3 * q:3 * q + 3] = [sol.params[L.eigval], n, flag]
            sols.append(sol)
        return row, sols
    return sweep_distributed(list(bs), unit, 3 * nstart)


def beyn_moments_mgpu(families, G, V, K=1, N=16, nsnap=None, points=None, zmap=None):
    """The single-process multi-GPU entry of the C ABI (``wae_beyn_moments_mgpu``; what a Julia host calls): ``families`` is a
    list of LinearOperatorFamily replicas, one per GPU (``device=g``), all with the same terms and parameters.  Returns the
    moment tensor d x l x 2K (numpy) and the merged solve statistics.  nsnap=None: the automatic rule of
    compute_moment_matrices (40 snapshot points for contours of >= 64 points, d >= 1000).  zmap = (c, rho): moments in the
    variable (z - c)/rho, as in beyn_moments_distributed_rb (the systems are those of the true z)."""
    import ctypes as C

    from .. import _lib
    from .beyn import coefficient_table
    L0 = families[0]
    zs, ws = gauss_points(G, N) if points is None else points
    zs = np.ascontiguousarray(zs, dtype=np.complex128)
    ws = np.ascontiguousarray(ws, dtype=np.complex128)
    fams = [L.ensure_solver() for L in families]
    d = L0.size()
    ct = np.ascontiguousarray(coefficient_table(L0, zs), dtype=np.complex128) if len(zs) else np.zeros((0, len(L0.terms)), dtype=np.complex128)
    if zmap is not None:
        zs = np.ascontiguousarray((zs - complex(zmap[0])) / complex(zmap[1]))
    Vf = np.asfortranarray(np.asarray(V, dtype=np.complex128))
    l = Vf.shape[1]
    if nsnap is None:
        nsnap = min(40, len(zs) // 2) if (len(zs) >= 64 and d >= 1000) else 0
    A = np.zeros((d, l, 2 * K), dtype=np.complex128, order="F")
    info = _lib.SolveInfo()
    hs = (C.c_void_p * len(fams))(*[f.handle for f in fams])
    code = _lib.check(_lib.lib().wae_beyn_moments_mgpu(hs, len(fams), len(zs), _lib.zptr(zs), _lib.zptr(ws), _lib.zptr(ct), _lib.zptr(Vf), l, K,
                                                      L0.solver_tol, L0.solver_maxit, int(nsnap), _lib.zptr(A), C.byref(info)))
    fams[0]._report(code, info, "beyn_moments_mgpu", fatal=True)
    return A, fams[0].last_info
