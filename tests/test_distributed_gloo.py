"""world_size-2 gloo test (CPU) of the N>1 path: quadrature points sharded round-robin over the ranks, partial
moment tensors summed with one all-reduce, result identical to the single-process moments.  The per-rank moment
producer is injected: here the CPU oracle (the HIP producer needs a GPU); the sharding / reduction / reshaping code
is exactly the one bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wae_amd  # noqa: F401
    from oracle import fixtures as F
    from oracle import solvers as OS
    from wae_amd.nlevp.beyn import moments2eigs
    from wae_amd.nlevp.distributed import beyn_moments_distributed, rank_world

    assert rank_world() == (rank, world)
    T = F.qep1()
    G = [2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j]
    V = OS.initial_V(3, 6)
    K, N = 2, 16

    def moment_fn(zs, ws):            # partial moments of this rank's shard, CPU oracle arithmetic
        A = np.zeros((3, 6, 2 * K), dtype=complex)
        for z, w in zip(zs, ws):
            X = OS._solve(T(z), V) * w
            for p in range(2 * K):
                A[:, :, p] += z ** p * X
        return torch.from_numpy(np.asfortranarray(A).reshape(-1, order="F").view(np.float64).copy())

    A = beyn_moments_distributed(G, N, (3, 6, 2 * K), moment_fn)
    Om, P = moments2eigs(A)
    if rank == 0:
        full = OS.compute_moment_matrices(T, G, V, K=K, N=N)
        q.put((float(np.max(np.abs(A - full))), [complex(x) for x in Om]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_beyn_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, Om = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-12
    for w in (1 / 3, 0.5, 1.0, 1j, -1j):
        assert min(abs(o - w) for o in Om) < 1e-9


def _failing_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wae_amd  # noqa: F401
    from wae_amd.nlevp.distributed import RankFailure, beyn_moments_distributed

    def moment_fn(zs, ws):            # rank 1's share fails (a stalled inner solve raises in DeviceFamily._report)
        if rank == 1:
            raise ValueError("inner solve did not converge")
        return torch.zeros(3 * 6 * 4 * 2, dtype=torch.float64)

    try:
        beyn_moments_distributed([2 + 2j, -2 + 2j, -2 - 2j, 2 - 2j], 16, (3, 6, 4), moment_fn)
    except RankFailure:
        q.put((rank, "RankFailure"))
        raise SystemExit(3)
    except ValueError:
        q.put((rank, "ValueError"))
        raise SystemExit(4)
    q.put((rank, "no error"))


def test_a_failing_rank_takes_every_rank_down_before_the_collective():
    """a rank whose share raises must not leave the others blocked in the all-reduce until the process-group
    timeout -- every rank agrees on a status word first and all of them raise."""
    import time
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert time.time() - t0 < 120
    assert got == {0: "RankFailure", 1: "ValueError"}
    assert [p.exitcode for p in procs] == [3, 4]


def test_shard_covers_all_points_once():
    sys.path.insert(0, ROOT)
    import wae_amd  # noqa: F401
    from wae_amd.nlevp.distributed import shard_points
    z = np.arange(13) + 0j
    for world in (1, 2, 3, 8, 16):
        got = np.concatenate([shard_points(z, z, r, world)[0] for r in range(world)])
        assert sorted(got.real.astype(int).tolist()) == list(range(13))


# ------------------------------------------------------------------------------------------------------
# independent units: Bloch wave numbers / start values dealt to the ranks, one gather at the end
# ------------------------------------------------------------------------------------------------------
def _sweep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wae_amd  # noqa: F401
    from oracle import bloch as OB
    from oracle import fixtures as F
    from oracle import solvers as OS
    from wae_amd.helmholtz import annulus
    from wae_amd.nlevp.distributed import bloch_sweep_distributed, refine_distributed

    # (1) one start value per rank: Rijke tube, the CPU oracle's householder injected as the method
    L = F.rijke_family(n=0.0)
    starts = [2 * np.pi * 250.0, 2 * np.pi * 650.0, 2 * np.pi * 300.0]          # ragged: 2 + 1 over two ranks
    tab, sols = refine_distributed(L, starts, method=OS.householder, maxiter=12, tol=1e-10)
    assert sorted(sols) == list(range(rank, 3, world))
    # (2) Bloch sweep on a small unit cell, mslp from shared start values
    cell = annulus.build_unit_cell(grid=(4, 12, 4), DOS=12, tau=2e-4)
    Lb = OB.bloch_family(cell["terms_ext"], cell["nsector"], 12, tau=2e-4, n=0.0)
    bs = [0, 1, 2]
    tab2, keep = bloch_sweep_distributed(Lb, bs, [2 * np.pi * 200.0], method=OS.mslp, maxiter=15, tol=1e-9)
    if rank == 0:
        q.put((tab, tab2))
    dist.barrier()
    dist.destroy_process_group()


def test_sweeps_gather_results_of_all_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tab, tab2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # Rijke passive modes (tests/golden: the n=0 spectrum has its first two modes near 272 and 695 Hz)
    f = tab[:, 0].real / 2 / np.pi
    assert abs(f[0] - f[2]) < 1e-6 and 200 < f[0] < 350 and 600 < f[1] < 800
    assert np.all(tab[:, 2].real == 1)                       # flag: converged
    # single-process reference of the Bloch sweep
    sys.path.insert(0, ROOT)
    import wae_amd  # noqa: F401
    from oracle import bloch as OB
    from oracle import solvers as OS
    from wae_amd.helmholtz import annulus
    cell = annulus.build_unit_cell(grid=(4, 12, 4), DOS=12, tau=2e-4)
    Lb = OB.bloch_family(cell["terms_ext"], cell["nsector"], 12, tau=2e-4, n=0.0)
    for k, b in enumerate([0, 1, 2]):
        Lb.params["b"] = complex(b)
        sol, n, flag = OS.mslp(Lb, 2 * np.pi * 200.0, maxiter=15, tol=1e-9)
        assert abs(tab2[k, 0] - sol.params["ω"]) <= 1e-9 * abs(tab2[k, 0]) and tab2[k, 2].real == flag
    assert abs(tab2[1, 0] - tab2[0, 0]) > 1.0               # different wave numbers, different eigenvalues
