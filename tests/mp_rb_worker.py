"""Worker of tests/test_gpu_distributed.py: one rank of a two-process run of beyn_moments_distributed_rb (the path bench.py
takes for N > 1), both ranks on device 0, process group over gloo (RCCL needs one GPU per rank).
usage: mp_rb_worker.py <out.npz> <l> <S>   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, l, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import wae_amd  # noqa: F401
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp.distributed import beyn_moments_distributed_rb
    L, pb = annulus_family("small", n=1.0, tau=2e-4)
    L.solver_tol = 1e-11
    L.solver_ref = 2 * np.pi * 500.0
    G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
    d = pb["d"]
    V = np.random.default_rng(5).standard_normal((d, l)) + 0j
    timings = {}
    if os.environ.get("WAE_TEST_FAIL_RANK") == str(rank):       # this rank's inner solves cannot converge: one iteration allowed
        L.solver_maxit = 1
    try:
        buf, info = beyn_moments_distributed_rb(L, G, V, 1, 16, S, timings=timings)
    except Exception as e:          # noqa: BLE001
        print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
        L._drop_device()
        sys.exit(7)
    torch.cuda.synchronize()
    A = buf.cpu().numpy().view(np.complex128).reshape((d, l, 2), order="F")
    # every rank holds the reduced tensor: check that on the ranks themselves, rank 0 reports
    t = torch.from_numpy(np.ascontiguousarray(A).view(np.float64).copy())
    t0 = t.clone()
    dist.broadcast(t0, 0)
    same = bool(torch.equal(t, t0))
    flags = [None] * world
    dist.all_gather_object(flags, same)
    if rank == 0:
        np.savez(out, A=A, split=info["snapshot_split"], all_same=all(flags), n_unconverged=info["n_unconverged"],
                 snapshots=info["snapshots"], snapshot_columns=info["snapshot_columns"], projected_columns=info["projected_columns"])
    dist.barrier()
    dist.destroy_process_group()
    L._drop_device()


if __name__ == "__main__":
    main()
