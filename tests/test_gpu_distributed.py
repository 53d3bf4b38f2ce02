"""The N > 1 path of bench.py (nlevp/distributed.py: beyn_moments_distributed_rb -- all-gather of the snapshot bases or raw
snapshots, wae_rb_export / wae_rb_import, round-robin points, all-reduce of the moments) executed in TWO real processes on
one GPU (gloo process group: RCCL wants one GPU per rank), against the single-process moments; and the single-process
multi-GPU entry of the C ABI, wae_beyn_moments_mgpu, on the one device this box has (its RCCL collectives run with one rank)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAMMA = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference(l):
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp import compute_moment_matrices
    L, pb = annulus_family("small", n=1.0, tau=2e-4)
    L.solver_tol = 1e-11
    L.solver_ref = 2 * np.pi * 500.0
    V = np.random.default_rng(5).standard_normal((pb["d"], l)) + 0j
    A = compute_moment_matrices(L, GAMMA, V, K=1, N=16, rb=0)
    L._drop_device()
    return A


@pytest.mark.parametrize("l,branch", [(4, "columns"), (3, "points")])
def test_two_process_projected_guesses_match_single_process(tmp_path, l, branch):
    """l = 4 on two ranks: the snapshot phase is split by probe column (bases all-gathered and imported); l = 3: not
    divisible, the snapshot POINTS are split and every rank rebuilds the basis from the gathered raw snapshots."""
    A0 = _reference(l)
    port = _free_port()
    out = str(tmp_path / "rank0.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_rb_worker.py"), out, str(l), "24"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        o, _ = p.communicate(timeout=600)
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r = np.load(out)
    assert str(r["split"]) == branch and bool(r["all_same"]) and int(r["n_unconverged"]) == 0
    assert int(r["snapshots"]) == 24
    assert np.max(np.abs(r["A"] - A0)) <= 1e-8 * np.max(np.abs(A0))


@pytest.mark.parametrize("l,nsnap", [(4, 24), (4, 0), (5, 24)])
def test_single_process_mgpu_entry_on_one_device(l, nsnap):
    """wae_beyn_moments_mgpu with ngpu = 1: the same code as on 8 GPUs with one rank -- per-device host thread, snapshot phase,
    RCCL all-gather of the basis (loaded by dlopen), slab merge, wae_rb_import, projected phase, RCCL reduce to device 0 --
    against the plain single-GPU moments.  (l = 5, nsnap = 24 with one device still takes the column branch; the point
    branch needs ngpu > 1 and is covered by the two-process test above and by construction.)"""
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp.distributed import beyn_moments_mgpu
    A0 = _reference(l)
    L, pb = annulus_family("small", n=1.0, tau=2e-4)
    L.solver_tol = 1e-11
    L.solver_ref = 2 * np.pi * 500.0
    V = np.random.default_rng(5).standard_normal((pb["d"], l)) + 0j
    A, info = beyn_moments_mgpu([L], GAMMA, V, K=1, N=16, nsnap=nsnap)
    assert info["n_unconverged"] == 0
    assert np.max(np.abs(A - A0)) <= 1e-8 * np.max(np.abs(A0))
    # a second call reuses the cached communicator; duplicate devices are refused
    A2, _ = beyn_moments_mgpu([L], GAMMA, V, K=1, N=16, nsnap=nsnap)
    assert np.max(np.abs(A2 - A0)) <= 1e-8 * np.max(np.abs(A0))
    from wae_amd import _lib
    with pytest.raises(_lib.WaeError):
        beyn_moments_mgpu([L, L], GAMMA, V, K=1, N=16, nsnap=nsnap)
    L._drop_device()
