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


@pytest.mark.parametrize("l,branch", [(4, "hybrid"), (4, "columns"), (3, "points")])
def test_two_process_projected_guesses_match_single_process(tmp_path, l, branch):
    """l = 4 on two ranks, "hybrid" (the default): snapshot POINTS shared out for the solves (full-width batches from zero, mode 3), one
    all-to-all of the raw solutions, probe COLUMNS shared out for the basis (mode 4), bases all-gathered and imported; "columns"
    (WAE_SNAPSHOT_SPLIT=columns): every rank solves all snapshot points for its columns progressively; l = 3: not divisible,
    the snapshot POINTS are split and every rank rebuilds the whole basis from the gathered raw snapshots."""
    A0 = _reference(l)
    port = _free_port()
    out = str(tmp_path / "rank0.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", WAE_SNAPSHOT_SPLIT="columns" if branch == "columns" else "hybrid")
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


def test_two_process_run_fails_on_every_rank_when_one_rank_stalls(tmp_path):
    """rank 1 is allowed one Krylov iteration per solve: its snapshot phase raises (DeviceFamily: contour integrals are fatal);
    rank 0 must not wait in the all-gather -- both leave promptly with the worker's error code"""
    import time
    port = _free_port()
    procs = []
    t0 = time.time()
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", WAE_TEST_FAIL_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_rb_worker.py"), str(tmp_path / "x.npz"), "4", "24"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [7, 7], "\n".join(logs)
    assert "RankFailure" in logs[0] and "WaeError" in logs[1], "\n".join(logs)
    assert time.time() - t0 < 200


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


@pytest.mark.parametrize("l,nsnap,ngpu,split", [(4, 24, 2, "hybrid"), (4, 24, 2, "columns"), (3, 24, 2, "-"), (4, 0, 2, "-"), (6, 25, 3, "hybrid"),
                                                (6, 24, 3, "columns"), (5, 24, 3, "-")])
def test_single_process_mgpu_entry_with_several_virtual_ranks(monkeypatch, l, nsnap, ngpu, split):
    """The G > 1 logic of wae_beyn_moments_mgpu on a one-GPU box: WAE_MGPU_EXCHANGE=copy replaces the two RCCL collectives by
    device-to-device copies (and a fixed-order sum), which allows several handles on ONE device to act as ranks.  Everything else
    is the code an 8-GPU node runs: a host thread and a stream per rank, column shares (l divisible by the rank count: the basis
    slabs gathered and merged back into column order by merge_slabs_kernel with G > 1, the projected terms interleaved, wae_rb_import
    on every rank) -- with the snapshot phase shared out "hybrid" (points for the solves, columns for the basis: modes 3 and 4,
    slice_cols_kernel; 25 snapshots on 3 ranks: one joins the remaining points) or by "columns" -- or point shares (raw snapshots
    gathered, every rank rebuilds the basis, slot0 = the used snapshot count), round-robin projected phase, reduction to rank 0.
    Against the plain single-GPU moments (<= 1e-8)."""
    from wae_amd import _lib
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp.distributed import beyn_moments_mgpu
    monkeypatch.setenv("WAE_MGPU_EXCHANGE", "copy")
    monkeypatch.setenv("WAE_SNAPSHOT_SPLIT", "columns" if split == "columns" else "hybrid")
    A0 = _reference(l)
    fams = []
    for _ in range(ngpu):
        L, pb = annulus_family("small", n=1.0, tau=2e-4)
        L.solver_tol = 1e-11
        L.solver_ref = 2 * np.pi * 500.0
        fams.append(L)
    V = np.random.default_rng(5).standard_normal((pb["d"], l)) + 0j
    A, info = beyn_moments_mgpu(fams, GAMMA, V, K=1, N=16, nsnap=nsnap)
    assert info["n_unconverged"] == 0
    assert np.max(np.abs(A - A0)) <= 1e-8 * np.max(np.abs(A0))
    with pytest.raises(_lib.WaeError):                      # the same handle twice is still refused
        beyn_moments_mgpu([fams[0], fams[0]], GAMMA, V, K=1, N=16, nsnap=nsnap)
    for L in fams:
        L._drop_device()


def _bench(args, **envkw):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", WAE_BENCH_PREFAULT_GB="0", **envkw)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--preset", "small", "--N", "32", "--l", "8", "--K", "2", "--steps", "1",
                        "--warmup", "1", "--tol", "1e-11", "--no-newton", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_and_matches_one_rank():
    """`python bench.py --gpus 2` (no launcher): two ranks over gloo on this box's one GPU -- the N > 1 path of the scaling bench,
    entered the way a driver that only knows `--gpus N` enters it.  Same eigenvalues as the one-rank run."""
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2"], WAE_BENCH_BACKEND="gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["rccl_ranks"] == 2
    assert two["roofline"]["frac"] > 0 and two["solver"]["n_unconverged"] == 0
    assert one["eigenpairs"] == two["eigenpairs"] > 0
    a, b = np.array(one["eigenvalues_hz"]), np.array(two["eigenvalues_hz"])
    assert np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(a))


def test_bench_mgpu_mode_drives_two_handles_from_one_process():
    """`bench.py --mgpu --gpus 2`: the single-process entry a Julia host uses (wae_beyn_moments_mgpu), two handles on this one
    device as virtual ranks (WAE_MGPU_EXCHANGE=copy)."""
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2", "--mgpu"], WAE_MGPU_EXCHANGE="copy")
    assert two["n_gpus"] == 2 and "wae_beyn_moments_mgpu" in two["launch"]
    assert one["eigenpairs"] == two["eigenpairs"] > 0
    a, b = np.array(one["eigenvalues_hz"]), np.array(two["eigenvalues_hz"])
    assert np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(a))
