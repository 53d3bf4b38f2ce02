"""`python bench.py --gpus N` must BE an N-rank run: started without a launcher it starts its own ranks as a
child process (torch.distributed.run, before torch is imported in the parent), relays rank 0's line and the exit code; started by a
launcher whose WORLD_SIZE disagrees with --gpus it refuses.  `--rehearse` stops after the rendezvous, so this runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WAE_BENCH_BACKEND="gloo", **kw)
    return env


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    # (--l and --n are prefixes of torch.distributed.run's own options: the launcher must pass their long spellings on)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--l", "4", "--n=0.5"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j["ranks_counted"] == 2 and j["launched_by"] == "bench.py"


def test_gpus_1_is_a_plain_single_rank_run():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rehearse"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert _json_line(p.stdout)["n_gpus"] == 1


def test_world_size_that_disagrees_with_gpus_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--mgpu", "--rehearse"], env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "single-process" in p.stderr


def test_a_failing_rank_fails_the_launching_process():
    """no GPU here: without --rehearse every rank stops at the 'needs a GPU' assertion; the parent must report that, not success"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--preset", "tiny"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
