import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Host BLAS threads: a GPU box shows the test process all 256 cores of its host but gives it a share of 16; OpenBLAS then starts 256
# threads and scipy's SuperLU (the sparse-LU references of the parity tests) takes 2.9 s per factorisation of an 8 736-DoF matrix
# instead of 0.1 s.  Four threads here and in the child processes the tests start (they inherit the environment).
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
_BLAS_LIMIT = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    global _BLAS_LIMIT
    try:                                              # (numpy may have been imported before this file: limit the loaded pools too)
        from threadpoolctl import threadpool_limits
        _BLAS_LIMIT = threadpool_limits(limits=4)
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def c3_family():
    """ONE family of BASELINE configs[2] (995 328 DoF) for the whole GPU session: the numpy assembly of the annulus (8 s), the upload
    (2 s) and the multigrid set-up (1 s) are paid once; tests/test_gpu_fullsize.py::test_c3_one_million_dof_pass and the C3 case of
    tests/test_gpu_tile_parity.py share it (keep `pytest -m gpu` short).  Solver options as bench.py's."""
    import numpy as np
    from wae_amd.helmholtz.family import annulus_family
    L, pb = annulus_family("C3", tau=2e-4)
    L.solver_tol = 1e-10
    L.solver_ref = 2 * np.pi * 500.0
    L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
    yield L, pb
    L._drop_device()
