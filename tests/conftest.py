import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runs: MIOpen looks its convolution solvers up in the find-db recorded on the MI355X (the kernels the bench uses)
    # instead of choosing by heuristic; must happen before the process first touches MIOpen. Harmless without a GPU.
    try:
        from multiview_inpaint_amd.svd import bench_svd
        bench_svd.use_shipped_miopen_db()
    except Exception:
        pass
    _start_rank_launcher()


_LAUNCHER = None


def _start_rank_launcher():
    """A multiprocessing forkserver started NOW — at configure time nothing in this process has initialised the GPU, so the server
    (and every rank it forks later for tests/test_dist_gpu_ranks.py) is clean. Only on a box with a GPU; counting devices does not
    initialise them."""
    global _LAUNCHER
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
        import multiprocessing as mp
        from multiprocessing import forkserver
        ctx = mp.get_context("forkserver")
        forkserver.ensure_running()
        _LAUNCHER = ctx
    except Exception:
        _LAUNCHER = None


@pytest.fixture(scope="session")
def rank_launcher():
    return _LAUNCHER


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
