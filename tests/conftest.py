import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # another build of the library (same-box A/B runs, tools/ab_lib.sh) must never be what the tests load — popped FIRST (a pure
    # os.environ operation), so the forkserver below and every rank it forks inherit an environment without it
    if os.environ.pop("MVI_HIP_LIB", None):
        sys.stderr.write("[mvi] tests ignore MVI_HIP_LIB: the in-tree multiview_inpaint_amd/csrc/libmvi_hip.so is what is tested\n")
    # then, before torch is imported or any HIP call can run in this process: the clean parent of the rank tests
    _start_rank_launcher()
    # GPU runs: MIOpen looks its convolution solvers up in the find-db recorded on the MI355X (the kernels the bench uses)
    # instead of choosing by heuristic; must happen before the process first touches MIOpen. Harmless without a GPU.
    try:
        from multiview_inpaint_amd.svd import bench_svd
        bench_svd.use_shipped_miopen_db()
    except Exception:
        pass


_LAUNCHER = None


def _start_rank_launcher():
    """A multiprocessing forkserver started NOW — the first thing pytest_configure does, before torch is imported and before any
    HIP / HSA call can have run in this process — so the server (a fresh fork + exec child) and every rank it forks later for
    tests/test_dist_gpu_ranks.py are clean. Only on a box with a GPU, told by the driver's device node (no runtime call)."""
    global _LAUNCHER
    try:
        if not os.path.exists("/dev/kfd") or os.environ.get("MVI_NO_RANK_LAUNCHER"):
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
