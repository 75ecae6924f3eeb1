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


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
