import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def report():
    """Append-only diagnostics file (gpurun_out/ is merged back from the GPU box)."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    f = open(os.path.join(d, "test_report.txt"), "a")

    def log(*a):
        print(*a, file=f, flush=True)
        print(*a, flush=True)
    yield log
    f.close()


def rel_err(a, b):
    import torch
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
