import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "small_scene.npz"))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """State in the log WHY the launcher tests ("train_tensoIR.py runs unmodified") did not run on a box without a
    TensoIR checkout, and how to run them."""
    ref = os.environ.get("TENSOIR_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "train_tensoIR.py")):
        n = sum(1 for r in terminalreporter.stats.get("skipped", []) if "test_launcher" in getattr(r, "nodeid", ""))
        if n:
            terminalreporter.write_line(
                f"SKIPPED: {n} tests/test_launcher.py tests -- no TensoIR checkout on this box (TENSOIR_REFERENCE={ref} has no "
                "train_tensoIR.py; the reference is Python and does not travel to the GPU box).  Where a checkout and a GPU exist "
                "together: TENSOIR_REFERENCE=<checkout> python -m pytest tests/test_launcher.py -m gpu.  The CPU leg of those tests "
                "(launcher rebinding, argument plumbing up to the first kernel call) runs in the build container; the same call "
                "sequence without the scripts: tests/test_gpu_train_loop.py (evidence of earlier rounds: profiles/r05_launcher_tests.log)")
