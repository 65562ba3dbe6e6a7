import os
import sys

# Before anything loads libgomp (torch, the oracle's OpenMP build): idle OpenMP workers must SLEEP, not spin.  The product's host
# side runs its own std::threads (csrc/host/parallel.h) next to the twin's and torch's OpenMP pools in one pytest process; with
# libgomp's default active waiting the spinning workers take the cores the next test needs, and the CPU suite takes 17 minutes on
# an 8-CPU box instead of 4.5 (measured, round 4).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
