import os
import sys

import pytest

# the product's default arithmetic mode is "fast"; the parity tests compare against the CPU oracle bit for bit unless a test
# selects a mode itself, so the suite's default is "exact" (read by instascene_amd.rasterizer at import)
os.environ.setdefault("ISR_MODE", "exact")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
