import os
import sys

import pytest

# the product's default mode is "fast_reflists" (FAST arithmetic, the reference's tile lists); the parity tests compare against the CPU oracle bit for bit unless a test
# selects a mode itself, so the suite's default is "exact" (read by instascene_amd.rasterizer at import).  The shipped default
# is covered by the tests that select MODE_FAST themselves (test_gpu_fuzz.py: 40 scenes gated by cause; test_gpu_fullsize.py:
# C1 / C3 / C5 at full size; test_gpu_rasterizer.py; the harness / drop-in tests run both), and the whole suite is green under
# `ISR_MODE=fast_reflists python -m pytest tests -m gpu` (and ISR_MODE=fast, the shorter lists) as well
os.environ.setdefault("ISR_MODE", "exact")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
