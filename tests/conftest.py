import os
import sys

import pytest

# the GEMM launcher latches its tuning knobs at the first call; test processes re-read them per call so that a test
# can force a tile variant (tests/test_ops_gpu.py::test_gemm_tail_split)
os.environ.setdefault("CVLM_GEMM_VARIANT_LIVE", "1")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
