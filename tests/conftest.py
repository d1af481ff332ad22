import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; builds oracle/libmfsr_oracle.so on demand)."""
    from tests.kernels import OracleKernels

    return OracleKernels()


@pytest.fixture(scope="session")
def hip():
    """The product path: HIP kernels through the C-ABI on cuda:0.  No fallback."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no HIP device is visible")
    from tests.kernels import HipKernels

    return HipKernels()
