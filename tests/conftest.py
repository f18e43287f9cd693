import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure, oracle/)."""
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def cuda():
    """torch is only plumbing here: device buffers for depth frames."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    from semantic_slam_amd import capi
    capi.load()  # fail loudly if the HIP library is missing: there is no fallback path
    return torch
