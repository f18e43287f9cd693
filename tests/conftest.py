import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Bring the binaries up to date with the sources (they are git-ignored; `make` is dependency-tracked, so this is a
    no-op when nothing changed; hipcc cross-compiles gfx950 without a GPU, ~60 s from scratch).  On the GPU box the
    prebuilt files travel with the snapshot."""
    import shutil
    import subprocess
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "semantic_slam_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


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
