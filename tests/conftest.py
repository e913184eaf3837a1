import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure): C restatement of the PCL 1.8.0 path"""
    import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """the tests drive the in-tree HIP library: build it when it is missing or older than its sources (hipcc
    cross-compiles gfx950 without a GPU; a no-op when pcl_tracking_amd/_build/libpft_hip.so is current)"""
    from pcl_tracking_amd import build

    build.build()
    yield


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """PyTorch bundles its own HIP runtime; in a process that also uses torch on the GPU (test_gpu_dist,
    bench.py) torch has to initialise it before libpft_hip.so does.  No-op without a GPU."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield
