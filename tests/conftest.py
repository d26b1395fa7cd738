import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def product_lib():
    """The C-ABI library must exist: tests never fall back to anything else."""
    from dspi_amd import host
    if not host.LIB_PATH.exists():
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "dspi_amd", "csrc"), "libdspi_mi355x.so"], check=True)
    return host.lib()


@pytest.fixture(autouse=True)
def q28_wave_layout(request, monkeypatch):
    """The Q28 chain kernel has two wave layouts, chosen by launch size (dspi_kernels.hip chain_kernel NW: seven waves up to one
    workgroup per CU, four beyond).  Test-sized launches would all take the first; so every other GPU test (by a hash of its id)
    forces the four-wave layout, and the suite covers both with everything it has.  test_q28_wave_layouts sets its own."""
    if request.node.get_closest_marker("gpu") and "DSPI_Q28_WAVES" not in os.environ:
        import zlib
        if zlib.crc32(request.node.nodeid.encode()) & 1:
            monkeypatch.setenv("DSPI_Q28_WAVES", "4")
    yield


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
