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


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
