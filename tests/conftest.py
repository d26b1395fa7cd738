import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both_layouts: run the GPU test once on the packed float kernel and once on the latency layout (DSPI_F32_LAYOUT)")
    config.addinivalue_line("markers", "auto_layout: leave the choice between the packed kernel and the latency layout to the library's size rule")
    config.addinivalue_line("markers", "all_layouts: both_layouts plus a third run under the library's own size rule")


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
    """The Q28 chain runs on three kernels, chosen by size: the latency layout (dspi_chain_q28_lat.inc: one stream per workgroup) for
    contexts that leave the chip underfilled, and chain_kernel's two wave layouts beyond (seven waves up to one workgroup per CU, four
    beyond).  Test-sized contexts would all take the first; so by a hash of the test's id a third of the GPU tests force the four-wave
    layout and a third the seven-wave one (DSPI_Q28_WAVES, which also keeps the latency layout out), and the suite covers all three with
    everything it has.  test_q28_wave_layouts sets its own."""
    if request.node.get_closest_marker("gpu") and "DSPI_Q28_WAVES" not in os.environ and "DSPI_Q28_LAYOUT" not in os.environ:
        import zlib
        h = zlib.crc32(request.node.nodeid.encode()) % 3
        if h: monkeypatch.setenv("DSPI_Q28_WAVES", "4" if h == 1 else "7")
    yield


@pytest.fixture(autouse=True)
def _layout(request, monkeypatch):
    """Float contexts of test size (a few hundred streams) would all take the latency layout (dspi_chain_skew*.inc: the library's choice
    for launches that leave the chip underfilled) and the packed kernel — the one the headline is measured on — would only be seen by the
    full-size tests.  So GPU tests pin the packed kernel (DSPI_F32_LAYOUT=packed) unless they say otherwise: `both_layouts` runs a test
    once on each, `auto_layout` leaves the size rule in charge, and a test that sets the variable itself (monkeypatch) wins."""
    if not request.node.get_closest_marker("gpu"):
        yield; return
    param = getattr(request, "param", "packed")
    if request.node.get_closest_marker("auto_layout") or param == "auto": monkeypatch.delenv("DSPI_F32_LAYOUT", raising=False)
    else: monkeypatch.setenv("DSPI_F32_LAYOUT", "skew" if param == "latency" else "packed")
    yield


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("all_layouts") and "_layout" in metafunc.fixturenames:
        metafunc.parametrize("_layout", ["packed", "latency", "auto"], indirect=True)      # "auto": the library's own size rule picks the kernel
    elif metafunc.definition.get_closest_marker("both_layouts") and "_layout" in metafunc.fixturenames:
        metafunc.parametrize("_layout", ["packed", "latency"], indirect=True)


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
