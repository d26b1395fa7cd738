"""include/dspi_detmath.h as the DEVICE computes it (dspi_debug_detmath: the header compiled by hipcc for gfx950) against the host build of the
same header and against binary128: every bit, over the leveller's argument ranges, exact ties, and arguments that take the double-double step
(on the device that step is a non-inlined function: its call path is exercised here by construction, not by luck).  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

from conftest import has_gpu
from dspi_amd.host import Dspi
import test_detmath as TD

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no GPU")]


def device(d, which, a, b=None):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.empty_like(a)
    d.L.dspi_debug_detmath.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    assert d.L.dspi_debug_detmath(d.h, which, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data) == 0
    return out


def test_device_equals_host_equals_binary128():
    L = TD.build()
    d = Dspi(1, 4, device=0)
    rng = np.random.default_rng(11)
    xs, sa, sb = TD.slow_arguments(L, 64, 64)
    assert len(xs) >= 8 and len(sa) >= 8
    # log10f: the leveller's rms_sq + 1e-30f range, the neighbourhood of 1, any positive normal, the step-2 arguments (repeated: whole waves take the call)
    x = np.concatenate([(10 ** rng.uniform(-30, 1, 1_000_000)), rng.uniform(0.5, 2.0, 500_000), np.repeat(xs, 70),
                        rng.integers(0x00800000, 0x7f000000, 500_000).astype(np.uint32).view(np.float32)]).astype(np.float32)
    host = np.empty_like(x); exact = np.empty_like(x)
    L.t_log10f_v(x.ctypes.data, host.ctypes.data, x.size); L.q_log10f_v(x.ctypes.data, exact.ctypes.data, x.size)
    dev = device(d, 0, x)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32)) and np.array_equal(dev.view(np.uint32), exact.view(np.uint32))
    # powf: alpha ^ count, 10 ^ (dB / 20), exact ties, the step-2 arguments
    m = np.arange(4097, 5792, 2, dtype=np.float64)
    ta = np.concatenate([m * 2.0 ** e for e in range(-14, 3)]).astype(np.float32)
    a = np.concatenate([rng.uniform(0.9, 1.0, 1_000_000), np.full(1_000_000, 10.0), ta, np.repeat(sa, 70)]).astype(np.float32)
    b = np.concatenate([rng.integers(1, 193, 1_000_000), rng.uniform(-4, 4, 1_000_000), np.full(ta.size, 2.0), np.repeat(sb, 70)]).astype(np.float32)
    host = np.empty_like(a); exact = np.empty_like(a)
    L.t_powf_v(a.ctypes.data, b.ctypes.data, host.ctypes.data, a.size); L.q_powf_v(a.ctypes.data, b.ctypes.data, exact.ctypes.data, a.size)
    dev = device(d, 1, a, b)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32)) and np.array_equal(dev.view(np.uint32), exact.view(np.uint32))
    d.close()
