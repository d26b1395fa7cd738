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
    # the forms the chain kernels use (step 1 + exception tables): the same floats, on the device, the listed exceptions included
    import os, re
    tab = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dspi_detmath_tables.h")).read()
    def listed(name): return np.array([int(m, 16) for m in re.findall(r"\{0x([0-9a-f]+)u, 0x[0-9a-f]+u\}", re.search(r"#define %s \{(.*?)\}\s*$" % name, tab, re.M).group(1))], np.uint32)
    xl = np.concatenate([x, np.repeat(listed("DSPI_DM_LOG10_EXC").view(np.float32), 70)]).astype(np.float32)
    ref = np.empty_like(xl); L.t_log10f_v(xl.ctypes.data, ref.ctypes.data, xl.size)
    assert np.array_equal(device(d, 2, xl).view(np.uint32), ref.view(np.uint32))
    ye = np.concatenate([rng.uniform(-46, 39, 1_000_000), rng.uniform(-2, 2, 1_000_000), np.repeat(listed("DSPI_DM_EXP10_EXC").view(np.float32), 70), [0.0, 38.5, 39.0, -44.8, -45.0]]).astype(np.float32)
    ten = np.full_like(ye, 10.0); ref = np.empty_like(ye); L.t_powf_v(ten.ctypes.data, ye.ctypes.data, ref.ctypes.data, ye.size)
    ref[np.abs(ref) < np.float32(2.0 ** -126)] = 0.0          # (the device flushes subnormal results, as the oracle's MXCSR does in the chain: FTZ is part of the contract)
    assert np.array_equal(device(d, 3, ye).view(np.uint32), ref.view(np.uint32))
    al = np.array([np.exp(np.float32(-np.log(np.float32(10.0)) / np.float32(fs * t))) for fs in (44100.0, 48000.0, 96000.0) for t in (0.1, 2.0, 0.05, 1.0, 0.02, 0.5)], np.float32)
    pa = np.repeat(al, 192); pb = np.tile(np.arange(1, 193, dtype=np.float32), len(al))
    ref = np.empty_like(pa); L.t_powf_v(pa.ctypes.data, pb.ctypes.data, ref.ctypes.data, pa.size)
    assert np.array_equal(device(d, 4, pa, pb).view(np.uint32), ref.view(np.uint32))
    d.close()
