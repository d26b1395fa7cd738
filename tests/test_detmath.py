"""include/dspi_detmath.h (the leveller's per-block log10f/powf, shared by oracle and device code) against glibc."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_detmath_test.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(ROOT, "tests", "_detmath_test.c")
    open(src, "w").write('#include "../include/dspi_detmath.h"\n'
                         'float t_log10f(float x){return dspi_det_log10f(x);}\nfloat t_powf(float a,float b){return dspi_det_powf(a,b);}\n'
                         'void t_log10f_v(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=dspi_det_log10f(x[i]);}\n'
                         'void t_powf_v(const float*a,const float*b,float*y,int n){for(int i=0;i<n;i++)y[i]=dspi_det_powf(a[i],b[i]);}\n')
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, src], check=True)
    L = ctypes.CDLL(SO)
    L.t_log10f_v.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.t_powf_v.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_log10f_within_1ulp_of_double_reference(lib):
    rng = np.random.default_rng(1)
    x = np.concatenate([10 ** rng.uniform(-30, 1, 200000), rng.uniform(0.5, 2.0, 100000), [1e-30, 1.0, 0.7079458]]).astype(np.float32)
    y = np.empty_like(x)
    lib.t_log10f_v(x.ctypes.data, y.ctypes.data, len(x))
    ref = np.log10(x.astype(np.float64)).astype(np.float32)       # correctly rounded float of the double result
    d = ulp_diff(y, ref)
    assert d.max() <= 1 and (d > 0).mean() < 1e-4


def test_powf_within_1ulp_of_double_reference(lib):
    rng = np.random.default_rng(2)
    # the two call shapes of the leveller: alpha^count and 10^(dB/20)
    a = np.concatenate([rng.uniform(0.9, 1.0, 150000), np.full(150000, 10.0)]).astype(np.float32)
    b = np.concatenate([rng.integers(1, 193, 150000).astype(np.float32), rng.uniform(-2, 2, 150000).astype(np.float32)])
    y = np.empty_like(a)
    lib.t_powf_v(a.ctypes.data, b.ctypes.data, y.ctypes.data, len(a))
    ref = np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    d = ulp_diff(y, ref)
    assert d.max() <= 1 and (d > 0).mean() < 1e-4


def test_against_glibc_float_functions(lib):
    """glibc's powf/log10f are themselves not correctly rounded; the distance is reported and bounded."""
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    libm.log10f.restype = ctypes.c_float; libm.log10f.argtypes = [ctypes.c_float]
    lib.t_powf.restype = ctypes.c_float; lib.t_powf.argtypes = [ctypes.c_float, ctypes.c_float]
    lib.t_log10f.restype = ctypes.c_float; lib.t_log10f.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(3)
    worst = 0
    for _ in range(20000):
        x = np.float32(10 ** rng.uniform(-12, 0.5))
        worst = max(worst, int(ulp_diff(np.array([lib.t_log10f(x)], np.float32), np.array([libm.log10f(x)], np.float32))[0]))
        db = np.float32(rng.uniform(-1, 1.75))
        worst = max(worst, int(ulp_diff(np.array([lib.t_powf(10.0, db)], np.float32), np.array([libm.powf(10.0, db)], np.float32))[0]))
    assert worst <= 2


def test_edge_cases(lib):
    lib.t_powf.restype = ctypes.c_float; lib.t_powf.argtypes = [ctypes.c_float, ctypes.c_float]
    assert lib.t_powf(0.5, 0.0) == 1.0 and lib.t_powf(0.0, 3.0) == 0.0 and lib.t_powf(1.0, 96.0) == 1.0
    assert lib.t_powf(10.0, 0.0) == 1.0 and abs(lib.t_powf(10.0, 1.0) - 10.0) == 0.0
