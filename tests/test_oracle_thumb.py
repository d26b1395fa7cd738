"""The oracle's Q28 block biquad (oracle/orc_chain.c:q28_biquad_block, exported for this test) against the reference's own assembly:
executed here from its text by tests/thumb.py when the reference tree is present, and through the vectors that execution produced
(tests/golden/q28_thumb_biquad.npz) everywhere else.  Closes the one leaf the oracle could only restate by reading."""
import ctypes as C
import os

import numpy as np
import pytest

import orclib
from thumb import Thumb

ASM = "/root/reference/firmware/DSPi/dsp_process_rp2040.S"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "q28_thumb_biquad.npz")


def oracle_block(lib, coef, state, bypass, x):
    coef = np.ascontiguousarray(coef, dtype=np.int32); st = np.ascontiguousarray(state, dtype=np.int32).copy()
    byp = np.ascontiguousarray(bypass, dtype=np.uint8); y = np.ascontiguousarray(x, dtype=np.int32).copy()
    lib.orc_debug_q28_biquad_block(coef.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), byp.ctypes.data_as(C.c_void_p),
                                   y.ctypes.data_as(C.c_void_p), C.c_uint32(len(y)), C.c_uint32(coef.shape[0]))
    return y, st


def libs():
    out = [("standalone", orclib.load(0, False))]
    if orclib.ref_available(0, "ref"): out.append(("ref leaf build", orclib.load(0, True)))
    return out


def test_golden_vectors_from_the_executed_assembly():
    g = np.load(GOLD)
    for name, lib in libs():
        for k in range(int(g["n"])):
            y, st = oracle_block(lib, g[f"coef{k}"], g[f"state{k}"], g[f"bypass{k}"], g[f"x{k}"])
            assert np.array_equal(y, g[f"y{k}"]) and np.array_equal(st, g[f"state_out{k}"]), (name, k)


def test_fast_mul_q28_inside_the_assembly():
    """One band with b1 = b2 = a1 = a2 = 0 and zero state: y = fast_mul_q28(b0, x) as the assembly inlines it (:272-285) — identities
    from dsp_pipeline.c:47-58: 1.0 * x = x, and the dropped low x low partial product."""
    if not os.path.exists(ASM): pytest.skip("reference not present")
    import sys; sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_thumb_golden import run_thumb, CBC
    t = Thumb(open(ASM).read(), symbols={"channel_band_counts": CBC})
    x = np.array([123456789, -987654321, 1 << 28, -(1 << 28), 0x7FFFFFFF, -0x80000000, 0xFFFF, 0x10000], dtype=np.int64).astype(np.int32)
    coef = np.array([[1 << 28, 0, 0, 0, 0]], dtype=np.int32)
    y, st = run_thumb(t, coef, np.zeros((1, 2), np.int32), np.zeros(1, np.uint8), x)
    assert np.array_equal(y, x)                                              # unity coefficient: exact
    y, _ = run_thumb(t, np.array([[0xFFFF, 0, 0, 0, 0]], dtype=np.int32), np.zeros((1, 2), np.int32), np.zeros(1, np.uint8), np.array([0xFFFF], dtype=np.int32))
    assert int(y[0]) == 0                                                    # al * bl is never formed


def test_executed_assembly_against_the_oracle():
    if not os.path.exists(ASM): pytest.skip("reference not present")
    import sys; sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_thumb_golden import cases, run_thumb, CBC
    t = Thumb(open(ASM).read(), symbols={"channel_band_counts": CBC})
    rng = np.random.default_rng(7)
    for k, (coef, state, bypass, x) in enumerate(cases(rng, 40)):
        y, st = run_thumb(t, coef, state, bypass, x, channel=k % 7)
        for name, lib in libs():
            yo, so = oracle_block(lib, coef, state, bypass, x)
            assert np.array_equal(y, yo) and np.array_equal(st, so), (name, k)
