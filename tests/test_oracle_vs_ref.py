"""Standalone oracle == oracle/_ref (reference leaf sources compiled in place) bit-for-bit, across flavours, rates,
packet sizes, bit depths, parameter paths and both libm modes.  Runs where /root/reference (or a prebuilt
oracle/_ref/*.so) exists; skipped elsewhere — tests/test_oracle_golden.py carries the pin to other machines."""
import numpy as np
import pytest

import orclib
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL

pytestmark = pytest.mark.skipif(not (orclib.ref_available(1) and orclib.ref_available(0)), reason="oracle/_ref not built (needs /root/reference)")

CASES = [
    (1, 48000, 48, 16, False), (1, 96000, 96, 16, True), (1, 44100, 45, 24, True), (1, 44100, 44, 16, False),
    (0, 48000, 48, 16, False), (0, 96000, 96, 24, True), (0, 44100, 45, 16, True),
]


def pair(flavor, detmath):
    return Oracle(flavor, ref=False, detmath=detmath, x86_casts=True), Oracle(flavor, ref=True, detmath=detmath, x86_casts=True)


@pytest.mark.parametrize("flavor,fs,B,depth,detmath", CASES)
def test_full_chain_bit_exact(flavor, fs, B, depth, detmath):
    a, b = pair(flavor, detmath)
    blob = WL.full_chain_blob(flavor)
    for o in (a, b):
        assert o.set_rate(fs) == 0
        o.set_volume(-12 * 256)
        assert o.load_bulk(blob) == 0
    for first in (2, 15, 16, 18, 19):       # noise, sweep, bursts, silence tail, full-scale square
        pcm = WL.synth_pcm16(1, B * 25, fs, first_stream=first)[0]
        data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm[None])[0]
        ra, rb = a.process(data, 25, B, depth), b.process(data, 25, B, depth)
        for x, y in zip(ra[:3], rb[:3]):
            assert np.array_equal(x, y)
        assert ra[3] == rb[3] and a.status() == b.status()
    for t in range(9):
        assert a.tap(t) == b.tap(t), f"state tap {t}"


@pytest.mark.parametrize("flavor", [1, 0])
def test_control_surface_matches(flavor):
    """bulk apply/collect run the reference's own bulk_params.c in the _ref build."""
    a, b = pair(flavor, False)
    rng = np.random.default_rng(5)
    blob = WL.full_chain_blob(flavor)
    for trial in range(6):
        bl = blob.copy()
        bl["eq"]["gain_db"] = rng.uniform(-12, 12, size=bl["eq"]["gain_db"].shape).astype(np.float32)
        bl["eq"]["freq"] = rng.uniform(15, 30000, size=bl["eq"]["freq"].shape).astype(np.float32)
        bl["eq"]["q"] = rng.uniform(0.05, 25, size=bl["eq"]["q"].shape).astype(np.float32)
        bl["eq"]["type"] = rng.integers(0, 6, size=bl["eq"]["type"].shape)
        bl["outputs"]["gain_db"] = rng.uniform(-70, 25, size=9).astype(np.float32)     # exercises the Taylor dB clamp
        bl["crosspoints"]["gain_db"] = rng.uniform(-30, 12, size=(2, 9)).astype(np.float32)
        bl["header"]["format_version"] = [6, 5, 4, 3, 2, 6][trial]
        for o in (a, b):
            assert o.load_bulk(bl) == 0
        assert a.collect_bulk() == b.collect_bulk()
        for t in (0, 1, 2, 3, 4, 5, 7):
            assert a.tap(t) == b.tap(t), f"tap {t} trial {trial}"
    # error codes of bulk_params_apply (bulk_params.c:181-203)
    for field, val, code in (("format_version", 7, -1), ("format_version", 1, -1), ("platform_id", 1 - flavor, -2), ("num_channels", 3, -3), ("payload_length", 100, -4)):
        bl = blob.copy(); bl["header"][field] = val
        assert a.load_bulk(bl) == code and b.load_bulk(bl) == code


@pytest.mark.skipif(not orclib.ref_available(1, "ref", True), reason="oracle/_ref/libref_f32_fma.so not built")
@pytest.mark.parametrize("fs,B,depth", [(48000, 48, 16), (96000, 96, 24), (44100, 45, 16)])
def test_firmware_float_contract(fs, B, depth):
    """The float flavour as the firmware is built — GCC's FMA contraction (oracle/Makefile FMA_FLAGS): the standalone oracle's
    explicit fused pattern (orc_leaf.c MAD) against the reference leaf sources compiled with contraction on; coefficients
    of every design function and every output word."""
    a = Oracle(1, ref=False, detmath=True, x86_casts=True, fma=True); b = Oracle(1, ref=True, detmath=True, x86_casts=True, fma=True)
    canon = Oracle(1, ref=False, detmath=True, x86_casts=True)
    blob = WL.full_chain_blob(1)
    for o in (a, b, canon):
        assert o.set_rate(fs) == 0
        o.set_volume(-12 * 256)
        assert o.load_bulk(blob) == 0
    for t in range(9):
        assert a.tap(t) == b.tap(t), f"state tap {t}"
    differs = 0
    for first in (2, 15, 16, 18, 19):
        pcm = WL.synth_pcm16(1, B * 25, fs, first_stream=first)[0]
        data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm[None])[0]
        ra, rb, rc = a.process(data, 25, B, depth), b.process(data, 25, B, depth), canon.process(data, 25, B, depth)
        for x, y in zip(ra[:3], rb[:3]):
            assert np.array_equal(x, y)
        assert a.status() == b.status()
        differs += int((ra[0] != rc[0]).sum())
    assert differs > 0, "the contracted build must not be the canonical one"
