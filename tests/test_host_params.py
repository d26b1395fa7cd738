"""The product's host-side parameter model (dspi_amd/csrc/dspi_params.cpp, through the C-ABI with a host-only context)
against the oracle: identical blobs in -> identical derived coefficient images and identical blobs out.  CPU only."""
import struct

import numpy as np
import pytest

from orclib import Oracle
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi, DspiError, E_NODEVICE

# DevImage layout (dspi_amd/csrc/dspi_image.h)
BAND = np.dtype([("c", "<u4", 6), ("kind", "<u4"), ("pad", "<u4")])
IMAGE = np.dtype([("eq", BAND, (11, 10)), ("loud", BAND, 2), ("flags", "<u4"), ("ch_bypassed", "<u4"), ("out_enabled", "<u4"), ("out_mute", "<u4"),
                  ("fs_hz", "<u4"), ("mute_transition", "<u4"), ("preamp", "<u4", 2), ("vol", "<u4"), ("master", "<u4"), ("mix", "<u4", (2, 9)),
                  ("out_gain_lin", "<f4", 9), ("delay_samples", "<i4", 9), ("lv", "<f4", 9), ("lv_alpha_rms_q28", "<i4"), ("xf", "<u4", 3), ("pad", "<u4", 5)])
assert IMAGE.itemsize == 3840


def oracle_image_view(o: Oracle, flavor: int):
    """Rebuild the expected per-band (kind, coefficients) from the oracle's Biquad array."""
    C = o.C
    out = []
    if flavor:
        bq = np.frombuffer(o.tap(0), dtype=np.dtype([("b", "<u4", 5), ("s", "<u4", 2), ("sva", "<u4", 3), ("svm", "<u4", 3), ("svic", "<u4", 2),
                                                     ("svf_type", "<u4"), ("use_svf", "u1"), ("bypass", "u1"), ("pad", "u1", 2)])).reshape(C, 12)
        for ch in range(C):
            for b in range(10):
                q = bq[ch, b]
                if q["bypass"]:
                    out.append((0, [0] * 6)); continue
                if q["use_svf"]:
                    t = int(q["svf_type"])
                    if t == W.FILTER_LOWPASS: out.append((2, list(q["sva"]) + [0, 0, 0]))
                    elif t == W.FILTER_HIGHPASS: out.append((3, list(q["sva"]) + [q["svm"][1], 0, 0]))
                    elif t == W.FILTER_PEAKING: out.append((4, list(q["sva"]) + [q["svm"][1], 0, 0]))
                    else: out.append((5, list(q["sva"]) + list(q["svm"])))
                else:
                    out.append((1, list(q["b"]) + [0]))
    else:
        bq = np.frombuffer(o.tap(0), dtype=np.dtype([("b", "<u4", 5), ("s", "<u4", 2), ("bypass", "u1"), ("pad", "u1", 3)])).reshape(C, 12)
        for ch in range(C):
            for b in range(10):
                q = bq[ch, b]
                out.append((0, [0] * 6) if q["bypass"] else (1, list(q["b"]) + [0]))
    return out


def check_against_oracle(d: Dspi, o: Oracle, flavor: int):
    img = np.frombuffer(d.debug_image(), dtype=IMAGE)[0]
    exp = oracle_image_view(o, flavor)
    k = 0
    for ch in range(o.C):
        for b in range(10):
            kind, c = exp[k]; k += 1
            assert int(img["eq"][ch, b]["kind"]) == kind, (ch, b)
            if kind:
                assert [int(v) for v in img["eq"][ch, b]["c"]] == [int(v) for v in c], (ch, b, kind)
    assert img["delay_samples"][: o.N].tolist() == np.frombuffer(o.tap(5), dtype=np.int32).tolist()
    lv = np.frombuffer(o.tap(3), dtype=np.float32)       # alpha_rms, attack, release, threshold, ratio, knee, makeup, gate, max_gain
    assert img["lv"].view(np.uint32).tolist() == lv.view(np.uint32).tolist()
    xs = np.frombuffer(o.tap(2), dtype=np.uint32)        # lp_a0 lp_b1 stL stR ap_a ...
    assert [int(img["xf"][0]), int(img["xf"][1]), int(img["xf"][2])] == [int(xs[0]), int(xs[1]), int(xs[4])]
    row = o.scalar(0)
    loud = np.frombuffer(o.tap(1), dtype=np.uint8)
    if row >= 0 and o.vendor_get(W.REQ["GET_LOUDNESS"], 0) == b"\x01":       # the selected row of the loudness table (loudness.h:10-23)
        rec = 28 if flavor else 24
        tab = loud.reshape(61, 2, rec)
        for j in range(2):
            words = tab[row, j, :rec - 4].view(np.uint32).tolist()
            if tab[row, j, rec - 4]:
                assert int(img["loud"][j]["kind"]) == 0
            else:
                assert int(img["loud"][j]["kind"]) == (5 if flavor else 1)
                assert [int(v) for v in img["loud"][j]["c"]][:len(words)] == words, ("loudness stage", j)
    assert int(img["fs_hz"]) == o.scalar(10)
    assert bool(img["flags"] & 8) == (not o.scalar(4)) and bool(img["flags"] & 2) == (not o.scalar(5))     # crossfeed / leveller on
    assert bool(img["flags"] & 16) == (o.scalar(1) != 2) and bool(img["flags"] & 32) == bool(o.scalar(2))
    if flavor:
        assert img["preamp"].view(np.float32).tolist() == [o.scalar_f(1), o.scalar_f(2)]
        assert float(img["master"].view(np.float32)) == o.scalar_f(0)
    else:
        assert img["preamp"].view(np.int32).tolist() == [o.scalar(8), o.scalar(9)]
        assert int(img["master"].view(np.int32)) == o.scalar(7)
    assert d.collect_bulk() == o.collect_bulk()
    assert d.save_slot(2) == o.save_slot(2)
    return img, row, loud


# (flavour, float contract): canonical float, float with the firmware build's FMA contraction, Q28
FL = [(1, False), (1, True), (0, False)]


@pytest.mark.parametrize("flavor,fma", FL)
def test_boot_state_matches(product_lib, flavor, fma):
    d, o = Dspi(flavor, 3, device=None, fma=fma), Oracle(flavor, fma=fma)
    check_against_oracle(d, o, flavor)


@pytest.mark.parametrize("flavor,fma", FL)
@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_bulk_load_images(product_lib, flavor, fma, fs):
    d, o = Dspi(flavor, 2, device=None, fma=fma), Oracle(flavor, fma=fma)
    assert d.set_rate(fs) == 0 and o.set_rate(fs) == 0
    d.set_volume(-20 * 256); o.set_volume(-20 * 256)
    rng = np.random.default_rng(fs + flavor)
    for trial in range(5):
        blob = WL.full_chain_blob(flavor)
        if trial:
            blob["eq"]["gain_db"] = rng.uniform(-15, 15, size=blob["eq"]["gain_db"].shape).astype(np.float32)
            blob["eq"]["freq"] = (10 ** rng.uniform(0.8, 4.6, size=blob["eq"]["freq"].shape)).astype(np.float32)
            blob["eq"]["q"] = rng.uniform(0.05, 25, size=blob["eq"]["q"].shape).astype(np.float32)
            blob["eq"]["type"] = rng.integers(0, 6, size=blob["eq"]["type"].shape)
            blob["outputs"]["gain_db"] = rng.uniform(-70, 25, size=9).astype(np.float32)
            blob["outputs"]["delay_ms"] = rng.uniform(0, 100, size=9).astype(np.float32)
            blob["preamp"]["preamp_db"] = rng.uniform(-30, 25, size=2).astype(np.float32)
            blob["leveller"]["amount"] = rng.uniform(-10, 120)
            blob["global_"]["loudness_ref_spl"] = rng.uniform(60, 100); blob["global_"]["loudness_intensity_pct"] = rng.uniform(20, 180)
            blob["crossfeed"]["custom_fc"] = rng.uniform(500, 2000); blob["crossfeed"]["custom_feed_db"] = rng.uniform(1, 14)
            blob["crossfeed"]["preset"] = trial % 4
            blob["header"]["format_version"] = [6, 6, 5, 4, 2][trial]
        assert d.load_bulk(blob) == 0 and o.load_bulk(blob) == 0
        check_against_oracle(d, o, flavor)


@pytest.mark.parametrize("flavor", [1, 0])
def test_bulk_error_codes(product_lib, flavor):
    d = Dspi(flavor, 1, device=None)
    blob = WL.full_chain_blob(flavor)
    for field, val, code in (("format_version", 7, -1), ("format_version", 1, -1), ("platform_id", 1 - flavor, -2), ("num_channels", 5, -3),
                             ("num_output_channels", 2, -3), ("payload_length", 64, -4), ("payload_length", 3000, -4)):
        b = blob.copy(); b["header"][field] = val
        assert d.load_bulk(b) == code
    assert d.load_bulk(blob.tobytes()[:-1]) == -4


@pytest.mark.parametrize("flavor,fma", FL)
def test_vendor_requests_match_oracle(product_lib, flavor, fma):
    d, o = Dspi(flavor, 2, device=None, fma=fma), Oracle(flavor, fma=fma)
    d.set_rate(48000); o.set_rate(48000)
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    script = [
        (R["SET_PREAMP"], 0, f(-4.5)), (R["SET_PREAMP_CH"], 1, f(2.25)), (R["SET_MASTER_VOLUME"], 0, f(-7.0)), (R["SET_MASTER_VOLUME"], 0, f(-200.0)),
        (R["SET_MASTER_VOLUME"], 0, f(float("nan"))), (R["SET_BYPASS"], 0, b"\x01"), (R["SET_BYPASS"], 0, b"\x00"),
        (R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 3, W.FILTER_PEAKING, 0, 900.0, 30.0, 5.0)),      # Q clamps on the design copy only
        (R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 4, 9, W.FILTER_HIGHSHELF, 0, 30000.0, 0.7, -4.0)),
        (R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 1, 11, W.FILTER_PEAKING, 0, 900.0, 1.0, 5.0)),       # band >= 10: ignored
        (R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 3, W.FILTER_PEAKING, 0, 9000.0, 1.0, 5.0)),       # SVF -> biquad path change
        (R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 3, W.FILTER_FLAT, 0, 9000.0, 1.0, 5.0)),
        (R["SET_LOUDNESS_REF"], 0, f(120.0)), (R["SET_LOUDNESS_INTENSITY"], 0, f(150.0)), (R["SET_LOUDNESS"], 0, b"\x01"),
        (R["SET_CROSSFEED_PRESET"], 0, b"\x03"), (R["SET_CROSSFEED_FREQ"], 0, f(5000.0)), (R["SET_CROSSFEED_FEED"], 0, f(-3.0)),
        (R["SET_CROSSFEED"], 0, b"\x01"), (R["SET_CROSSFEED_ITD"], 0, b"\x00"), (R["SET_CROSSFEED_PRESET"], 0, b"\x09"),
        (R["SET_LEVELLER_ENABLE"], 0, b"\x01"), (R["SET_LEVELLER_AMOUNT"], 0, f(80.0)), (R["SET_LEVELLER_SPEED"], 0, b"\x02"), (R["SET_LEVELLER_SPEED"], 0, b"\x07"),
        (R["SET_LEVELLER_MAX_GAIN"], 0, f(50.0)), (R["SET_LEVELLER_LOOKAHEAD"], 0, b"\x00"), (R["SET_LEVELLER_GATE"], 0, f(-120.0)),
        (R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", 1, 2, 1, 1, -3.0)), (R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", 2, 0, 1, 0, 0.0)),
        (R["SET_OUTPUT_ENABLE"], 2, b"\x01"), (R["SET_OUTPUT_ENABLE"], (9 if flavor else 5) - 1, b"\x01"),     # PDM refused while a Core-1 EQ output is on
        (R["SET_OUTPUT_GAIN"], 2, f(-6.0)), (R["SET_OUTPUT_MUTE"], 1, b"\x01"), (R["SET_OUTPUT_DELAY"], 2, f(12.5)), (R["SET_OUTPUT_DELAY"], 1, f(-3.0)),
        (R["SET_DELAY"], 3, f(7.0)), (R["SET_CHANNEL_GAIN"], 1, f(-2.0)), (R["SET_CHANNEL_MUTE"], 2, b"\x01"),
        (R["SET_CHANNEL_NAME"], 1, b"Left woofer"), (R["SET_MASTER_VOLUME_MODE"], 0, b"\x01"), (R["SET_PREAMP"], 0, b"\x00\x00"),   # short payload: ignored
    ]
    for req, wv, payload in script:
        assert d.vendor_set(req, wv, payload) == 0 and o.vendor_set(req, wv, payload) == 0, hex(req)
        d.set_volume(-30 * 256); o.set_volume(-30 * 256)
        check_against_oracle(d, o, flavor)
    for req in range(0x40, 0xE0):
        for wv in (0, 1, 2, 0x0103, 0x0231, 9, 15):
            a, b = d.vendor_get(req, wv, 64), o.vendor_get(req, wv, 64)
            if req in (0xA0, 0x53, 0x50, 0x83, 0xD6):
                continue
            assert a == b, (hex(req), wv)
    for wv in (0x0101, 0x0101, 0x0200, 0x0107, 0x0100, 0x0001):      # REQ_SET_OUTPUT_TYPE answers on the IN side (usb_audio.c:2984-3016)
        assert d.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1) == o.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1), hex(wv)
        assert [d.vendor_get(R["GET_OUTPUT_TYPE"], i, 1) for i in range(5)] == [o.vendor_get(R["GET_OUTPUT_TYPE"], i, 1) for i in range(5)]
        check_against_oracle(d, o, flavor)
    assert d.vendor_set(0xC0, 0, b"\x01") == -14 and o.vendor_set(0xC0, 0, b"\x01") == -1      # the OUT direction of 0xC0 does not exist; pin / flash requests: unsupported
    assert d.vendor_get(W.REQ["GET_ALL_PARAMS"], 0, 4096) == o.collect_bulk()


@pytest.mark.parametrize("flavor,fma", FL)
def test_preset_slot_roundtrip_and_legacy_versions(product_lib, flavor, fma):
    d, o = Dspi(flavor, 1, device=None, fma=fma), Oracle(flavor, fma=fma)
    blob = WL.full_chain_blob(flavor)
    assert d.load_bulk(blob) == 0 and o.load_bulk(blob) == 0
    img = d.save_slot(7)
    assert img == o.save_slot(7)
    slot = np.frombuffer(img, dtype=W.preset_slot_dtype(flavor)).copy()
    for version in (12, 11, 10, 9, 8, 7, 5):
        s = slot.copy(); s["version"] = version; s["preamp_db"] = -9.0; s["master_volume_db"] = -11.0
        raw = W.seal_slot(s[0])
        for mode in (0, 1):
            d.vendor_set(W.REQ["SET_MASTER_VOLUME_MODE"], 0, bytes([mode])); o.vendor_set(W.REQ["SET_MASTER_VOLUME_MODE"], 0, bytes([mode]))
            d.factory_defaults(); o.factory_defaults()
            assert d.load_slot(raw, 7) == 0 and o.load_slot(raw, 7) == 0
            check_against_oracle(d, o, flavor)
    bad = bytearray(img); bad[500] ^= 0x10
    assert d.load_slot(bytes(bad)) == 3 and d.load_slot(img, expect_slot=1) == 3 and d.load_slot(img[:100]) == 3


def test_copy_on_write_images(product_lib):
    d = Dspi(1, 130, device=None)
    base = d.collect_bulk(0)
    d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0), stream=77)
    assert d.collect_bulk(77) != base and d.collect_bulk(76) == base and d.collect_bulk(129) == base
    d.vendor_set(W.REQ["SET_BYPASS"], 0, b"\x01")        # ALL: reaches both images
    assert d.vendor_get(W.REQ["GET_BYPASS"], 0, stream=77) == b"\x01" and d.vendor_get(W.REQ["GET_BYPASS"], 0, stream=3) == b"\x01"
    assert struct.unpack("<f", d.vendor_get(W.REQ["GET_PREAMP"], 0, stream=77))[0] == -6.0
    assert struct.unpack("<f", d.vendor_get(W.REQ["GET_PREAMP"], 0, stream=3))[0] == 0.0


@pytest.mark.parametrize("flavor", [1, 0])
def test_equal_images_fold_back_after_broadcast_calls(product_lib, flavor):
    """Streams that were separated by per-stream calls and then receive the same whole state again (broadcast factory reset + blob,
    or a broadcast request that removes the only difference) share one parameter object again; streams whose parameters — or
    whose pending state operations — still differ stay apart, and every stream still answers with its own state."""
    S = 300
    d = Dspi(flavor, S, device=None)
    blob = WL.full_chain_blob(flavor)
    d.set_rate(48000); d.set_volume(-9 * 256); assert d.load_bulk(blob) == 0
    assert d.image_count() == 1
    R = W.REQ
    rng = np.random.default_rng(4)
    for s_ in range(S):
        d.vendor_set(R["SET_PREAMP"], 0, struct.pack("<f", -15.0 + 0.1 * s_), stream=s_)
        if s_ % 3 == 0: d.vendor_set(R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", int(rng.integers(0, 7)), int(rng.integers(0, 10)), W.FILTER_PEAKING, 0, 900.0, 1.1, 3.0), stream=s_)
        if s_ % 5 == 1: d.vendor_set(R["SET_LEVELLER_ENABLE"], 0, b"\x00", stream=s_)
        if s_ % 7 == 2: d.set_mute(True, stream=s_)
        if s_ % 11 == 3: d.set_volume(-256 * s_ % 40, stream=s_)
    assert d.image_count() == S
    d.vendor_set(R["SET_PREAMP"], 0, struct.pack("<f", -2.0))                 # ALL: removes one difference; streams 4, 8, 10 ... differ in nothing else
    plain = [s_ for s_ in range(S) if s_ % 3 and s_ % 5 != 1 and s_ % 7 != 2 and s_ % 11 != 3]
    distinct = {(d.collect_bulk(s_), s_ % 7 == 2, (-256 * s_ % 40) if s_ % 11 == 3 else None) for s_ in range(S)}
    assert d.image_count() == len(distinct) and 1 < len(distinct) < S - len(plain) + 2
    assert len({d.collect_bulk(s_) for s_ in plain}) == 1 and d.collect_bulk(plain[0]) != d.collect_bulk(0)
    d.set_mute(False); d.set_volume(-9 * 256)
    d.factory_defaults(); assert d.load_bulk(blob) == 0
    assert d.image_count() == 1
    ref = Dspi(flavor, 1, device=None); ref.set_rate(48000); ref.set_volume(-9 * 256); ref.load_bulk(blob); ref.factory_defaults(); ref.load_bulk(blob)
    assert all(d.collect_bulk(s_) == ref.collect_bulk(0) for s_ in (0, 1, 77, S - 1))
    d.vendor_set(R["SET_PREAMP"], 0, struct.pack("<f", -6.0), stream=77)       # and apart again
    assert d.image_count() == 2 and d.collect_bulk(77) != d.collect_bulk(76)


def test_process_without_device_fails_loudly(product_lib):
    d = Dspi(1, 4, device=None)
    with pytest.raises(DspiError) as e:
        d.process_host(np.zeros((4, 48, 2), np.int16), 1, 48)
    assert e.value.code == E_NODEVICE
