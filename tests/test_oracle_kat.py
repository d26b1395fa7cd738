"""Known-answer tests derived from identities stated in the reference's own comments: the reference ships
no tests, so these analytical pins are the only "golden vectors" it offers besides its source."""
import math
import struct
import zlib

import numpy as np
import pytest

from orclib import Oracle
from dspi_amd import wire as W, workloads as WL


def f32(b, off=0):
    return struct.unpack_from("<f", b, off)[0]


def test_wire_layout_offsets():
    """bulk_params.h:42-205: 2896 bytes, section offsets as compiled on the host."""
    offs = {n: W.WIRE_BULK.fields[n][1] for n in W.WIRE_BULK.names}
    assert W.WIRE_BULK.itemsize == 2896
    assert offs == {"header": 0, "global_": 16, "crossfeed": 32, "legacy": 48, "delays": 64, "crosspoints": 108, "outputs": 252,
                    "pins": 360, "eq": 368, "channel_names": 2480, "i2s_config": 2832, "leveller": 2848, "preamp": 2864, "master_volume": 2880}
    assert W.preset_slot_dtype(1).itemsize == 2864 and W.preset_slot_dtype(0).itemsize == 1840


@pytest.mark.parametrize("flavor", [1, 0])
def test_preset_crc_is_zlib_crc32(flavor):
    """flash_storage.c:282-291 (poly 0xEDB88320, init/final 0xFFFFFFFF) == zlib.crc32 over bytes after the 12-byte header."""
    o = Oracle(flavor)
    img = o.save_slot(4)
    magic, version, slot, crc = struct.unpack_from("<IHHI", img)
    assert magic == 0x44535033 and version == 12 and slot == 4
    assert crc == (zlib.crc32(img[12:]) & 0xFFFFFFFF) == W.slot_crc(img)
    bad = bytearray(img); bad[100] ^= 1
    assert o.load_slot(bytes(bad)) == 3            # PRESET_ERR_CRC
    assert o.load_slot(img, expect_slot=5) == 3    # slot_index sanity check (validate_slot)
    assert o.load_slot(img, expect_slot=4) == 0


def test_crossfeed_identities():
    """crossfeed.c:65: 4.5 dB -> level_ratio 1.679, G = 0.373 ; x = exp(-2 pi fc/fs) ; mono input -> unity at DC (crossfeed.c:14)."""
    o = Oracle(1)
    o.set_rate(48000)
    o.vendor_set(W.REQ["SET_CROSSFEED"], 0, b"\x01")
    st = o.tap(2)
    lp_a0, lp_b1 = f32(st, 0), f32(st, 4)
    x = math.exp(-2 * math.pi * 700 / 48000)
    G = 1 / (1 + 10 ** (4.5 / 20))
    assert abs(10 ** (4.5 / 20) - 1.679) < 1e-3 and abs(G - 0.373) < 1e-3
    assert abs(lp_b1 - x) < 1e-6 and abs(lp_a0 - G * (1 - x)) < 1e-6
    # DC mono: out = (in - lp) + ap(lp) -> in
    blob = WL.config2_blob(); blob["eq"]["type"][:] = 0; blob["crossfeed"]["enabled"] = 1; blob["master_volume"]["master_volume_db"] = 0.0
    o.set_volume(-256)    # -1 dB: avoid the 0 dB sign quirk
    assert o.load_bulk(blob) == 0
    pcm = np.full((48 * 400, 2), 8000, dtype=np.int16)
    pairs, _, _, _ = o.process(pcm, 400, 48)
    expect = 8000 / 32768 * (0x7215 / 32768) * 8388607
    assert abs(pairs[0, -1, 0] - expect) / expect < 2e-4 and pairs[0, -1, 0] == pairs[0, -1, 1]


def test_leveller_coefficients():
    """leveller.c:37-40 alpha = exp(-ln10/(Fs*T)); :75-77 ratio = 1 + amount/100*19; presets {attack, release, rms}."""
    o = Oracle(1)
    o.set_rate(96000)
    o.vendor_set(W.REQ["SET_LEVELLER_AMOUNT"], 0, struct.pack("<f", 50.0))
    o.vendor_set(W.REQ["SET_LEVELLER_SPEED"], 0, b"\x02")
    c = struct.unpack("<9f", o.tap(3))
    a = lambda t: math.exp(-math.log(10) / (96000 * t))
    assert abs(c[0] - a(0.1)) < 1e-6 and abs(c[1] - a(0.02)) < 1e-6 and abs(c[2] - a(0.5)) < 1e-6
    assert c[3] == -20.0 and abs(c[4] - (1 + 0.5 * 19)) < 1e-6 and c[5] == 6.0 and c[6] == 0.0


def test_svf_biquad_crossover_and_flat_detection():
    """dsp_pipeline.c:88 SVF iff f < Fs/7.5; :6-17 flat when type FLAT, freq <= 0, or |gain| < 0.01 dB on peaking/shelves."""
    o = Oracle(1)
    o.set_rate(48000)
    def band(ch, b, t, f, q, g):
        return o.vendor_set(W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, b, t, 0, f, q, g))
    band(0, 0, W.FILTER_PEAKING, 6399.0, 1.0, 3.0)
    band(0, 1, W.FILTER_PEAKING, 6401.0, 1.0, 3.0)
    band(0, 2, W.FILTER_PEAKING, 1000.0, 1.0, 0.005)
    band(0, 3, W.FILTER_LOWPASS, 0.0, 1.0, 0.0)
    bq = np.frombuffer(o.tap(0), dtype=np.uint8).reshape(11, 12, 68)
    use_svf = bq[0, :, 64]; bypass = bq[0, :, 65]
    assert use_svf[0] == 1 and use_svf[1] == 0 and bypass[0] == 0 and bypass[1] == 0
    assert bypass[2] == 1 and bypass[3] == 1


@pytest.mark.parametrize("flavor", [1, 0])
def test_factory_defaults(flavor):
    """dsp_pipeline.c:201-213: HPF 80 Hz on S/PDIF outputs, LPF 80 Hz on the sub; config.h:93-95 + dsp_pipeline.c:226-230:
    sub aligned by +128 samples; flash_storage.c:1183-1198: stereo pass-through on pair 1 only; master volume -20 dB."""
    o = Oracle(flavor)
    C, N = o.C, o.N
    rec = np.frombuffer(o.tap(7), dtype=W.EQ_PARAM_PACKET).reshape(C, 12)
    for ch in range(2, C - 1):
        assert rec[ch, 0]["type"] == W.FILTER_HIGHPASS and rec[ch, 0]["freq"] == 80.0
    assert rec[C - 1, 0]["type"] == W.FILTER_LOWPASS and rec[C - 1, 0]["freq"] == 80.0
    o.set_rate(48000)
    dly = np.frombuffer(o.tap(5), dtype=np.int32)
    assert dly.tolist() == [0] * (N - 1) + [128]
    assert o.scalar(1) == 0 and abs(o.scalar_f(4) + 20.0) < 1e-6          # core1 idle, master -20 dB
    assert o.vendor_get(W.REQ["GET_OUTPUT_ENABLE"], 0) == b"\x01" and o.vendor_get(W.REQ["GET_OUTPUT_ENABLE"], 2) == b"\x00"


def test_host_volume_sign_quirk():
    """usb_audio.c:410-434: the 0 dB table entry 0x8000 lands in an int16_t -> vol_mul = -32768: polarity flips at 0 dB only."""
    o = Oracle(1)
    o.set_volume(0)
    assert o.scalar(6) == -32768
    o.set_volume(-256)
    assert o.scalar(6) == 0x7215


def test_bypass_is_bit_identical_passthrough():
    """all stages off, unity gains, -1 dB host volume: out = trunc(in/32768 * vol * 8388607) exactly (float flavour)."""
    o = Oracle(1)
    o.set_rate(48000); o.set_volume(-256)
    blob = WL.config2_blob(); blob["eq"]["type"][:] = 0
    assert o.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(1, 48 * 20, 48000, first_stream=1, mix=False)[0]
    pairs, _, _, _ = o.process(pcm, 20, 48)
    vol = np.float32(0x7215) * np.float32(1 / 32768)
    n = 48 * 6     # past the 256-sample preset mute that load_bulk engages (main.c:1130) and its 8 ms ramp
    exp = (pcm.astype(np.float32) * np.float32(1 / 32768)) * vol
    exp = np.trunc(np.clip(exp, -1, 1) * np.float32(8388607.0)).astype(np.int32)
    assert np.array_equal(pairs[0, -n:], exp[-n:])
