"""S/PDIF subframe oracle (oracle/orc_spdif.c): pinned against the reference's own spdif_update_subframe (header compiled
in place, oracle/_ref/libref_spdif.so) and against IEC 60958 structure (preambles, parity, channel status)."""
import numpy as np
import pytest

import orclib

X, Y, Z = 0xC9, 0x69, 0x39


def test_restatement_matches_the_reference_header():
    if not orclib.spdif_ref_available():
        pytest.skip("oracle/_ref/libref_spdif.so not built (needs /root/reference)")
    rng = np.random.default_rng(3)
    for fs in (44100, 48000, 96000, 32000):
        x = rng.integers(-(1 << 31), 1 << 31, size=(1000, 2), dtype=np.int64).astype(np.int32)
        x[:8] = [[0, 0], [0x7FFFFF, -0x800000], [-1, 1], [0x555555, 0x2AAAAA], [0x800000, 0x7FFFFF], [1, 2], [3, 4], [0xFF, 0xFF00]]
        for pos in (0, 7, 191):
            a, na = orclib.spdif_encode(x, pos, fs)
            b, nb = orclib.spdif_encode(x, pos, fs, ref=True)
            assert np.array_equal(a, b) and na == nb == (pos + 1000) % 192


def decode(l, h):
    """Undo the cell-pair coding: every time slot is two cells, the second cell carries the data bit."""
    bits64 = (int(h) << 32) | int(l)
    return [(bits64 >> (2 * k + 1)) & 1 for k in range(4, 32)]       # slots 4..31 (slots 0-3 = preamble)


def test_structure_preambles_parity_channel_status():
    fs = 48000
    x = np.random.default_rng(9).integers(-(1 << 23), 1 << 23, size=(400, 2)).astype(np.int32)
    out, nxt = orclib.spdif_encode(x, 0, fs)
    assert nxt == 400 % 192
    status = [0x04, 0x00, 0x00, 0x02, 0x0B]        # consumer, PCM; 48 kHz; 24-bit word length (audio_spdif.c:83-89, :253)
    for i in range(400):
        pos = i % 192
        assert (out[i, 0] & 0xFF) == (Z if pos == 0 else X) and (out[i, 2] & 0xFF) == Y
        for side in range(2):
            bits = decode(out[i, 2 * side], out[i, 2 * side + 1])
            sample = int(x[i, side]) & 0xFFFFFF
            assert bits[:24] == [(sample >> k) & 1 for k in range(24)]          # 24 audio bits, LSB first, slots 4-27
            v, u, c, p = bits[24:28]
            assert v == 0 and u == 0
            assert c == ((status[pos // 8] >> (pos % 8)) & 1 if pos < 40 else 0)
            assert (sum(bits[:27]) + p) % 2 == 0                                # even parity over slots 4-31
        # first cell of every data slot is 1 (the table's 0x5555 pattern)
        assert (int(out[i, 0]) >> 8) & 0x555555 == 0x555555 and int(out[i, 1]) & 0x55555555 == 0x55555555


def test_i2s_restatement_matches_the_reference_producer_give():
    """orc_i2s_frames against the reference's own i2s_wrap_producer_give (audio_i2s_multi.c:198-243, compiled in place): packets of
    every length the firmware sees, consumer buffers that do and do not divide them; whole consumer buffers come out."""
    if not orclib.i2s_ref_available():
        pytest.skip("oracle/_ref/libref_i2s.so not built (needs /root/reference)")
    rng = np.random.default_rng(11)
    x = rng.integers(-(1 << 23), 1 << 23, size=(2000, 2), dtype=np.int64).astype(np.int32)
    x[:6] = [[0, 0], [0x7FFFFF, -0x800000], [-1, 1], [0x555555, -0x2AAAAB], [1, 2], [0x123456, -0x123456]]
    want = orclib.i2s_frames(x)
    assert np.array_equal(want[:6, 0], np.array([0, 0x7FFFFF00, 0xFFFFFF00, 0x55555500, 0x100, 0x12345600], dtype=np.uint32))
    assert not (want & 0xFF).any()
    for packet, consumer in ((48, 48), (96, 48), (45, 48), (44, 192), (1, 7), (97, 64)):
        got = orclib.i2s_ref_give(x, packet, consumer)
        assert len(got) == (2000 // consumer) * consumer
        assert np.array_equal(got, want[:len(got)])
