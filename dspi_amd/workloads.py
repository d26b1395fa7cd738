"""Workload definitions shared by tests and bench.py: the BASELINE.json configurations as
DSPi bulk blobs, and the synthetic PCM generator of SURVEY.md §8(d).

Everything here is data preparation (blob packing, PRNG); no chain arithmetic.
"""
from __future__ import annotations

import numpy as np

from . import wire as W

PEQ_FREQS = [32.0, 64.0, 125.0, 250.0, 500.0, 1000.0, 2000.0, 4000.0, 8000.0, 16000.0]
PEQ_GAINS = [3.0, -2.0, 4.0, -3.0, 2.0, -4.0, 3.0, -2.0, 4.0, -3.0]


def _master_peq(b, freqs=PEQ_FREQS, gains=PEQ_GAINS):
    for ch in (0, 1):
        for k in range(10):
            W.set_band(b, ch, k, W.FILTER_PEAKING, freqs[k], 1.41, gains[k])


def _passthrough_pair0(b):
    b["crosspoints"][0, 0]["enabled"] = 1
    b["crosspoints"][1, 1]["enabled"] = 1
    b["outputs"][0]["enabled"] = 1
    b["outputs"][1]["enabled"] = 1


def config1_blob() -> np.ndarray:
    """BASELINE config 1: Q28, master 10-band PEQ only, outputs 0-1 pass-through, everything else off."""
    b = W.new_bulk(W.FLAVOR_Q28)
    _master_peq(b)
    _passthrough_pair0(b)
    b["master_volume"]["master_volume_db"] = 0.0
    return b


def config2_blob(all_biquad: bool = False) -> np.ndarray:
    """BASELINE config 2: f32, 48 kHz, master L/R 10-band PEQ; at 48 kHz bands below 6.4 kHz take the
    SVF path and the rest the biquad path.  all_biquad spreads the bands over 6.4-20 kHz instead."""
    b = W.new_bulk(W.FLAVOR_F32)
    if all_biquad:
        freqs = [6500.0, 7200.0, 8000.0, 9000.0, 10000.0, 11500.0, 13000.0, 15000.0, 17500.0, 20000.0]
        _master_peq(b, freqs=freqs)
    else:
        _master_peq(b)
    _passthrough_pair0(b)
    b["master_volume"]["master_volume_db"] = 0.0
    return b


_OUT_TYPES = [W.FILTER_PEAKING, W.FILTER_LOWSHELF, W.FILTER_HIGHSHELF, W.FILTER_PEAKING, W.FILTER_LOWPASS,
              W.FILTER_PEAKING, W.FILTER_HIGHPASS, W.FILTER_PEAKING, W.FILTER_HIGHSHELF]
_OUT_FREQS = [120.0, 300.0, 9000.0, 1500.0, 18000.0, 3500.0, 25.0, 7000.0, 12000.0]


def full_chain_blob(flavor: int, max_delay_ms: float | None = None) -> np.ndarray:
    """BASELINE config 3 (f32, 9 outputs) / config 5 (Q28, 5 outputs): every stage active.

    preamp -3/-3 dB, loudness on (ref 83, 100 %), master 10-band PEQ, leveller on (amount 50, slow,
    max 15 dB, lookahead on, gate -96), crossfeed preset 0 with ITD, all outputs enabled with L+R
    crosspoints (one phase-inverted), 10 bands per output (band 0 = the 80 Hz HPF/LPF defaults, the
    rest mixed types incl. shelves), output gains -1..-N dB, delays {0,.5,1,2,5,10,20,40,80} ms.
    """
    C, N, _, _, _ = W.dims(flavor)
    b = W.new_bulk(flavor)
    b["preamp"]["preamp_db"][:] = -3.0
    b["global_"]["preamp_gain_db"] = -3.0
    b["global_"]["loudness_enabled"] = 1
    _master_peq(b)
    lv = b["leveller"]
    lv["enabled"], lv["amount"], lv["speed"], lv["max_gain_db"], lv["lookahead"], lv["gate_threshold_db"] = 1, 50.0, 0, 15.0, 1, -96.0
    b["crossfeed"]["enabled"], b["crossfeed"]["preset"], b["crossfeed"]["itd_enabled"] = 1, 0, 1
    delays = [0.0, 0.5, 1.0, 2.0, 5.0, 10.0, 20.0, 40.0, 80.0]
    if flavor == W.FLAVOR_Q28:
        delays = [0.0, 0.5, 2.0, 10.0, 40.0]
    for o in range(N):
        for inp in (0, 1):
            xp = b["crosspoints"][inp, o]
            xp["enabled"] = 1
            xp["gain_db"] = -6.0 if inp == (o & 1) else -9.0
        b["crosspoints"][1, 2]["phase_invert"] = 1
        out = b["outputs"][o]
        out["enabled"] = 1
        out["gain_db"] = -(o + 1.0)
        d = delays[o]
        if max_delay_ms is not None:
            d = min(d, max_delay_ms)
        out["delay_ms"] = d
        ch = 2 + o
        is_sub = (o == N - 1)
        W.set_band(b, ch, 0, W.FILTER_LOWPASS if is_sub else W.FILTER_HIGHPASS, 80.0, 0.707, 0.0)
        for k in range(1, 10):
            j = (k + 2 * o) % len(_OUT_TYPES)
            t = _OUT_TYPES[j]
            f = _OUT_FREQS[j] * (1.0 + 0.03 * o)
            g = [2.5, -3.5, 1.5, -2.0, 3.0][(k + o) % 5]
            W.set_band(b, ch, k, t, f, 0.9 + 0.1 * ((k + o) % 4), g)
    b["master_volume"]["master_volume_db"] = 0.0
    return b


# ------------------------------------------------------------------------------------------------
# synthetic PCM
# ------------------------------------------------------------------------------------------------
def _xorshift32(x: np.ndarray) -> np.ndarray:
    x ^= (x << np.uint32(13))
    x ^= (x >> np.uint32(17))
    x ^= (x << np.uint32(5))
    return x


def synth_pcm16(n_streams: int, n_frames: int, fs: int = 48000, first_stream: int = 0, mix: bool = True) -> np.ndarray:
    """int16 array [stream][frame][2].  Per stream s the PRNG is xorshift32 seeded
    0x9E3779B9 ^ (s*2654435761); stream classes by (s % 20): 0-13 white noise +-16384 (-6 dBFS),
    14-15 log sine sweep 20 Hz->20 kHz at -12 dBFS with L/R 90 degrees apart, 16-17 speech-like bursts
    (1 s at -30 dBFS / 1 s at -6 dBFS), 18 digital silence after 0.5 s of noise, 19 full-scale square.
    With mix=False every stream is white noise."""
    s = np.arange(first_stream, first_stream + n_streams, dtype=np.uint64)
    state = (np.uint32(0x9E3779B9) ^ ((s * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    state = np.where(state == 0, np.uint32(1), state).astype(np.uint32)
    out = np.empty((n_streams, n_frames, 2), dtype=np.int16)
    noise = np.empty((n_streams, n_frames, 2), dtype=np.int16)
    with np.errstate(over="ignore"):
        for i in range(n_frames):
            for c in range(2):
                state = _xorshift32(state)
                noise[:, i, c] = ((state >> np.uint32(16)).astype(np.int32) % 32769 - 16384).astype(np.int16)
    out[:] = noise
    if not mix:
        return out
    cls = (np.arange(first_stream, first_stream + n_streams) % 20)
    t = np.arange(n_frames) / float(fs)
    dur = max(n_frames / float(fs), 1e-3)
    k = np.log(20000.0 / 20.0) / dur
    phase = 2 * np.pi * 20.0 * (np.exp(k * t) - 1.0) / k
    sweep = np.stack([np.sin(phase), np.cos(phase)], axis=-1) * (32767.0 * 10 ** (-12 / 20.0))
    sq = np.where((np.arange(n_frames) // 24) % 2 == 0, 32767, -32768).astype(np.int16)
    burst_gain = np.where((t % 2.0) < 1.0, 10 ** (-30 / 20.0) / 0.5, 1.0)[:, None]     # noise is at -6 dBFS
    for j in range(n_streams):
        c = cls[j]
        if c in (14, 15):
            out[j] = sweep.astype(np.int16)
        elif c in (16, 17):
            out[j] = (noise[j].astype(np.float64) * burst_gain).astype(np.int16)
        elif c == 18:
            out[j, int(0.5 * fs):] = 0
        elif c == 19:
            out[j, :, 0] = sq
            out[j, :, 1] = -sq - 1
    return out


def pcm16_to_pcm24_bytes(pcm16: np.ndarray, seed: int = 7) -> np.ndarray:
    """24-bit variant: value = (v16 << 8) + low_byte; returns uint8 [stream][frame*6] packed LE."""
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, size=pcm16.shape, dtype=np.int32)
    v = (pcm16.astype(np.int32) << 8) + low
    u = v.astype("<i4").view(np.uint8).reshape(pcm16.shape + (4,))[..., :3]
    return np.ascontiguousarray(u).reshape(pcm16.shape[0], -1)
