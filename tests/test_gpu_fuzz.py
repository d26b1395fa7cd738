"""Randomised (fixed seeds) parity sweep: GPU chain vs the oracle over random presets — band types, frequencies, Q, gains,
output enables / mutes / delays / crosspoints, leveller / crossfeed / loudness settings, preamp and volumes — at random
rates, packet lengths and input depths.  The scenario tests pin the paths one by one; this one looks for combinations
nobody thought of.  Every case is two launches (state carries across) and compares every word, peak and status byte."""
import os

import numpy as np
import pytest

from conftest import has_gpu
from dspi_amd import wire as W

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]

RATES = [(44100, (44, 45)), (48000, (48,)), (96000, (96,))]


def random_blob(rng, flavor, fs):
    C, N, _, _, _ = W.dims(flavor)
    b = W.new_bulk(flavor)
    g = b["global_"]
    g["preamp_gain_db"] = rng.uniform(-12, 6)
    g["bypass"] = rng.random() < 0.1
    g["loudness_enabled"] = rng.random() < 0.5
    g["loudness_ref_spl"] = rng.uniform(70, 95)
    g["loudness_intensity_pct"] = rng.uniform(0, 150)
    b["preamp"]["preamp_db"][:] = rng.uniform(-12, 6, 2)
    b["master_volume"]["master_volume_db"] = rng.choice([0.0, -3.0, -20.0, -60.0])
    xf = b["crossfeed"]
    xf["enabled"] = rng.random() < 0.5
    xf["preset"] = rng.integers(0, 4)
    xf["itd_enabled"] = rng.integers(0, 2)
    xf["custom_fc"] = rng.uniform(400, 1200)
    xf["custom_feed_db"] = rng.uniform(2, 10)
    lv = b["leveller"]
    lv["enabled"] = rng.random() < 0.6
    lv["speed"] = rng.integers(0, 3)
    lv["lookahead"] = rng.integers(0, 2)
    lv["amount"] = rng.uniform(0, 100)
    lv["max_gain_db"] = rng.uniform(0, 24)
    lv["gate_threshold_db"] = rng.choice([-96.0, -60.0, -40.0])
    max_ms = (4096 if flavor else 2048) * 1000.0 / fs
    for ch in range(C):
        for k in range(10):
            t = rng.choice([W.FILTER_FLAT, W.FILTER_PEAKING, W.FILTER_LOWSHELF, W.FILTER_HIGHSHELF, W.FILTER_LOWPASS, W.FILTER_HIGHPASS],
                           p=[0.2, 0.35, 0.12, 0.12, 0.1, 0.11])
            f = float(np.exp(rng.uniform(np.log(12.0), np.log(0.49 * fs))))       # beyond both clamps on purpose
            q = float(np.exp(rng.uniform(np.log(0.05), np.log(30.0))))
            gain = 0.0 if rng.random() < 0.1 else rng.uniform(-12, 12)
            W.set_band(b, ch, k, t, f, q, gain)
    for o in range(N):
        out = b["outputs"][o]
        out["enabled"] = rng.random() < 0.85
        out["mute"] = rng.random() < 0.1
        out["gain_db"] = rng.choice([0.0, rng.uniform(-20, 6)])
        out["delay_ms"] = rng.choice([0.0, rng.uniform(0, 0.3), rng.uniform(0, max_ms * 1.1)])
        for inp in (0, 1):
            xp = b["crosspoints"][inp, o]
            xp["enabled"] = rng.random() < 0.7
            xp["phase_invert"] = rng.random() < 0.2
            xp["gain_db"] = rng.choice([0.0, rng.uniform(-18, 3)])
    return b


@pytest.mark.parametrize("flavor", (1, 0))
@pytest.mark.parametrize("seed", range(int(os.environ.get("DSPI_FUZZ_SEEDS", 12))))      # more seeds: DSPI_FUZZ_SEEDS=200 pytest ...
def test_random_presets(flavor, seed):
    from test_gpu_parity import compare
    rng = np.random.default_rng(1000 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    B = int(rng.choice(Bs))
    depth = 16 if rng.random() < 0.5 else 24
    vol = int(rng.choice([0, -5 * 256, -20 * 256, -40 * 256, 3 * 256]))
    S = int(rng.choice([3, 20, 67, 130]))
    blob = random_blob(rng, flavor, fs)
    compare(flavor, fs, B, 2 * int(rng.integers(6, 20)), S, blob, vol=vol, depth=depth, calls=2,
            check_streams=sorted(set(int(x) for x in rng.integers(0, S, 6))), first_stream=int(rng.integers(0, 40)))
