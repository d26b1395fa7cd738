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

def _seeds(default):
    """seeds of one test: DSPI_FUZZ_SEEDS of them (default: a handful) starting at DSPI_FUZZ_SEED0 (default 0) — a longer or a FRESH sweep is
    `DSPI_FUZZ_SEED0=400 DSPI_FUZZ_SEEDS=400 pytest tests/test_gpu_fuzz.py -m gpu`"""
    s0 = int(os.environ.get("DSPI_FUZZ_SEED0", 0))
    return range(s0, s0 + int(os.environ.get("DSPI_FUZZ_SEEDS", default)))


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


@pytest.mark.all_layouts
@pytest.mark.parametrize("flavor", (1, W.F32_FMA, 0))
@pytest.mark.parametrize("seed", _seeds(12))      # more seeds: DSPI_FUZZ_SEEDS=200 pytest ...
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


def random_request(rng, flavor, fs):
    """One random SET request of the DSP subset (config.h:111-251) with an in-range or deliberately out-of-range payload."""
    import struct
    C, N, _, _, _ = W.dims(flavor)
    R = W.REQ
    f = lambda v: struct.pack("<f", float(v))
    pick = int(rng.integers(0, 22))
    if pick == 0:
        ch, band = int(rng.integers(0, C)), int(rng.integers(0, 10))
        return R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, band, int(rng.integers(0, 6)), 0, float(np.exp(rng.uniform(np.log(15), np.log(0.48 * fs)))),
                                                 float(np.exp(rng.uniform(np.log(0.08), np.log(25)))), float(rng.uniform(-12, 12)))
    if pick == 1: return R["SET_PREAMP"], 0, f(rng.uniform(-20, 8))
    if pick == 2: return R["SET_BYPASS"], 0, bytes([int(rng.integers(0, 2))])
    if pick == 3: return R["SET_LOUDNESS"], 0, bytes([int(rng.integers(0, 2))])
    if pick == 4: return R["SET_LOUDNESS_REF"], 0, f(rng.uniform(60, 100))
    if pick == 5: return R["SET_LOUDNESS_INTENSITY"], 0, f(rng.uniform(0, 200))
    if pick == 6: return R["SET_CROSSFEED"], 0, bytes([int(rng.integers(0, 2))])
    if pick == 7: return R["SET_CROSSFEED_PRESET"], 0, bytes([int(rng.integers(0, 4))])
    if pick == 8: return R["SET_CROSSFEED_FREQ"], 0, f(rng.uniform(300, 2500))
    if pick == 9: return R["SET_CROSSFEED_FEED"], 0, f(rng.uniform(0, 16))
    if pick == 10: return R["SET_CROSSFEED_ITD"], 0, bytes([int(rng.integers(0, 2))])
    if pick == 11:
        return R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", int(rng.integers(0, 2)), int(rng.integers(0, N)), int(rng.integers(0, 2)), int(rng.integers(0, 2)),
                                                    float(rng.uniform(-20, 4)))
    if pick == 12: return R["SET_OUTPUT_ENABLE"], int(rng.integers(0, N)), bytes([int(rng.integers(0, 2))])
    if pick == 13: return R["SET_OUTPUT_GAIN"], int(rng.integers(0, N)), f(rng.uniform(-30, 8))
    if pick == 14: return R["SET_OUTPUT_MUTE"], int(rng.integers(0, N)), bytes([int(rng.integers(0, 2))])
    if pick == 15: return R["SET_OUTPUT_DELAY"], int(rng.integers(0, N)), f(rng.choice([0.0, rng.uniform(0, 0.4), rng.uniform(0, 100)]))
    if pick == 16: return R["SET_LEVELLER_ENABLE"], 0, bytes([int(rng.integers(0, 2))])
    if pick == 17: return R["SET_LEVELLER_AMOUNT"], 0, f(rng.uniform(-10, 120))
    if pick == 18: return R[str(rng.choice(["SET_LEVELLER_SPEED", "SET_LEVELLER_LOOKAHEAD"]))], 0, bytes([int(rng.integers(0, 3))])
    if pick == 19: return R[str(rng.choice(["SET_LEVELLER_MAX_GAIN", "SET_LEVELLER_GATE"]))], 0, f(rng.uniform(-100, 30))
    if pick == 20: return R["SET_PREAMP_CH"], int(rng.integers(0, 2)), f(rng.uniform(-20, 8))
    return R["SET_MASTER_VOLUME"], 0, f(rng.choice([0.0, -128.0, rng.uniform(-70, 0)]))


@pytest.mark.all_layouts
@pytest.mark.parametrize("flavor", (1, W.F32_FMA, 0))
@pytest.mark.parametrize("seed", _seeds(6))
def test_random_request_sequences(flavor, seed):
    """Random vendor SET requests between launches — to every stream or to one — with their side effects on audio state
    (filter-path resets, crossfeed / leveller resets, delay changes, mutes), host volume / mute changes, and a sample-rate
    change in the middle; compared packet by packet with one oracle per stream."""
    from dspi_amd.host import Dspi
    from dspi_amd import workloads as WL
    from orclib import Oracle
    rng = np.random.default_rng(77000 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    S = 5
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    for x in [d] + o:
        assert x.set_rate(fs) == 0
        x.set_volume(-12 * 256)
        assert x.load_bulk(WL.full_chain_blob(flavor)) == 0
    n_steps, per = 10, 5
    B = int(rng.choice(Bs))
    pcm = WL.synth_pcm16(S, 97 * per * n_steps, 48000, first_stream=int(rng.integers(0, 30)))
    pos = 0
    for k in range(n_steps):
        for _ in range(int(rng.integers(1, 4))):
            req, wv, payload = random_request(rng, flavor, fs)
            target = None if rng.random() < 0.5 else int(rng.integers(0, S))
            rc_o = [oo.vendor_set(req, wv, payload) for i, oo in enumerate(o) if target is None or i == target]
            rc_d = d.vendor_set(req, wv, payload) if target is None else d.vendor_set(req, wv, payload, stream=target)
            assert (rc_d == 0) == all(r == 0 for r in rc_o), f"step {k}: request {req:#x} accepted differently"
        if rng.random() < 0.3:
            v = int(rng.choice([0, -256 * 30, -256 * 3, 256 * 2]))
            d.set_volume(v); [oo.set_volume(v) for oo in o]
        big = rng.random()
        if big < 0.08:                                   # host mute toggles
            mu = bool(rng.integers(0, 2)); d.set_mute(mu); [oo.set_mute(mu) for oo in o]
        elif big < 0.14:                                 # a whole-state blob in the middle of the stream (bulk_params_apply)
            blob = random_blob(rng, flavor, fs)
            assert d.load_bulk(blob) == 0 and all(oo.load_bulk(blob) == 0 for oo in o)
        elif big < 0.18:                                 # preset load: delay lines cleared, preset mute envelope (flash_storage.c:832, main.c:449-458)
            ref = Oracle(flavor); ref.set_rate(fs); ref.load_bulk(random_blob(rng, flavor, fs)); image = ref.save_slot(0); ref.close()
            assert d.load_slot(image) == 0 and all(oo.load_slot(image) == 0 for oo in o)
        elif big < 0.21:
            d.factory_defaults(); [oo.factory_defaults() for oo in o]
        if k == n_steps // 2:
            fs, Bs = RATES[(seed + 1) % 3]
            B = int(rng.choice(Bs))
            assert d.set_rate(fs) == 0 and all(oo.set_rate(fs) == 0 for oo in o)
        chunk = np.ascontiguousarray(pcm[:, pos:pos + per * B]); pos += per * B
        pairs, sub, peaks = d.process_host(chunk, per, B)
        for s in range(S):
            rp, rs, rk, _ = o[s].process(chunk[s], per, B)
            assert np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), f"step {k} stream {s}"
            assert o[s].status() == d.status(s), f"step {k} stream {s}: status"
    d.close()


@pytest.mark.all_layouts
@pytest.mark.parametrize("flavor", (1, W.F32_FMA, 0))
@pytest.mark.parametrize("seed", _seeds(6))
def test_output_pointer_and_layout_combinations(flavor, seed):
    """dspi_process with any subset of {pairs, sub, peaks} requested, in either layout, 16- or 24-bit input, must hand back
    exactly what a full stream-major call returns (one context per variant, same input, two launches each)."""
    from dspi_amd.host import Dspi
    from dspi_amd import workloads as WL
    rng = np.random.default_rng(4200 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    B = int(rng.choice(Bs)); blocks = int(rng.integers(3, 9)); S = int(rng.choice([5, 70, 131]))
    depth = 16 if rng.random() < 0.5 else 24
    blob = random_blob(rng, flavor, fs)
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs, first_stream=int(rng.integers(0, 20)))
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    half = (lambda c: data[:, c * blocks * B:(c + 1) * blocks * B]) if depth == 16 else (lambda c: data[:, c * blocks * B * 6:(c + 1) * blocks * B * 6])

    def run(**kw):
        d = Dspi(flavor, S, device=0); assert d.set_rate(fs) == 0; d.set_volume(-9 * 256); assert d.load_bulk(blob) == 0
        outs = [d.process_host(np.ascontiguousarray(half(c)), blocks, B, depth, **kw) for c in range(2)]
        if kw.get("tiled"):
            outs = [d.untile(p, s_) + (k,) for (p, s_, k) in outs]
        d.close()
        return outs
    full = run()
    for _ in range(4):
        kw = dict(want_pairs=bool(rng.integers(0, 2)), want_sub=bool(rng.integers(0, 2)), want_peaks=bool(rng.integers(0, 2)), tiled=bool(rng.integers(0, 2)))
        got = run(**kw)
        for c in range(2):
            for name, a_, b_ in zip(("pairs", "sub", "peaks"), full[c], got[c]):
                if b_ is not None:
                    assert np.array_equal(a_, b_), f"{name} differ with {kw}, launch {c}"


@pytest.mark.parametrize("flavor", (1, W.F32_FMA, 0))
@pytest.mark.parametrize("seed", _seeds(4))
def test_random_preset_per_stream(flavor, seed):
    """Every stream loads its own random blob: band kinds, flags, delays and leveller / crossfeed / loudness settings all
    differ from lane to lane inside one workgroup (per-lane parameter kernels, SIMT divergence on the band forms); a few
    streams keep a shared blob so that packed rows and per-lane rows coexist in one launch."""
    from dspi_amd.host import Dspi
    from dspi_amd import workloads as WL
    from orclib import Oracle
    rng = np.random.default_rng(9100 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    B = int(rng.choice(Bs)); blocks = int(rng.integers(4, 10)); S = int(rng.choice([9, 40, 140]))
    depth = 16 if rng.random() < 0.5 else 24
    shared = random_blob(rng, flavor, fs)
    d = Dspi(flavor, S, device=0); assert d.set_rate(fs) == 0; d.set_volume(-6 * 256); assert d.load_bulk(shared) == 0
    blobs = {}
    for s in range(S):
        if rng.random() < 0.7:
            blobs[s] = random_blob(rng, flavor, fs)
            assert d.load_bulk(blobs[s], stream=s) == 0
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs, first_stream=int(rng.integers(0, 20)))
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    step = blocks * B * (1 if depth == 16 else 6)        # int16 [S][frames][2] or bytes [S][frames * 6]
    outs = [d.process_host(np.ascontiguousarray(data[:, c * step:(c + 1) * step]), blocks, B, depth) for c in range(2)]
    pairs = np.concatenate([o_[0] for o_ in outs], axis=2); sub = np.concatenate([o_[1] for o_ in outs], axis=1); peaks = np.concatenate([o_[2] for o_ in outs], axis=1)
    for s in sorted(set(int(x) for x in rng.integers(0, S, 12))):
        o = Oracle(flavor, detmath=True); assert o.set_rate(fs) == 0; o.set_volume(-6 * 256); assert o.load_bulk(shared) == 0
        if s in blobs: assert o.load_bulk(blobs[s]) == 0
        rp, rs, rk, _ = o.process(data[s], blocks * 2, B, depth)
        assert np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), f"stream {s} ({'own' if s in blobs else 'shared'} blob)"
        assert o.status() == d.status(s), f"stream {s}: status"
        o.close()
    d.close()


@pytest.mark.all_layouts
@pytest.mark.parametrize("flavor", (1, W.F32_FMA))
@pytest.mark.parametrize("seed", _seeds(8))
def test_random_numbers_one_random_structure(flavor, seed):
    """One random preset for everybody, then every stream changes NUMBERS in it — band gains and Q at the band's own type and frequency,
    preamps, master volume, output gains, the gains of routed crosspoints, leveller amount / speed / max gain / gate, crossfeed — so that
    most workgroups hold several presets of one structure: paired presets on the latency layout (dspi_chain_skew.inc SkNum: a workgroup's
    stream slots each read their own image), value tiles on the packed kernel.  Where a number does change the structure (a band gone
    flat, a loudness shelf switched by the volume) the workgroup runs once per image: either way every sampled stream must equal its
    own oracle over two launches, with more changes in between."""
    import struct
    from dspi_amd.host import Dspi
    from dspi_amd import workloads as WL
    from orclib import Oracle
    rng = np.random.default_rng(9900 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    B = int(rng.choice(list(Bs) + [1, 7])); blocks = int(rng.integers(3, 9)); S = int(rng.choice([7, 33, 70, 140]))
    depth = 16 if rng.random() < 0.5 else 24
    blob = random_blob(rng, flavor, fs)
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    watch = sorted(set(int(x) for x in rng.integers(0, S, 10)) | {0, S - 1})
    d = Dspi(flavor, S, device=0); o = {s_: Oracle(flavor, detmath=True) for s_ in watch}
    for x in [d] + list(o.values()):
        assert x.set_rate(fs) == 0; x.set_volume(-6 * 256); assert x.load_bulk(blob) == 0

    def numbers():
        reqs = [(R["SET_PREAMP_CH"], 0, f(float(rng.uniform(-12, 3)))), (R["SET_PREAMP_CH"], 1, f(float(rng.uniform(-12, 3)))), (R["SET_MASTER_VOLUME"], 0, f(float(rng.uniform(-20, 0))))]
        for _ in range(int(rng.integers(2, 10))):
            ch, band = int(rng.integers(0, 11)), int(rng.integers(0, 10))
            p_ = blob["eq"][ch][band]
            if int(p_["type"]) == W.FILTER_FLAT: continue
            reqs.append((R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, band, int(p_["type"]), 0, float(p_["freq"]), float(rng.uniform(0.3, 4.0)), float(rng.uniform(0.5, 11.0)) * (1.0 if rng.random() < 0.5 else -1.0))))
        for _ in range(2): reqs.append((R["SET_OUTPUT_GAIN"], int(rng.integers(0, 9)), f(float(rng.uniform(-18, 5)))))
        for _ in range(3):
            i_, o2 = int(rng.integers(0, 2)), int(rng.integers(0, 9))
            xp = blob["crosspoints"][i_][o2]
            if int(xp["enabled"]): reqs.append((R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", i_, o2, 1, int(xp["phase_invert"]), float(rng.uniform(-15, 2)))))
        if int(blob["leveller"]["enabled"]):
            reqs += [(R["SET_LEVELLER_AMOUNT"], 0, f(float(rng.uniform(5, 100)))), (R["SET_LEVELLER_SPEED"], 0, bytes([int(rng.integers(0, 3))])),
                     (R["SET_LEVELLER_MAX_GAIN"], 0, f(float(rng.uniform(1, 22)))), (R["SET_LEVELLER_GATE"], 0, f(float(rng.uniform(-90, -45))))]
        if int(blob["crossfeed"]["enabled"]) and rng.random() < 0.7:
            reqs += [(R["SET_CROSSFEED_PRESET"], 0, b"\x03"), (R["SET_CROSSFEED_FREQ"], 0, f(float(rng.uniform(500, 1500)))), (R["SET_CROSSFEED_FEED"], 0, f(float(rng.uniform(3, 12))))]
        return reqs

    pcm = WL.synth_pcm16(S, B * blocks * 2, fs, first_stream=int(rng.integers(0, 20)))
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    step = blocks * B * (1 if depth == 16 else 6)
    for c in range(2):
        for s_ in range(S):
            if c == 1 and rng.random() < 0.6: continue
            for req, wv, pl in numbers():
                rc = d.vendor_set(req, wv, pl, stream=s_)
                if s_ in o: assert o[s_].vendor_set(req, wv, pl) == rc
        chunk = np.ascontiguousarray(data[:, c * step:(c + 1) * step])
        # the second launch in a random output form: tiled words, silent outputs left unwritten (into zeroed buffers: what the firmware's
        # zero-fill gives), the sticky clip flags along
        kw = dict(tiled=bool(rng.random() < 0.5), enabled_only=bool(rng.random() < 0.3), clip=True) if c == 1 else {}
        pairs, sub, peaks = d.process_host(chunk, blocks, B, depth, **kw)
        if kw.get("tiled"): pairs, sub = d.untile(pairs, sub)
        for s_ in watch:
            rp, rs, rk, _ = o[s_].process(chunk[s_], blocks, B, depth)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), f"launch {c}, stream {s_}, {kw}"
            if kw.get("clip"): assert int(d.last_clip[s_]) == int.from_bytes(o[s_].status()[-2:], "little"), f"clip flags, stream {s_}"
            assert o[s_].status() == d.status(s_), f"launch {c}, stream {s_}: status"
    for x in o.values(): x.close()
    d.close()

