"""GPU parity: the HIP path through the C-ABI (libdspi_mi355x.so) against the CPU oracle, bit-exact, on the same seeded
inputs; golden fixtures; and size-independent properties at BASELINE.json's full stream count.  Needs an MI355X."""
import glob
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import has_gpu
import orclib
from orclib import Oracle, PdmOracle
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no GPU")]

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if "thumb" not in os.path.basename(p))      # (q28_thumb_biquad.npz: test_oracle_thumb.py)
# float canonical, float with the firmware build's FMA contraction (every scenario runs both ways), Q28
FLAVORS_WITH_KERNEL = (1, W.F32_FMA, 0)


def on_latency_layout() -> bool:
    """The run of a `both_layouts` test that pins the latency layout (conftest._layout)."""
    return os.environ.get("DSPI_F32_LAYOUT") == "skew"


def latency_plan(plan) -> bool:
    """Every float lane on the latency layout: nothing left for the packed, per-lane-value or one-stream kernels."""
    return plan["latency_layout"] > 0 and plan["packed_shared"] == plan["packed_per_lane_values"] == plan["packed_per_lane_values_and_bands"] == plan["one_stream_per_lane_images"] == 0


def oracle_run(flavor, fs, vol, blob, data, blocks, B, depth, setup=None):
    o = Oracle(flavor, detmath=True)
    assert o.set_rate(fs) == 0
    o.set_volume(vol)
    assert o.load_bulk(blob) == 0
    if setup: setup(o)
    return o.process(data, blocks, B, depth), o.status()


def compare(flavor, fs, B, blocks, S, blob, vol=-20 * 256, depth=16, calls=2, check_streams=None, setup=None, first_stream=0):
    d = Dspi(flavor, S, device=0)
    assert d.set_rate(fs) == 0
    d.set_volume(vol)
    assert d.load_bulk(blob) == 0
    if setup: setup(d)
    pcm = WL.synth_pcm16(S, B * blocks, fs, first_stream=first_stream)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = blocks // calls
    outs = []
    bpf = 2 if depth == 16 else 6
    for c in range(calls):        # several launches: state must carry across dspi_process calls
        sl = data[:, c * per * B:(c + 1) * per * B] if depth == 16 else data[:, c * per * B * 6:(c + 1) * per * B * 6]
        outs.append(d.process_host(np.ascontiguousarray(sl), per, B, depth, clip=True))
    clip_flags = d.last_clip          # dspi_out.clip_flags (DSPI_OUT_CLIP_FLAGS): every stream's sticky bits after the last call
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s in (check_streams if check_streams is not None else range(S)):
        (rp, rs, rk, rclip), status = oracle_run(flavor, fs, vol, blob, data[s], per * calls, B, depth, setup)
        assert np.array_equal(rp, pairs[s]), f"pairs differ, stream {s}: {np.argwhere(rp != pairs[s])[:3].tolist()}"
        assert np.array_equal(rs, sub[s]), f"sub differs, stream {s}"
        assert np.array_equal(rk, peaks[s]), f"peaks differ, stream {s}"
        assert status == d.status(s), f"status differs, stream {s}"
        assert int(clip_flags[s]) == int.from_bytes(status[-2:], "little") == rclip, f"clip flags differ, stream {s}"
    d.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_config2_master_peq(flavor):
    """BASELINE config 2 (float) / config 1 chain (Q28): master 10-band PEQ only; both SVF and biquad paths."""
    if flavor:
        compare(1, 48000, 48, 40, 130, WL.config2_blob(False), vol=-10 * 256)
        compare(1, 48000, 48, 40, 20, WL.config2_blob(True), vol=-10 * 256)
    else:
        compare(0, 48000, 48, 40, 70, WL.config1_blob(), vol=-10 * 256)
        compare(0, 48000, 48, 20, 5, WL.config1_blob(), vol=0)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
@pytest.mark.parametrize("fs,B,depth", [(96000, 96, 16), (48000, 48, 24), (44100, 45, 16), (44100, 44, 24)])
def test_full_chain(flavor, fs, B, depth):
    """BASELINE config 3 / config 5: every stage on; all stream classes of the synthetic generator (noise, sweep, bursts,
    decay to digital silence, full-scale square); ragged stream count; 44/45-frame packets take the tail kernel."""
    compare(flavor, fs, B, 24, 85, WL.full_chain_blob(flavor), depth=depth)


# delays of every class at 44.1 / 48 / 96 kHz: none, shorter than 64 frames, than a packet, a few packets, within a packet's length of the line's
# 4096 positions (42.6 ms at 96 kHz, 85.2 ms at 48 kHz, 92.8 ms at 44.1 kHz: the rows just ahead of the write index), at and beyond them (alias)
_EDGE_DELAYS = (0.0, 0.3, 42.6, 0.0, 95.0, 0.05, 85.2, 1.7, 92.8)


def _latency_blob(xfeed=True, loud=True, delays=_EDGE_DELAYS):
    """A preset of the latency layout's class: master PEQ (+ loudness, crossfeed), leveller off, outputs routed, gained and delayed
    but without EQ; output 3 muted, output 5 disabled, crosspoint patterns both / left only / right only / inverted."""
    b = WL.config2_blob(False)
    b["preamp"]["preamp_db"][:] = (-2.0, -4.5)
    b["global_"]["loudness_enabled"] = 1 if loud else 0
    b["crossfeed"]["enabled"], b["crossfeed"]["preset"], b["crossfeed"]["itd_enabled"] = (1 if xfeed else 0), 1, 1
    for o in range(9):
        out = b["outputs"][o]
        out["enabled"] = 0 if o == 5 else 1
        out["mute"] = 1 if o == 3 else 0
        out["gain_db"] = -0.5 * o
        out["delay_ms"] = delays[o]
        for inp in (0, 1):
            xp = b["crosspoints"][inp, o]
            xp["enabled"] = 1 if (o % 3 == 0 or (o % 3 == 1 and inp == 0) or (o % 3 == 2 and inp == 1)) else 0
            xp["gain_db"] = -3.0 - inp
        b["crosspoints"][1, 6]["phase_invert"] = 1
    return b


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("fs,B,depth", [(48000, 48, 16), (96000, 96, 24), (44100, 45, 16), (44100, 44, 24), (48000, 1, 16), (48000, 7, 16), (48000, 97, 16), (96000, 192, 16)])
def test_latency_layout(flavor, fs, B, depth, monkeypatch):
    """The float chain's latency layout (dspi_chain_skew.inc: the EQ cascade skewed across lanes, outputs frame-parallel), forced for
    every eligible image: master PEQ with SVF and biquad bands, loudness shelves, crossfeed, delays shorter and longer than a packet and
    at the aliasing maximum, muted / disabled outputs, every packet length class, both input depths, ragged stream count, three calls."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    blob = _latency_blob()
    blocks = 24 if B >= 44 else 150
    S = 37          # 18 pairs + a pair that holds one stream: three workgroup parts, one of them partly filled
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-7 * 256); assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = blocks // 3
    unit = B if depth == 16 else B * 6                 # elements of the second axis per packet: frames, or bytes of packed 24-bit frames
    outs = [d.process_host(np.ascontiguousarray(data[:, c * per * unit:(c + 1) * per * unit]), per, B, depth) for c in range(3)]
    assert d.launch_plan()["latency_layout"] > 0 and d.launch_plan()["packed_shared"] == 0
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s in range(S):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-7 * 256); assert o.load_bulk(blob) == 0
        rp, rs, rk, _ = o.process(data[s][:per * 3 * unit], per * 3, B, depth)
        assert np.array_equal(rp, pairs[s]), f"pairs differ, stream {s}: {np.argwhere(rp != pairs[s])[:3].tolist()}"
        assert np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), s
        assert o.status() == d.status(s), s
    d.close()


@pytest.mark.parametrize("flavor", (W.F32_FMA, 0), ids=("f32fma", "q28"))
def test_one_packet_calls_both_ways_of_waiting(flavor, monkeypatch):
    """Small calls on host buffers wait for their launches by polling a completion word the stream writes behind them (hipStreamWriteValue32,
    round 6) or, with DSPI_DIRECT_POLL=query, the stream itself (rounds 4-5, and the fallback): the same words either way, call after call,
    equal to the oracle; the library's own record says which calls it counted (dspi_debug_direct_stats)."""
    fl = int(flavor)
    fs, B, calls, S = (96000, 96, 60, 5) if fl else (48000, 48, 60, 5)
    blob = WL.full_chain_blob(fl)
    pcm = WL.synth_pcm16(S, B * calls, fs)
    got = {}
    for mode in ("flag", "query"):
        if mode == "query": monkeypatch.setenv("DSPI_DIRECT_POLL", "query")
        else: monkeypatch.delenv("DSPI_DIRECT_POLL", raising=False)
        monkeypatch.delenv("DSPI_F32_LAYOUT", raising=False)
        d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
        outs = [d.process_host(np.ascontiguousarray(pcm[:, c * B:(c + 1) * B]), 1, B) for c in range(calls)]
        st = d.direct_stats()
        assert st["calls"] == calls and st["spin_budget_us"] >= 300.0, st
        got[mode] = (np.concatenate([o[0] for o in outs], axis=2), np.concatenate([o[1] for o in outs], axis=1), np.concatenate([o[2] for o in outs], axis=1))
        d.close()
    for a, b in zip(got["flag"], got["query"]): assert np.array_equal(a, b)
    for s in (0, S - 1):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-20 * 256); assert o.load_bulk(blob) == 0
        rp, rs, rk, _ = o.process(pcm[s], calls, B)
        assert np.array_equal(rp, got["flag"][0][s]) and np.array_equal(rs, got["flag"][1][s]) and np.array_equal(rk, got["flag"][2][s]), s


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("fs,B", [(48000, 48), (44100, 45), (96000, 192), (48000, 7)])
def test_latency_layout_lines_of_disabled_outputs(flavor, fs, B, monkeypatch):
    """The reference runs an output's delay line whenever its delay is non-zero, enabled or not (usb_audio.c:898-911); the latency layout's first
    shape fetches and zeroes such lines once per batch (dspi_chain_skew.inc, "lines of disabled outputs").  What the lines HOLD must stay the
    reference's: outputs are disabled and enabled again between calls (the line's old samples come out delayed; the zeros written while the
    output was off come out after them), with delays shorter than a packet, of several packets, within a batch of the line's length, and at
    the aliasing maximum; three disabled delayed outputs in one pair's reach (two take the batch path, the third the packet path), the sub
    with its alignment delay alone, and BASELINE config 2's own preset (only the sub's line runs)."""
    import struct
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    blob = _latency_blob(delays=(0.4, 3.1, 84.9 if fs == 48000 else 42.0, 85.4, 0.0, 0.05, 20.0, 1.7, 10.0))
    S, per = 21, (10 if B >= 44 else 60)
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-7 * 256); assert d.load_bulk(blob) == 0
    ors = []
    for s in range(S):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-7 * 256); assert o.load_bulk(blob) == 0
        ors.append(o)
    # call by call: which outputs are enabled (output 5 starts disabled in the blob)
    plans = [{}, {1: 0, 2: 0, 8: 0}, {3: 0, 6: 0}, {1: 1, 8: 1}, {2: 1, 3: 1, 5: 1}, {6: 1, 0: 0, 7: 0}, {0: 1, 7: 1}]
    pcm = WL.synth_pcm16(S, B * per * len(plans), fs)
    for c, plan in enumerate(plans):
        for o_, en in plan.items():
            d.vendor_set(W.REQ["SET_OUTPUT_ENABLE"], o_, struct.pack("<B", en))
            for o in ors: o.vendor_set(W.REQ["SET_OUTPUT_ENABLE"], o_, struct.pack("<B", en))
        x = np.ascontiguousarray(pcm[:, c * per * B:(c + 1) * per * B])
        gp, gs, gk = d.process_host(x, per, B)
        assert d.launch_plan()["latency_layout"] > 0 and d.launch_plan()["packed_shared"] == 0
        for s in range(S):
            rp, rs, rk, _ = ors[s].process(x[s], per, B)
            assert np.array_equal(rp, gp[s]), f"call {c}, stream {s}: pairs differ at {np.argwhere(rp != gp[s])[:3].tolist()}"
            assert np.array_equal(rs, gs[s]) and np.array_equal(rk, gk[s]), (c, s)
            assert ors[s].status() == d.status(s), (c, s)
    d.close()
    # BASELINE config 2's own preset: two outputs enabled, no user delay — the sub's alignment line is the only one that runs
    compare(flavor, fs, B, 2 * per, 19, WL.config2_blob(False), vol=-10 * 256)


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("fs,B,depth", [(48000, 48, 16), (96000, 96, 24), (44100, 45, 16), (44100, 44, 24), (48000, 1, 16), (48000, 7, 16), (48000, 13, 16), (48000, 14, 16), (48000, 97, 16), (96000, 192, 16)])
def test_latency_layout_output_rows(flavor, fs, B, depth, monkeypatch):
    """The latency layout's second shape (dspi_chain_skew.inc, EQO): presets whose OUTPUTS run EQs — the full chain of BASELINE config 3
    with the leveller off: loudness, master PEQ, crossfeed, 2 x 9 matrix, nine 10-band output EQs of every band form, gains, delays, the
    sub.  Two stream pairs per workgroup, every output a systolic row a batch behind the master rows; every packet length class (shorter
    than the extra lead of the master rows included), both input depths, a ragged stream count, three calls."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    blob = WL.full_chain_blob(1)
    blob["leveller"]["enabled"] = 0
    for o in range(9): blob["outputs"][o]["delay_ms"] = _EDGE_DELAYS[o]
    blocks = 24 if B >= 44 else 150
    S = 11          # five pairs + a pair that holds one stream: three workgroups, the last one half filled
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-7 * 256); assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = blocks // 3
    unit = B if depth == 16 else B * 6
    outs = [d.process_host(np.ascontiguousarray(data[:, c * per * unit:(c + 1) * per * unit]), per, B, depth) for c in range(3)]
    assert d.launch_plan()["latency_layout"] > 0 and d.launch_plan()["packed_shared"] == 0
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s in range(S):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-7 * 256); assert o.load_bulk(blob) == 0
        rp, rs, rk, _ = o.process(data[s][:per * 3 * unit], per * 3, B, depth)
        assert np.array_equal(rp, pairs[s]), f"pairs differ, stream {s}: {np.argwhere(rp != pairs[s])[:3].tolist()}"
        assert np.array_equal(rs, sub[s]), f"sub differs, stream {s}: {np.argwhere(rs != sub[s])[:3].tolist()}"
        assert np.array_equal(rk, peaks[s]), f"peaks differ, stream {s}: {np.argwhere(rk != peaks[s])[:3].tolist()}"
        assert o.status() == d.status(s), s
    d.close()


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("fs,B,depth,lookahead", [(48000, 48, 16, 1), (96000, 96, 24, 1), (44100, 45, 16, 0), (44100, 44, 24, 1), (48000, 1, 16, 1), (48000, 7, 16, 0), (48000, 13, 16, 1),
                                                  (48000, 97, 16, 1), (96000, 192, 16, 1)])
def test_latency_layout_leveller(flavor, fs, B, depth, lookahead, monkeypatch):
    """The latency layout's third shape (dspi_chain_skew_lev.inc): BASELINE config 3's preset itself — leveller ON (its per-packet gain
    decision, the ramp, the gain-cap limiter on look-ahead-delayed samples), crossfeed, nine equalised outputs — on a small context.
    Four calls, so that the look-ahead line and the envelopes cross launch boundaries; then the same context continues on the packed
    kernel (DSPI_F32_LAYOUT=packed) and must carry on where the latency layout left the rings and states, and back."""
    blob = WL.full_chain_blob(1)
    blob["leveller"]["lookahead"] = lookahead
    for o in range(9): blob["outputs"][o]["delay_ms"] = _EDGE_DELAYS[(o + 3) % 9]
    blocks = 32 if B >= 44 else 200
    S = 7
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-7 * 256); assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = blocks // 4
    unit = B if depth == 16 else B * 6
    outs = []
    tweak = lambda c: struct.pack("<BBBBfff", 0, 3, W.FILTER_PEAKING, 0, 700.0, 1.1, 1.5 * c)
    for c in range(4):
        monkeypatch.setenv("DSPI_F32_LAYOUT", "packed" if c == 2 else "skew")
        if c >= 2: d.vendor_set(W.REQ["SET_EQ_PARAM"], 0, tweak(c))      # (a band changes: the launch lists are rebuilt, under the new setting)
        outs.append(d.process_host(np.ascontiguousarray(data[:, c * per * unit:(c + 1) * per * unit]), per, B, depth))
        plan = d.launch_plan()
        assert (plan["latency_layout"] == 0) == (c == 2) and (plan["packed_shared"] > 0) == (c == 2), (c, plan)
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s in range(S):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-7 * 256); assert o.load_bulk(blob) == 0
        parts = []
        for c in range(4):
            if c >= 2: o.vendor_set(W.REQ["SET_EQ_PARAM"], 0, tweak(c))
            parts.append(o.process(data[s][c * per * unit:(c + 1) * per * unit], per, B, depth))
        rp = np.concatenate([q[0] for q in parts], axis=1); rs = np.concatenate([q[1] for q in parts]); rk = np.concatenate([q[2] for q in parts])
        assert np.array_equal(rp, pairs[s]), f"pairs differ, stream {s}: {np.argwhere(rp != pairs[s])[:3].tolist()}"
        assert np.array_equal(rs, sub[s]), f"sub differs, stream {s}: {np.argwhere(rs != sub[s])[:3].tolist()}"
        assert np.array_equal(rk, peaks[s]), f"peaks differ, stream {s}: {np.argwhere(rk != peaks[s])[:3].tolist()}"
        assert o.status() == d.status(s), s
    d.close()


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("shape,B,per", [(1, 45, 5), (2, 45, 5), (3, 45, 5), (3, 97, 1), (2, 7, 27), (3, 96, 1)])
def test_latency_layout_bypassed_bands_keep_their_state(flavor, shape, B, per, monkeypatch):
    """A bypassed band's state is left alone (dsp_pipeline.c:288-289: `if (bq->bypass) continue`), so REQ_SET_BYPASS on and off again must
    find the master EQ's filter states where they were.  The latency layout's straight-line step loop updates every lane's state
    registers, and with zero coefficients an idle lane's s1 changes sign every step: launches with an ODD number of such steps (found by
    the fuzz sweep at seed 66 once one-packet launches took that loop) wrote the flipped state back.  Every shape, odd and even step
    counts, one packet per call and many."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    fs = 44100 if B == 45 else 48000
    blob = _latency_blob() if shape == 1 else WL.full_chain_blob(1)
    if shape == 2: blob["leveller"]["enabled"] = 0
    S = 5
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    for x in [d] + o:
        assert x.set_rate(fs) == 0
        x.set_volume(-8 * 256); assert x.load_bulk(blob) == 0
    seq = [None, 1, None, 0, None, 1, 0, None]      # REQ_SET_BYPASS payloads before each call (None: no request)
    pcm = WL.synth_pcm16(S, B * per * len(seq), fs, first_stream=4)
    for k, byp in enumerate(seq):
        if byp is not None:
            for x in [d] + o: assert x.vendor_set(W.REQ["SET_BYPASS"], 0, bytes([byp])) == 0
        part = np.ascontiguousarray(pcm[:, k * B * per:(k + 1) * B * per])
        pairs, sub, peaks = d.process_host(part, per, B)
        assert latency_plan(d.launch_plan())
        for s_ in range(S):
            rp, rs, rk, _ = o[s_].process(part[s_], per, B)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (k, s_)
            assert o[s_].status() == d.status(s_)
    d.close()


@pytest.mark.auto_layout
@pytest.mark.parametrize("shape", (1, 2, 3))
@pytest.mark.parametrize("S", (1, 3))
def test_one_stream_context_takes_the_latency_layout(shape, S):
    """The firmware's own use case: ONE stream (and three: a pair and a half).  A lane that holds a single stream is left to the one-stream
    kernel by the packed kernel; the latency layout serves it, so such contexts run on it entirely — by the library's own choice."""
    fs, B, blocks = 48000, 48, 30
    blob = _latency_blob() if shape == 1 else WL.full_chain_blob(1)
    if shape == 2: blob["leveller"]["enabled"] = 0
    d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-7 * 256); assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    outs = [d.process_host(np.ascontiguousarray(pcm[:, c * B * 10:(c + 1) * B * 10]), 10, B) for c in range(3)]
    plan = d.launch_plan()
    assert plan["latency_layout"] > 0 and plan["one_stream_per_lane_images"] == 0 and plan["packed_shared"] == 0, plan
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s in range(S):
        o = Oracle(W.F32_FMA, detmath=True); o.set_rate(fs); o.set_volume(-7 * 256); assert o.load_bulk(blob) == 0
        rp, rs, rk, _ = o.process(pcm[s], blocks, B)
        assert np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), s
        assert o.status() == d.status(s), s
    d.close()


def test_enabled_only_leaves_silent_outputs_unwritten(monkeypatch):
    """DSPI_OUT_ENABLED_ONLY (include/dspi.h): silent outputs (disabled pairs, the sub while off) may stay unwritten — the latency
    layout skips their stores —, every live word, peak and status byte is what it is without the flag."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    fs, B, blocks, S = 48000, 48, 9, 40
    blob = WL.config2_blob(False)
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    ref = Dspi(W.F32_FMA, S, device=0); ref.set_rate(fs); ref.set_volume(-10 * 256); assert ref.load_bulk(blob) == 0
    rp, rs, rk = ref.process_host(pcm, blocks, B)
    d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-10 * 256); assert d.load_bulk(blob) == 0
    import torch
    dev = torch.device("cuda", 0)
    tp = torch.full((S, 4, B * blocks, 2), 0x5A5A5A5A, dtype=torch.int32, device=dev); ts = torch.full((S, B * blocks), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    tk = torch.zeros((S, blocks, 11), dtype=torch.int16, device=dev)
    t_in = torch.from_numpy(pcm).to(dev)
    torch.cuda.synchronize()
    d.process_device(t_in.data_ptr(), blocks, B, 16, tp.data_ptr(), ts.data_ptr(), tk.data_ptr(), enabled_only=True); d.sync()
    p, s_, k = tp.cpu().numpy(), ts.cpu().numpy(), tk.cpu().numpy().view(np.uint16)
    assert d.launch_plan()["latency_layout"] > 0
    assert np.array_equal(p[:, 0], rp[:, 0]) and np.array_equal(k, rk)
    assert int(np.abs(rp[:, 1:]).max()) == 0 and int(np.abs(rs).max()) == 0            # what the flag lets the library skip is silence
    assert bool((p[:, 1:] == 0x5A5A5A5A).all()) and bool((s_ == 0x5A5A5A5A).all())     # ... and on device buffers it was skipped
    for s in (0, 1, S - 1): assert ref.status(s) == d.status(s)
    ref.close(); d.close()


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
def test_latency_layout_mixed_with_other_kernels(flavor, monkeypatch):
    """One context on the latency layout with lanes of every kind: most stream pairs share the latency-class preset, two streams carry presets
    of their own — their lanes run twice, once per image, each time with the other half inactive.  Then the shared preset changes class
    twice, mid-stream, on the same state arrays: an output EQ band becomes active (the latency layout's second shape: output rows), then
    the leveller is switched on (its third shape); tiled words."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    fs, B, blocks, S = 48000, 48, 12, 300
    blob = _latency_blob(xfeed=True, loud=False)
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-9 * 256); assert d.load_bulk(blob) == 0
    def special(x, s):
        if s in (5, 130): x.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -8.0)) if isinstance(x, Oracle) else x.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -8.0), stream=s)
    for s in (5, 130): special(d, s)
    pcm = WL.synth_pcm16(S, B * blocks * 3, fs)
    o1 = d.process_host(np.ascontiguousarray(pcm[:, :B * blocks]), blocks, B, tiled=True)
    plan = d.launch_plan()
    assert latency_plan(plan), plan      # streams 5 and 130 as half-active pairs of their own images, their lane partners 4 and 131 likewise on the common one
    p1, s1 = d.untile(o1[0], o1[1])
    # class change for everyone: output 1 gets a live EQ band -> output rows from here on
    eq = struct.pack("<BBBBfff", 3, 2, W.FILTER_PEAKING, 0, 900.0, 1.2, 4.0)
    d.vendor_set(W.REQ["SET_EQ_PARAM"], 0, eq)
    o2 = d.process_host(np.ascontiguousarray(pcm[:, B * blocks:2 * B * blocks]), blocks, B, tiled=True)
    plan = d.launch_plan()
    assert plan["latency_layout"] > 0 and plan["packed_shared"] == 0, plan
    p2, s2 = d.untile(o2[0], o2[1])
    # ... and the leveller on: the third shape (rings between the groups)
    d.vendor_set(W.REQ["SET_LEVELLER_ENABLE"], 0, b"\x01")
    o3 = d.process_host(np.ascontiguousarray(pcm[:, 2 * B * blocks:]), blocks, B, tiled=True)
    plan = d.launch_plan()
    assert plan["latency_layout"] > 0 and plan["packed_shared"] == 0, plan
    p3, s3 = d.untile(o3[0], o3[1])
    for s in (0, 4, 5, 6, 129, 130, 131, 255, 256, S - 1):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-9 * 256); assert o.load_bulk(blob) == 0
        special(o, s)
        rp, rs, rk, _ = o.process(pcm[s][:B * blocks], blocks, B)
        assert np.array_equal(rp, p1[s]) and np.array_equal(rs, s1[s]) and np.array_equal(rk, o1[2][s]), ("first call", s)
        o.vendor_set(W.REQ["SET_EQ_PARAM"], 0, eq)
        rp, rs, rk, _ = o.process(pcm[s][B * blocks:2 * B * blocks], blocks, B)
        assert np.array_equal(rp, p2[s]) and np.array_equal(rs, s2[s]) and np.array_equal(rk, o2[2][s]), ("after the first class change", s)
        o.vendor_set(W.REQ["SET_LEVELLER_ENABLE"], 0, b"\x01")
        rp, rs, rk, _ = o.process(pcm[s][2 * B * blocks:], blocks, B)
        assert np.array_equal(rp, p3[s]) and np.array_equal(rs, s3[s]) and np.array_equal(rk, o3[2][s]), ("after the second class change", s)
        assert o.status() == d.status(s)
    d.close()


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("tiled", (True, False), ids=("tiled", "stream"))
@pytest.mark.parametrize("fs,B", [(96000, 96), (44100, 45), (48000, 7)])
def test_words_written_ahead_keep_the_lines_history(flavor, tiled, fs, B):
    """The packed kernel writes the word of frame f + dly at frame f (tiled layout: every output; stream-major: the sub) and stores a
    launch's last 4 096 frames to the line whatever the delay, because the reference's lines always hold the last 4 096 samples and a
    LATER, LONGER delay reads further back (REQ_SET_OUTPUT_DELAY clears nothing: usb_audio.c:1944-1951).  Launches of very different
    lengths (one packet ... longer than the lines), delays shorter and longer than a launch, not multiples of the packet, at the aliasing
    maximum, growing and shrinking between launches, an output switched to delay 0 and back: every word, sub word, peak and status
    byte against the oracle fed the same requests."""
    S = 9
    blob = WL.full_chain_blob(1)
    line_ms = 4096 * 1000.0 / fs
    delays0 = [0.0, 0.4, 1.0, 3.3, 7.1, 0.1, line_ms - 1.0, 20.0, 5.0]
    for o in range(9): blob["outputs"][o]["delay_ms"] = delays0[o]
    d = Dspi(flavor, S, device=0); o_ = [Oracle(flavor, detmath=True) for _ in range(S)]
    for x in [d] + o_:
        assert x.set_rate(fs) == 0
        x.set_volume(-9 * 256); assert x.load_bulk(blob) == 0
    f = lambda v: struct.pack("<f", v)
    R = W.REQ
    # (packets per launch, requests before it): delays grow past what the previous launches' own delays needed, shrink, go to 0 and come back
    plan = [(1, []), (50, []), (3, [(2, 9.0), (5, 30.0)]), (1, [(2, 0.0)]), (4600 // B + 1, [(2, 40.0), (1, line_ms + 2.0), (8, 12.0)]), (2, [(5, 0.2), (8, 0.0)]),
            (30, [(2, 2.0), (3, line_ms - 0.5), (8, 33.0)])]
    total = sum(n for n, _ in plan)
    pcm = WL.synth_pcm16(S, B * total, fs, first_stream=11)
    at = 0
    for n, reqs in plan:
        for out, ms in reqs:
            for x in [d] + o_: assert x.vendor_set(R["SET_OUTPUT_DELAY"], out, f(ms)) == 0
        part = np.ascontiguousarray(pcm[:, at * B:(at + n) * B])
        pairs, sub, peaks = d.process_host(part, n, B, tiled=tiled)
        if tiled: pairs, sub = d.untile(pairs, sub)
        plan_now = d.launch_plan()
        assert plan_now["packed_shared"] > 0 and plan_now["latency_layout"] == 0, plan_now
        for s_ in range(S):
            rp, rs, rk, _ = o_[s_].process(part[s_], n, B)
            assert np.array_equal(rp, pairs[s_]), (at, n, s_, np.argwhere(rp != pairs[s_])[:3].tolist())
            assert np.array_equal(rs, sub[s_]), (at, n, s_, np.argwhere(rs != sub[s_])[:3].tolist())
            assert np.array_equal(rk, peaks[s_]), (at, n, s_, np.argwhere(rk != peaks[s_])[:3].tolist())
            assert o_[s_].status() == d.status(s_), (at, n, s_)
        at += n
    d.close()


@pytest.mark.parametrize("waves", ["4", "7", None])
def test_q28_wave_layouts(waves, monkeypatch):
    """The Q28 kernel's two wave layouts (four waves: two outputs per wave; seven: one output per wave, for launches of at most one
    workgroup per CU) and the size rule that picks between them: the same words, peaks and sticky clip bits (the seven-wave
    layout ORs its bits into shared words) — full chain, ragged packets, a clipping stream class, two calls."""
    if waves is None: monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
    else: monkeypatch.setenv("DSPI_Q28_WAVES", waves)
    compare(0, 48000, 48, 24, 150, WL.full_chain_blob(0), calls=2)
    compare(0, 44100, 45, 20, 70, WL.full_chain_blob(0), depth=24, calls=2, first_stream=15)


@pytest.mark.parametrize("lev", [1, 0])
def test_q28_latency_layout_alternates_with_the_chain_kernels(lev, monkeypatch):
    """The Q28 latency layout (dspi_chain_q28_lat.inc: one stream per workgroup, one lane per (channel, stage)) shares the state array, the
    delay lines and the leveller ring with chain_kernel<0>: a context changes kernels between two calls when its size or the caller's
    packet count moves it.  One context, SEVEN calls, the kernel forced per call — latency, four waves, latency, seven waves, ... —
    ragged 45-frame packets, one-packet calls in between, delays at the corners of the 2 048-sample lines, per-stream presets (rows with
    several images), leveller with look-ahead on / off: every word, peak and status byte of every stream against the oracle; then the
    size rule itself (nothing forced: a context of this size takes the latency layout)."""
    fs, B, S = 44100, 45, 70
    b = WL.full_chain_blob(0)
    max_ms = 2048 * 1000.0 / fs
    for o, ms in enumerate([0.05, 0.3, max_ms + 1.0, max_ms - 0.05, 2.5]): b["outputs"][o]["delay_ms"] = ms
    b["leveller"]["enabled"] = lev
    plan = [("lat", None, 3), ("chain", "4", 2), ("lat", None, 1), ("chain", "7", 4), ("lat", None, 1), ("lat", None, 13), ("chain", "4", 1)]
    total = sum(n for _, _, n in plan)
    pcm = WL.synth_pcm16(S, B * total, fs, first_stream=11)
    own = {3: -7.5, 64: 2.0, S - 1: -1.25}
    d = Dspi(0, S, device=0)
    assert d.set_rate(fs) == 0
    d.set_volume(-12 * 256); assert d.load_bulk(b) == 0
    for s_, db in own.items(): d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", db), stream=s_)
    outs, at = [], 0
    for layout, waves, n in plan:
        monkeypatch.setenv("DSPI_Q28_LAYOUT", layout)
        if waves: monkeypatch.setenv("DSPI_Q28_WAVES", waves)
        else: monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
        outs.append(d.process_host(np.ascontiguousarray(pcm[:, at * B:(at + n) * B]), n, B))
        at += n
    pairs = np.concatenate([o[0] for o in outs], axis=2); sub = np.concatenate([o[1] for o in outs], axis=1); peaks = np.concatenate([o[2] for o in outs], axis=1)
    for s_ in range(S):
        setup = (lambda o, db=own[s_]: o.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", db))) if s_ in own else None
        (rp, rs, rk, _), status = oracle_run(0, fs, -12 * 256, b, pcm[s_], total, B, 16, setup)
        assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), s_
        assert status == d.status(s_), s_
    d.close()
    monkeypatch.delenv("DSPI_Q28_LAYOUT", raising=False); monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
    compare(0, 48000, 48, 12, 150, b, calls=3)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
@pytest.mark.parametrize("lev", [1, 0])
def test_ragged_packets_delay_edges(flavor, lev):
    """44/45-frame packets (short last chunk) with the delay-line corner cases: delays shorter than a chunk (per-sample
    order), zero, and the maximum (a delay of MAX samples aliases to 0); one output disabled, one muted;
    leveller on and off (the hand-off runs in the intake wave when it is off); enough packets to wrap the lines."""
    b = WL.full_chain_blob(flavor)
    n_out = 9 if flavor else 5
    max_ms = (4096 if flavor else 2048) * 1000.0 / 44100
    delays = [0.05, 0.2, 0.3, 0.0, max_ms + 1.0, max_ms - 0.05, 1.0, 0.1, 0.25][:n_out]
    for o in range(n_out):
        b["outputs"][o]["delay_ms"] = delays[o]
    b["outputs"][1]["enabled"] = 0
    b["outputs"][2]["mute"] = 1
    b["leveller"]["enabled"] = lev
    compare(flavor, 44100, 45, 120 if flavor else 60, 20, b, calls=3)
    compare(flavor, 44100, 44, 30, 9, b, depth=24, calls=2, first_stream=14)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
@pytest.mark.parametrize("B", [1, 7, 17, 97])
def test_unusual_packet_lengths(flavor, B):
    """Packets of 1 frame (the g.B == 1 ramp, leveller.c:213), shorter than a chunk, one frame over a chunk, and the
    largest the firmware accepts (97); full chain, two launches."""
    compare(flavor, 48000, B, 200 if B < 4 else 40, 9, WL.full_chain_blob(flavor), calls=2)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_long_run_wraps_delay_lines(flavor):
    """More frames than a delay line holds (4096 float / 2048 Q28 positions): the shared write index wraps, the 80 ms /
    40 ms lines read across the wrap, and the leveller ring (1024) goes round several times; three launches."""
    fs, B = (96000, 96) if flavor else (48000, 48)
    compare(flavor, fs, B, 54, 6, WL.full_chain_blob(flavor), calls=3, first_stream=2)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_host_volume_sign_quirk_and_mute(flavor):
    compare(flavor, 96000 if flavor else 48000, 96 if flavor else 48, 12, 6, WL.full_chain_blob(flavor), vol=0)
    compare(flavor, 48000, 48, 12, 3, WL.full_chain_blob(flavor), setup=lambda x: x.set_mute(True))


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_vendor_requests_between_launches(flavor):
    """Parameter changes land on packet boundaries and carry their state side effects (filter path resets, crossfeed /
    leveller resets, preset mute envelope)."""
    fs, B = 48000, 48
    S = 4
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    steps = [
        lambda x: (x.set_rate(fs), x.set_volume(-15 * 256), x.load_bulk(WL.full_chain_blob(flavor))),
        lambda x: x.vendor_set(R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 2, W.FILTER_PEAKING, 0, 9000.0, 2.0, 6.0)),   # SVF->biquad at 48k
        lambda x: x.vendor_set(R["SET_LEVELLER_LOOKAHEAD"], 0, b"\x00"),
        lambda x: x.vendor_set(R["SET_CROSSFEED_PRESET"], 0, b"\x02"),
        lambda x: (x.vendor_set(R["SET_OUTPUT_DELAY"], 1, f(3.0)), x.vendor_set(R["SET_OUTPUT_MUTE"], 0, b"\x01")),
        lambda x: x.set_volume(-40 * 256),
        lambda x: x.vendor_set(R["SET_LEVELLER_ENABLE"], 0, b"\x00"),
        lambda x: x.factory_defaults(),
        lambda x: x.vendor_set(R["SET_MASTER_VOLUME"], 0, f(-3.0)),
        lambda x: x.load_slot(slot_image),
        # both outputs of pair 1 switched off while their delay lines still hold audio: the pair is zero-filled at once
        # (usb_audio.c:930-933), the meters still see the tail
        lambda x: (x.vendor_set(R["SET_OUTPUT_ENABLE"], 2, b"\x00"), x.vendor_set(R["SET_OUTPUT_ENABLE"], 3, b"\x00")),
        lambda x: x.vendor_set(R["SET_OUTPUT_ENABLE"], 3, b"\x01"),
        # a preset load that fails its CRC drops the mute it had armed (flash_storage.c:806) ...
        lambda x: x.load_slot(bad_image),
        # ... and a failed load followed, before the next packet, by a request that arms the mute again: the second one wins
        lambda x: (x.load_slot(bad_image), x.load_bulk(WL.full_chain_blob(flavor))),
        lambda x: (x.load_slot(bad_image), x.load_slot(slot_image)),
        # output slot 1 becomes I2S (REQ_SET_OUTPUT_TYPE, usb_audio.c:2984-3016): the switch mutes the pipeline (main.c:279)
        lambda x: set_type(x, 0x0101),
        lambda x: set_type(x, 0x0101),                              # no-op: no mute
        # a preset whose slot types differ from the live ones: the type switch re-arms the mute after the flash hold (main.c:957-972)
        lambda x: x.load_slot(slot_image),
        lambda x: x.load_slot(slot_image),                          # same types now: the flash hold stands
    ]
    set_type = lambda x, wv: x.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1, -1) if isinstance(x, Dspi) else x.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1)
    ref = Oracle(flavor); ref.load_bulk(WL.full_chain_blob(flavor)); slot_image = ref.save_slot(0)
    bad_image = bytearray(slot_image); bad_image[200] ^= 0x40; bad_image = bytes(bad_image)
    pcm = WL.synth_pcm16(S, B * 6 * len(steps), fs)
    for k, step in enumerate(steps):
        step(d)
        for oo in o: step(oo)
        chunk = np.ascontiguousarray(pcm[:, k * 6 * B:(k + 1) * 6 * B])
        pairs, sub, peaks = d.process_host(chunk, 6, B)
        for s in range(S):
            rp, rs, rk, _ = o[s].process(chunk[s], 6, B)
            assert np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), f"step {k} stream {s}"
            assert o[s].status() == d.status(s)
    d.close()


@pytest.mark.both_layouts
def test_per_stream_presets_and_clip_flags():
    """Copy-on-write images: streams with different presets in one workgroup (lane-masked launches)."""
    fs, B, S = 48000, 48, 70
    d = Dspi(1, S, device=0); o = [Oracle(1, detmath=True) for _ in range(S)]
    blob = WL.full_chain_blob(1)
    for x in [d] + o:
        x.set_rate(fs); x.set_volume(-2 * 256); x.load_bulk(blob)
    special = {3: -12.0, 19: 20.0, 64: 6.0, 69: 0.0}       # +20 dB on the full-scale square stream: guaranteed clip flags
    for s, db in special.items():
        d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", db), stream=s)
        o[s].vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", db))
    pcm = WL.synth_pcm16(S, B * 20, fs)
    pairs, sub, peaks = d.process_host(pcm, 20, B)
    for s in range(S):
        rp, rs, rk, _ = o[s].process(pcm[s], 20, B)
        assert np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s]), s
        assert o[s].status() == d.status(s)
    plan = d.launch_plan()      # 70 streams = one row of five images that differ in a preamp: the packed kernel with per-lane values, shared filters
    if on_latency_layout(): assert latency_plan(plan), plan      # (every (lane, image) pair a workgroup slot of its own, per-half stores)
    else: assert plan["packed_per_lane_values"] == 1 and plan["packed_shared"] == 0 and plan["one_stream_per_lane_images"] == 0, plan
    flags = int.from_bytes(d.status(19)[-2:], "little")          # stream class 19 = full-scale square
    assert flags != 0 and d.clear_clips(19) == flags and int.from_bytes(d.status(19)[-2:], "little") == 0
    assert int.from_bytes(d.status(18)[-2:], "little") == int.from_bytes(o[18].status()[-2:], "little")
    # DSPI_ALL_STREAMS: every stream's sticky flags go, whatever stream 0 held (it held none here)
    pairs, sub, peaks = d.process_host(pcm[:, :B * 4], 4, B)
    assert int.from_bytes(d.status(0)[-2:], "little") == 0 and int.from_bytes(d.status(19)[-2:], "little") != 0
    assert d.clear_clips() == 0
    assert all(int.from_bytes(d.status(s)[-2:], "little") == 0 for s in range(S))
    d.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_many_presets_one_process_call(flavor):
    """Dozens of parameter images in one context (every third stream its own preamp, some with the leveller off, some
    muted): launches are grouped by leveller on/off, not by image, and every stream still matches its own oracle."""
    fs, B, S, blocks = 48000, 48, 150, 8
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    blob = WL.full_chain_blob(flavor)
    for x in [d] + o:
        x.set_rate(fs); x.set_volume(-8 * 256); x.load_bulk(blob)
    for s_ in range(0, S, 3):
        db = struct.pack("<f", -12.0 + 0.25 * s_)
        d.vendor_set(W.REQ["SET_PREAMP"], 0, db, stream=s_); o[s_].vendor_set(W.REQ["SET_PREAMP"], 0, db)
        if s_ % 2 == 0:
            d.vendor_set(W.REQ["SET_LEVELLER_ENABLE"], 0, b"\x00", stream=s_); o[s_].vendor_set(W.REQ["SET_LEVELLER_ENABLE"], 0, b"\x00")
        if s_ % 5 == 0:
            d.vendor_set(W.REQ["SET_OUTPUT_MUTE"], 1, b"\x01", stream=s_); o[s_].vendor_set(W.REQ["SET_OUTPUT_MUTE"], 1, b"\x01")
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
    for c in range(2):
        chunk = np.ascontiguousarray(pcm[:, c * blocks * B:(c + 1) * blocks * B])
        pairs, sub, peaks = d.process_host(chunk, blocks, B)
        for s_ in range(S):
            rp, rs, rk, _ = o[s_].process(chunk[s_], blocks, B)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (c, s_)
            assert o[s_].status() == d.status(s_)
    d.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("fs,B,depth,S,lev", [(96000, 96, 16, 300, 1), (44100, 45, 24, 131, 1), (48000, 48, 16, 140, 0), (44100, 44, 16, 70, 0)])
def test_one_structure_different_numbers(fma, fs, B, depth, S, lev):
    """SURVEY §8f-1, the product-family case: every stream its own preset, all of ONE structure (same filter types, bypasses, routing
    pattern, delays) with different numbers everywhere a number can differ without changing the structure — every band's gain and Q,
    preamps, master volume, crosspoint gains, output gains, leveller amount / speed / gate / max gain, custom crossfeed — plus
    per-stream UAC1 volume and mute.  Such rows run the packed kernel with per-lane values (value tiles, dspi_image.h); the odd
    last stream of the 131-stream case and a stream of a different structure dropped into row 0 take the one-stream kernel.
    Numbers change again between the two calls (the tiles are rebuilt); every stream must match its own oracle."""
    blocks = 6
    flavor = W.F32_FMA if fma else 1
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    blob = WL.full_chain_blob(1)
    blob["leveller"]["enabled"] = lev
    for x in [d] + o:
        x.set_rate(fs); x.set_volume(-9 * 256); assert x.load_bulk(blob) == 0
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    rng = np.random.default_rng(23 + S)
    recipes = {}

    def numbers(s_, call):
        reqs = [(R["SET_PREAMP_CH"], 0, f(-15.0 + 0.02 * s_)), (R["SET_PREAMP_CH"], 1, f(-14.0 - 0.01 * s_)), (R["SET_MASTER_VOLUME"], 0, f(-0.05 * (s_ % 60) - call))]
        for _ in range(6):        # same type and frequency (the kind depends on both), new gain and Q
            ch, band = int(rng.integers(0, 11)), int(rng.integers(0, 10))
            p = blob["eq"][ch][band]
            if int(p["type"]) == W.FILTER_FLAT: continue
            reqs.append((R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, band, int(p["type"]), 0, float(p["freq"]), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 9.0)) * (1.0 if rng.random() < 0.5 else -1.0))))      # |gain| < 0.01 dB would make the band flat: another structure
        o_ = int(rng.integers(0, 9))
        reqs.append((R["SET_OUTPUT_GAIN"], o_, f(float(rng.uniform(-12, 3)))))
        for i_ in range(2):       # a routed crosspoint keeps a non-zero gain (its zero pattern is structure)
            o2 = int(rng.integers(0, 9))
            xp = blob["crosspoints"][i_][o2]
            if int(xp["enabled"]): reqs.append((R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", i_, o2, 1, int(xp["phase_invert"]), float(rng.uniform(-9, 0)))))
        reqs += [(R["SET_LEVELLER_AMOUNT"], 0, f(float(rng.uniform(10, 100)))), (R["SET_LEVELLER_SPEED"], 0, bytes([int(rng.integers(0, 3))])),
                 (R["SET_LEVELLER_MAX_GAIN"], 0, f(float(rng.uniform(3, 20)))), (R["SET_LEVELLER_GATE"], 0, f(float(rng.uniform(-90, -50))))]
        if int(blob["crossfeed"]["enabled"]):
            reqs += [(R["SET_CROSSFEED_PRESET"], 0, b"\x03"), (R["SET_CROSSFEED_FREQ"], 0, f(float(rng.uniform(500, 1500)))), (R["SET_CROSSFEED_FEED"], 0, f(float(rng.uniform(3, 12))))]
        return reqs

    pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = B * blocks * (1 if depth == 16 else 6)
    for call in range(2):
        for s_ in range(S):
            if call == 1 and s_ % 3: continue
            for req, wv, pl in numbers(s_, call):
                assert d.vendor_set(req, wv, pl, stream=s_) == 0 and o[s_].vendor_set(req, wv, pl) == 0
            if s_ % 11 == 5:
                v = -256 * int(rng.integers(0, 40)); d.set_volume(v, stream=s_); o[s_].set_volume(v)
            if s_ % 37 == 9:
                d.set_mute(call == 0, stream=s_); o[s_].set_mute(call == 0)
        if call == 0:             # one stream of another structure in the middle of row 0: its lane leaves the packed kernel
            for x in (lambda *a: d.vendor_set(*a, stream=17), o[17].vendor_set):
                assert x(R["SET_OUTPUT_MUTE"], 2, b"\x01") == 0
        chunk = np.ascontiguousarray(data[:, call * per:(call + 1) * per])
        pairs, sub, peaks = d.process_host(chunk, blocks, B, depth)
        plan = d.launch_plan()
        rows = (S + 127) // 128
        # row 0 holds the stream of another structure: it runs on the one-stream kernel; every other row is a per-lane-value row with
        # per-lane filters; an odd last stream adds one more one-stream item
        # ... on the latency layout: paired presets — a workgroup's stream slots each read their own image (two pairs per workgroup here:
        # a workgroup per four streams, plus the ones the stream of another structure splits)
        # (not all of them: a UAC1 volume can switch a loudness shelf on or off, which is structure — such a workgroup runs once per image)
        if on_latency_layout(): assert latency_plan(plan) and plan["latency_layout_paired"] >= (S + 3) // 4 * 3 // 4, plan
        else: assert plan["packed_per_lane_values_and_bands"] == rows - 1 and plan["one_stream_per_lane_images"] >= 1 and plan["packed_shared"] == 0, plan
        for s_ in range(S):
            rp, rs, rk, _ = o[s_].process(chunk[s_], blocks, B, depth)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (call, s_)
            assert o[s_].status() == d.status(s_)
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
@pytest.mark.parametrize("shape,fs,B,depth,S", [(1, 48000, 48, 16, 70), (1, 44100, 45, 24, 37), (2, 96000, 96, 16, 70), (3, 96000, 96, 16, 70), (3, 48000, 7, 24, 21), (1, 48000, 1, 16, 33)])
def test_latency_layout_paired_presets(flavor, shape, fs, B, depth, S, monkeypatch):
    """Verdict r03 item 7: a small context in which every stream has a preset of its own, all of one structure.  The latency layout used to
    give each of them a workgroup (the other stream slots idle); now a workgroup's slots each read their own image (dspi_chain_skew.inc
    SkNum): sixteen (shape 1) or four (shapes 2, 3) presets per workgroup.  All three shapes, numbers that differ wherever the shape reads
    one (master bands, preamps, volumes, crosspoint and output gains, output bands, leveller and crossfeed constants), against every
    stream's own oracle over two calls with changes in between; DSPI_SKEW_PAIRED=0 (one workgroup per preset, the old form) must give
    the same words."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    blocks = 5
    blob = _latency_blob() if shape == 1 else WL.full_chain_blob(1)
    blob["leveller"]["enabled"] = 1 if shape == 3 else 0
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
    data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    per = B * blocks * (1 if depth == 16 else 6)

    def numbers(rng, s_, call):
        reqs = [(R["SET_PREAMP_CH"], 0, f(-15.0 + 0.05 * s_)), (R["SET_PREAMP_CH"], 1, f(-14.0 - 0.03 * s_)), (R["SET_MASTER_VOLUME"], 0, f(-0.1 * (s_ % 40) - call))]
        for _ in range(8):        # same type and frequency (the kind depends on both), new gain and Q
            ch, band = int(rng.integers(0, 2 if shape == 1 else 11)), int(rng.integers(0, 10))
            p = blob["eq"][ch][band]
            if int(p["type"]) == W.FILTER_FLAT: continue
            reqs.append((R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, band, int(p["type"]), 0, float(p["freq"]), float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 9.0)) * (1.0 if rng.random() < 0.5 else -1.0))))
        reqs.append((R["SET_OUTPUT_GAIN"], int(rng.integers(0, 9)), f(float(rng.uniform(-12, 3)))))
        for i_ in range(2):       # a routed crosspoint keeps a non-zero gain (its zero pattern is structure)
            o2 = int(rng.integers(0, 9))
            xp = blob["crosspoints"][i_][o2]
            if int(xp["enabled"]): reqs.append((R["SET_MATRIX_ROUTE"], 0, struct.pack("<BBBBf", i_, o2, 1, int(xp["phase_invert"]), float(rng.uniform(-9, 0)))))
        if shape == 3:
            reqs += [(R["SET_LEVELLER_AMOUNT"], 0, f(float(rng.uniform(10, 100)))), (R["SET_LEVELLER_SPEED"], 0, bytes([int(rng.integers(0, 3))])),
                     (R["SET_LEVELLER_MAX_GAIN"], 0, f(float(rng.uniform(3, 20)))), (R["SET_LEVELLER_GATE"], 0, f(float(rng.uniform(-90, -50))))]
        if int(blob["crossfeed"]["enabled"]):
            reqs += [(R["SET_CROSSFEED_PRESET"], 0, b"\x03"), (R["SET_CROSSFEED_FREQ"], 0, f(float(rng.uniform(500, 1500)))), (R["SET_CROSSFEED_FEED"], 0, f(float(rng.uniform(3, 12))))]
        return reqs

    results = {}
    for paired in ("1", "0"):
        monkeypatch.setenv("DSPI_SKEW_PAIRED", paired)
        rng = np.random.default_rng(5 + S)
        d = Dspi(flavor, S, device=0)
        o = [Oracle(flavor, detmath=True) for _ in range(S)] if paired == "1" else []
        for x in [d] + o:
            x.set_rate(fs); x.set_volume(-9 * 256); assert x.load_bulk(blob) == 0
        out = []
        for call in range(2):
            for s_ in range(S):
                if call == 1 and s_ % 3: continue
                for req, wv, pl in numbers(rng, s_, call):
                    assert d.vendor_set(req, wv, pl, stream=s_) == 0
                    if o: assert o[s_].vendor_set(req, wv, pl) == 0
            chunk = np.ascontiguousarray(data[:, call * per:(call + 1) * per])
            pairs, sub, peaks = d.process_host(chunk, blocks, B, depth, clip=True)
            plan = d.launch_plan()
            ppw = 16 if shape == 1 else 4
            wgs = sum((min(S, r + 128) - r + ppw - 1) // ppw for r in range(0, S, 128))
            alone = 1 if S % ppw == 1 else 0        # a last workgroup that holds ONE stream is a shared-preset item
            assert latency_plan(plan), plan
            if paired == "1": assert plan["latency_layout"] == wgs and plan["latency_layout_paired"] == wgs - alone, (plan, wgs)
            else: assert plan["latency_layout_paired"] == 0 and plan["latency_layout"] == S, plan
            assert d.image_count() == S
            for s_ in range(S if o else 0):
                rp, rs, rk, _ = o[s_].process(chunk[s_], blocks, B, depth)
                assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (call, s_)
                assert o[s_].status() == d.status(s_)
            out.append((pairs.copy(), sub.copy(), peaks.copy(), d.last_clip.copy()))
        results[paired] = out
        d.close()
    for a_, b_ in zip(results["1"], results["0"]):
        for x_, y_ in zip(a_, b_): assert np.array_equal(x_, y_)


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor,fs,B,depth,S", [(1, 48000, 48, 16, 200), (1, 44100, 45, 24, 70), (0, 48000, 48, 16, 150), (0, 44100, 44, 24, 70)])
def test_every_stream_its_own_preset(flavor, fs, B, depth, S):
    """every stream (both flavours) carries a different preset — different band kinds at the same band index (SVF
    forms vs biquad, bypassed), leveller / crossfeed / loudness on or off, muted and disabled outputs, delays from 0 to
    the alias value, different preamps.  The one-stream float kernel reads per-lane parameter images, so this is two
    launches, not one per preset; every stream must still match its own oracle, across two calls."""
    blocks = 8
    d = Dspi(flavor, S, device=0); o = [Oracle(flavor, detmath=True) for _ in range(S)]
    blob = WL.full_chain_blob(flavor)
    C_, N_ = (11, 9) if flavor else (7, 5)
    for x in [d] + o:
        x.set_rate(fs); x.set_volume(-9 * 256); x.load_bulk(blob)
    R = W.REQ
    f = lambda v: struct.pack("<f", v)
    rng = np.random.default_rng(11)
    types = [W.FILTER_PEAKING, W.FILTER_LOWSHELF, W.FILTER_HIGHSHELF, W.FILTER_LOWPASS, W.FILTER_HIGHPASS, W.FILTER_FLAT]
    for s_ in range(S):
        reqs = [(R["SET_PREAMP"], 0, f(-15.0 + 0.1 * s_))]
        ch, band = int(rng.integers(0, C_)), int(rng.integers(0, 10))
        freq = float(rng.choice([60.0, 400.0, 3000.0, 9000.0, 15000.0]))          # < 6.4 kHz: SVF, above: biquad (48 kHz)
        reqs.append((R["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, band, types[s_ % len(types)], 0, freq, 0.9, float(rng.uniform(-6, 6)))))
        if s_ % 3 == 0: reqs.append((R["SET_LEVELLER_ENABLE"], 0, b"\x00"))
        if s_ % 4 == 1: reqs.append((R["SET_CROSSFEED"], 0, b"\x00"))
        if s_ % 5 == 2: reqs.append((R["SET_OUTPUT_MUTE"], int(rng.integers(0, N_)), b"\x01"))
        if s_ % 7 == 3: reqs.append((R["SET_OUTPUT_ENABLE"], int(rng.integers(2, N_ - 1)), b"\x00"))
        reqs.append((R["SET_OUTPUT_DELAY"], int(rng.integers(0, N_ - 1)), f(float(rng.choice([0.0, 0.1, 0.3, 7.7, 42.0 if not flavor else 85.0])))))
        for req, wv, pl in reqs:
            assert d.vendor_set(req, wv, pl, stream=s_) == o[s_].vendor_set(req, wv, pl)
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
    for c in range(2):
        chunk = np.ascontiguousarray(pcm[:, c * blocks * B:(c + 1) * blocks * B])
        data = chunk if depth == 16 else WL.pcm16_to_pcm24_bytes(chunk)
        pairs, sub, peaks = d.process_host(data, blocks, B, depth)
        for s_ in range(S):
            rp, rs, rk, _ = o[s_].process(data[s_], blocks, B, depth)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (c, s_)
            assert o[s_].status() == d.status(s_)
    # the same whole state for everybody again (factory reset, then one blob, broadcast): equal parameter objects fold back into one
    # image, the rows return to the shared-parameter kernels — with the preset mutes, state resets and the delay lines of each
    # stream's own history
    plan = d.launch_plan()
    if flavor and on_latency_layout(): assert latency_plan(plan), plan
    else: assert plan["packed_shared"] == 0 and plan["q28_shared"] == 0 and plan["one_stream_per_lane_images"] > 0, plan
    blob2 = WL.full_chain_blob(flavor); blob2["preamp"]["preamp_db"][0] = -4.5
    assert d.image_count() == S
    d.factory_defaults(); assert d.load_bulk(blob2) == 0
    assert d.image_count() == 1
    for x in o:
        x.factory_defaults(); assert x.load_bulk(blob2) == 0
    chunk = np.ascontiguousarray(pcm[:, :blocks * B])
    data = chunk if depth == 16 else WL.pcm16_to_pcm24_bytes(chunk)
    pairs, sub, peaks = d.process_host(data, blocks, B, depth)
    plan = d.launch_plan()
    rows = (S + (127 if flavor else 63)) // (128 if flavor else 64)
    if flavor and on_latency_layout(): assert latency_plan(plan), plan
    elif flavor: assert plan["packed_shared"] == rows and plan["packed_per_lane_values_and_bands"] == plan["packed_per_lane_values"] == 0 and plan["one_stream_per_lane_images"] == S % 2, plan
    else: assert plan["q28_shared"] == rows, plan
    for s_ in range(S):
        rp, rs, rk, _ = o[s_].process(data[s_], blocks, B, depth)
        assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), ("folded", s_)
        assert o[s_].status() == d.status(s_)
    d.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_tiled_output_layout(flavor):
    """DSPI_OUT_TILED ([tile][output][frame][R]) carries exactly the words of the stream-major layout: packed and
    one-stream kernels (two streams get their own preset), ragged last tile, tail packets on the second call."""
    fs, S = 48000, (200 if flavor else 100)
    blob = WL.full_chain_blob(flavor)
    ctx = []
    for _ in range(2):
        d = Dspi(flavor, S, device=0)
        d.set_rate(fs); d.set_volume(-12 * 256); assert d.load_bulk(blob) == 0
        for s_, db in ((5, -9.0), (130 % S, 3.0)):
            d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", db), stream=s_)
        ctx.append(d)
    assert ctx[0].tile_streams() == (128 if flavor else 64)
    for B, blocks in ((48, 10), (45, 7)):
        pcm = WL.synth_pcm16(S, B * blocks, fs)
        p0, s0, k0 = ctx[0].process_host(pcm, blocks, B)
        pt, st, k1 = ctx[1].process_host(pcm, blocks, B, tiled=True)
        p1, s1 = ctx[1].untile(pt, st)
        assert np.array_equal(p0, p1) and np.array_equal(s0, s1) and np.array_equal(k0, k1), (B, blocks)
    for d in ctx: d.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures_on_gpu(path):
    """Vectors generated from the reference's own leaf sources (tests/golden/make_golden.py).  The one Q28 vector that encodes the x86 cast
    artefact (see test_oracle_golden) is held to the golden up to the frame where the artefact starts, and to the oracle throughout."""
    g = np.load(path)
    flavor = int(g["flavor"])
    d = Dspi(flavor, 1, device=0, fma=bool(int(g["fma"])) if "fma" in g else False)      # f32fma_*: the reference compiled with GCC's contraction
    d.set_rate(int(g["fs"])); d.set_volume(int(g["volume"])); assert d.load_bulk(g["blob"].tobytes()) == 0
    data = g["pcm"][None]
    pairs, sub, peaks = d.process_host(np.ascontiguousarray(data), int(g["blocks"]), int(g["block_len"]), int(g["bit_depth"]))
    if "q28_full_48k_detmath" in path:
        # This vector drives the Q28 limiter's cast out of range (leveller.c:376) from frame 497 on, where the reference's x86 build yields
        # INT_MIN and the MCUs saturate (tests/test_oracle_golden.py::test_q28_limiter_cast_is_the_documented_divergence): its CRCs hold the
        # x86 artefact.  What the reference's object code pins here is everything BEFORE that frame — the golden's first 96 frames of words
        # and its first ten packets of meters, stored as they came out of the reference —; the whole vector is held to the firmware's
        # (saturating) semantics through the oracle.
        assert np.array_equal(pairs[0][:, :96], g["pairs_head"]) and np.array_equal(sub[0][:96], g["sub_head"])
        assert np.array_equal(peaks[0][:10], g["peaks"][:10])
        o = Oracle(flavor, detmath=True)
        assert o.set_rate(int(g["fs"])) == 0
        o.set_volume(int(g["volume"])); assert o.load_bulk(g["blob"].tobytes()) == 0
        rp, rs, rk, _ = o.process(g["pcm"], int(g["blocks"]), int(g["block_len"]), int(g["bit_depth"]))
        assert np.array_equal(pairs[0], rp) and np.array_equal(sub[0], rs) and np.array_equal(peaks[0], rk)
        assert not np.array_equal(pairs[0][:, -96:], g["pairs_tail"])      # (and the tail is where the two part ways)
        d.close()
        return
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
    assert crc(pairs[0]) == int(g["pairs_crc"]) and crc(sub[0]) == int(g["sub_crc"]) and crc(peaks[0]) == int(g["peaks_crc"])
    assert list(d.status(0)) == g["status"].tolist()
    d.close()


@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
def test_full_size_properties(flavor):
    """BASELINE config 3 at full size (65 536 streams), stream-major words, both float contracts: (i) streams are independent and
    placement-invariant — a stream gives the same words whatever its lane/workgroup; (ii) identical inputs give identical outputs in
    every lane; (iii) a sample of streams is bit-exact against the oracle; (iv) linear-phase sanity: digital silence in -> silence out."""
    import torch
    fs, B, blocks, S = 96000, 96, 4, 65536
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(1)) == 0
    base = WL.synth_pcm16(128, B * blocks, fs)
    idx = np.arange(S) % 128
    idx[1000] = 5; idx[40000] = 5; idx[65535] = 5          # the same stream content in far-apart lanes
    dev = torch.device("cuda", 0)
    pcm = torch.from_numpy(base).to(dev)[torch.from_numpy(idx).to(dev)].contiguous()
    pcm[7777] = 0
    frames = B * blocks
    pairs = torch.empty((S, 4, frames, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), 0); d.sync()
    assert torch.equal(pairs[1000], pairs[5]) and torch.equal(pairs[40000], pairs[5]) and torch.equal(pairs[65535], pairs[5])
    assert torch.equal(pairs[5 + 128 * 37], pairs[5]) and torch.equal(sub[5 + 128 * 400], sub[5])
    assert int(pairs[7777].abs().max()) == 0 and int(sub[7777].abs().max()) == 0
    for s in (0, 63, 64, 4095, 65471):
        (rp, rs, _, _), _ = oracle_run(flavor, fs, -20 * 256, WL.full_chain_blob(1), base[idx[s]], blocks, B, 16)
        assert np.array_equal(rp, pairs[s].cpu().numpy()) and np.array_equal(rs, sub[s].cpu().numpy())
    d.close()


@pytest.mark.parametrize("flavor,fs,B,S,tiled", [(W.F32_FMA, 96000, 96, 65536, False), (W.F32_FMA, 96000, 96, 65536, True), (1, 96000, 96, 65536, False), (0, 48000, 48, 16384, False)])
def test_full_size_distinct_input_sampled_across_every_workgroup(flavor, fs, B, S, tiled, monkeypatch):
    """the full-size checks above feed every workgroup the SAME streams, so a defect that needs distinct data in
    distinct workgroups under full occupancy (an address carry past 4 GB in the 9.66 GB line array with per-row content, a row reading
    its neighbour's ring) would be seen by a handful of streams only.  Here: BASELINE config 3 (Q28: config 5) at its own size, the
    SURVEY 8d input mix with DISTINCT noise per stream (generated on the device), three launches of 14 packets — shorter than four of the
    delays, so every launch reads what the one before left in the lines — and ONE stream of EVERY workgroup row (lane and lane half
    varying from row to row: 512 rows float, 256 Q28; the rows from 228 on lie beyond 4 GB in the line array) plus one stream of every
    input class checked against the oracle: every pair word, sub word and peak of every launch, status bytes and clip flags at the end."""
    import importlib.util
    import torch
    from concurrent.futures import ThreadPoolExecutor
    if S == 16384: monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
    spec = importlib.util.spec_from_file_location("dspi_bench_for_tests", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    fl = int(flavor)
    blocks, calls = 14, 3
    n_out, n_ch, P = (9, 11, 4) if fl else (5, 7, 2)
    dev = torch.device("cuda", 0)
    blob = WL.full_chain_blob(fl)
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
    R = d.tile_streams(); rows = S // R
    frames = B * blocks
    if fl:      # the lines of the last rows start beyond 4 GB (9 outputs x 4 096 positions x 128 streams x 4 B per row)
        assert (rows - 1) * 9 * 4096 * R * 4 > (1 << 32)
    sample = sorted({r * R + (37 * r + 5) % R for r in range(rows)} | {14, 15, 16, 17, 18, 19, S // 2 // 20 * 20 + 18, S // 2 // 20 * 20 + 19, S - 1})
    idx = torch.tensor(sample, device=dev)
    got = []
    clip = torch.empty((S,), dtype=torch.int16, device=dev)
    pcm_all = bench.synth_device(torch, dev, S, frames * calls, fs, 4321, True)      # one buffer over the three launches: the classes (sweep, bursts, square) run through
    host_pcm = pcm_all[idx].cpu().numpy()
    for c in range(calls):
        pcm = pcm_all[:, c * frames:(c + 1) * frames].contiguous()
        if tiled:
            pairs = torch.empty((rows, n_out - 1, frames, R), dtype=torch.int32, device=dev); sub = torch.empty((rows, frames, R), dtype=torch.int32, device=dev)
        else:
            pairs = torch.empty((S, P, frames, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
        peaks = torch.empty((S, blocks, n_ch), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=tiled, clip_ptr=clip.data_ptr()); d.sync()
        if tiled:
            gp = pairs[idx // R, :, :, idx % R].cpu().numpy().reshape(len(sample), P, 2, frames).transpose(0, 1, 3, 2)
            gs = sub[idx // R, :, idx % R].cpu().numpy()
        else:
            gp, gs = pairs[idx].cpu().numpy(), sub[idx].cpu().numpy()
        got.append((gp, gs, peaks[idx].cpu().numpy().view(np.uint16)))
        del pairs, sub, peaks, pcm
    clip_h = clip[idx].cpu().numpy().view(np.uint16)
    status = [d.status(s) for s in sample]

    def one(j):
        o = Oracle(flavor, detmath=True)
        assert o.set_rate(fs) == 0
        o.set_volume(-20 * 256)
        assert o.load_bulk(blob) == 0
        res = [o.process(np.ascontiguousarray(host_pcm[j][c * frames:(c + 1) * frames]), blocks, B, 16) for c in range(calls)]
        st = o.status(); o.close()
        return res, st
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        refs = list(ex.map(one, range(len(sample))))
    for j, s in enumerate(sample):
        res, st = refs[j]
        for c in range(calls):
            rp, rs, rk, rclip = res[c]
            assert np.array_equal(rp, got[c][0][j]), f"pair words differ: stream {s} (row {s // R}), launch {c}: {np.argwhere(rp != got[c][0][j])[:3].tolist()}"
            assert np.array_equal(rs, got[c][1][j]), f"sub words differ: stream {s} (row {s // R}), launch {c}"
            assert np.array_equal(rk, got[c][2][j]), f"peaks differ: stream {s} (row {s // R}), launch {c}"
        assert st == status[j], f"status bytes differ: stream {s}"
        assert int(clip_h[j]) == int.from_bytes(st[-2:], "little") == res[-1][3], f"clip flags differ: stream {s}"
    assert len(sample) >= (512 if fl else 256)
    d.close()


@pytest.mark.parametrize("flavor,fs,B,S", [(1, 96000, 96, 65536), (1, 44100, 45, 65536), (1, 48000, 16, 65536), (W.F32_FMA, 96000, 96, 65536), (W.F32_FMA, 44100, 45, 65536),
                                          (0, 48000, 48, 65536), (0, 44100, 44, 65536), (0, 48000, 16, 65536), (0, 48000, 20, 65536), (0, 48000, 7, 65536),
                                          (0, 48000, 48, 16384), (0, 44100, 45, 16384)])
def test_full_size_all_tiles_agree(flavor, fs, B, S, monkeypatch):
    """Race detector at full size on the path the bench times (tiled words, device buffers, several launches): every
    tile gets the same streams of input, so every tile must produce the words of tile 0 — which is checked against the
    oracle on a few streams.  A stale ring row, a lost barrier or a role mix-up in any workgroup shows up here (float:
    512 workgroups of twelve waves; Q28: 1024 workgroups whose two master waves meet in the ring)."""
    import torch
    # (Q28 at 16 384 streams = BASELINE config 5's own size: 256 rows, the seven-wave layout on every CU — the size rule must pick it)
    if S == 16384: monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
    blocks, calls = (14 if B > 20 else 60), 3
    n_out, n_ch = (9, 11) if flavor else (5, 7)
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(flavor)) == 0
    R = d.tile_streams()
    base = WL.synth_pcm16(R, B * blocks * calls, fs)
    dev = torch.device("cuda", 0)
    tiles = S // R
    frames = B * blocks
    pairs = torch.empty((tiles, n_out - 1, frames, R), dtype=torch.int32, device=dev); sub = torch.empty((tiles, frames, R), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, blocks, n_ch), dtype=torch.int16, device=dev)
    for c in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat(tiles, 1, 1).contiguous()
        torch.cuda.synchronize()        # the context launches on its own stream: the input must be complete before it starts
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True); d.sync()
        assert bool((pairs == pairs[0:1]).all()), f"launch {c}: a tile's pair words differ from tile 0"
        assert bool((sub == sub[0:1]).all()), f"launch {c}: a tile's sub words differ from tile 0"
        pk = peaks.view(tiles, R, blocks, n_ch)
        assert bool((pk == pk[0:1]).all()), f"launch {c}: a tile's peaks differ from tile 0"
    p0 = pairs[0].cpu().numpy(); s0 = sub[0].cpu().numpy()
    for s in (0, 1, R // 2, R - 1):
        (rp, rs, _, _), _ = oracle_run(flavor, fs, -20 * 256, WL.full_chain_blob(flavor), base[s], blocks * calls, B, 16)
        last = rp[:, (calls - 1) * frames:, :]                      # [pair][frame][side] of the last launch
        got = np.stack([np.stack([p0[2 * p, :, s], p0[2 * p + 1, :, s]], axis=-1) for p in range((n_out - 1) // 2)])
        assert np.array_equal(last, got) and np.array_equal(rs[(calls - 1) * frames:], s0[:, s])
    d.close()


@pytest.mark.parametrize("flavor,fs,B,blocks", [(W.F32_FMA, 96000, 96, 14), (1, 96000, 96, 14), (W.F32_FMA, 44100, 45, 15), (W.F32_FMA, 48000, 48, 50)],
                         ids=("fma-96k", "canonical-96k", "fma-44k1-ragged", "fma-48k-50pk"))
def test_full_size_stream_major_groups_agree(flavor, fs, B, blocks):
    """Race detector on the variant bench.py times by default: firmware float contract, STREAM-MAJOR words (the firmware's
    [stream][pair][frame][2] buffers, written as whole lines), device buffers, three launches.  Every 128-stream group gets the
    same input, so every group must produce group 0's words, sub words and peaks; group 0 is checked against the oracle."""
    import torch
    S, calls, R = 65536, 3, 128
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(1)) == 0
    base = WL.synth_pcm16(R, B * blocks * calls, fs)
    dev = torch.device("cuda", 0)
    groups, frames = S // R, B * blocks
    pairs = torch.empty((S, 4, frames, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, blocks, 11), dtype=torch.int16, device=dev)
    for c in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat(groups, 1, 1).contiguous()
        pairs.fill_(0x55555555); sub.fill_(0x55555555)            # a word nobody wrote shows up
        torch.cuda.synchronize()
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()); d.sync()
        pv, sv, kv = pairs.view(groups, R, 4, frames, 2), sub.view(groups, R, frames), peaks.view(groups, R, blocks, 11)
        assert bool((pv == pv[0:1]).all()), f"launch {c}: a group's pair words differ from group 0"
        assert bool((sv == sv[0:1]).all()), f"launch {c}: a group's sub words differ from group 0"
        assert bool((kv == kv[0:1]).all()), f"launch {c}: a group's peaks differ from group 0"
    p0, s0 = pairs[:R].cpu().numpy(), sub[:R].cpu().numpy()
    for s in (0, 1, 63, 64, R - 1):
        (rp, rs, _, _), _ = oracle_run(flavor, fs, -20 * 256, WL.full_chain_blob(1), base[s], blocks * calls, B, 16)
        assert np.array_equal(rp[:, (calls - 1) * frames:, :], p0[s]) and np.array_equal(rs[(calls - 1) * frames:], s0[s]), s
    d.close()


@pytest.mark.parametrize("flavor,S,bands,tiled", [(W.F32_FMA, 65536, False, False), (W.F32_FMA, 65536, False, True), (W.F32_FMA, 65536, True, False), (W.F32_FMA, 65536, True, True),
                                                  (1, 65536, True, True), (0, 16384, True, False), (0, 16384, True, True)],
                         ids=("fma-values-stream", "fma-values-tiled", "fma-bands-stream", "fma-bands-tiled", "canonical-bands-tiled", "q28-16384-stream", "q28-16384-tiled"))
def test_full_size_per_stream_presets(flavor, S, bands, tiled, monkeypatch):
    """SURVEY 8f-1 at the size bench.py --config perstream / perstream_eq quotes: EVERY stream its own parameter image (65 536 float /
    16 384 Q28 distinct objects, dsp_compute_coefficients per device: dsp_pipeline.c:61-175), the numbers taken from a table that repeats
    every row (preamp per stream; `bands`: a master EQ band's gain per stream too, so the band coefficients differ and the float rows take
    chain_kernel_pk<..., PV, PVB> instead of <..., PV>; Q28: the per-lane-parameter mode of every row).  Every row gets the same input, so
    every row must produce row 0's words, sub words and peaks over three launches — value tiles, per-row launch lists and the image table
    at 512 / 256 workgroups — and row 0's sampled streams are checked against their own oracles."""
    import torch
    fl = int(flavor)
    if not fl: monkeypatch.delenv("DSPI_Q28_WAVES", raising=False)
    fs, B, blocks, calls = (96000, 96, 6, 3) if fl else (48000, 48, 10, 3)
    n_out, n_ch, n_pairs = (9, 11, 4) if fl else (5, 7, 2)
    blob = WL.full_chain_blob(fl)
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
    R = d.tile_streams()

    def requests(col):
        reqs = [(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -9.0 + 0.05 * col))]
        if bands:
            p = blob["eq"][0][1]
            reqs.append((W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 1, int(p["type"]), 0, float(p["freq"]), float(p["q"]), 1.0 + 0.03 * col)))
        return reqs
    table = [requests(col) for col in range(R)]
    for s in range(S):
        for req, wv, pl in table[s % R]: assert d.vendor_set(req, wv, pl, stream=s) == 0
    assert d.image_count() == S
    base = WL.synth_pcm16(R, B * blocks * calls, fs)
    dev = torch.device("cuda", 0)
    groups, frames = S // R, B * blocks
    if tiled: pairs = torch.empty((groups, n_out - 1, frames, R), dtype=torch.int32, device=dev); sub = torch.empty((groups, frames, R), dtype=torch.int32, device=dev)
    else: pairs = torch.empty((S, n_pairs, frames, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, blocks, n_ch), dtype=torch.int16, device=dev)
    for c in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat(groups, 1, 1).contiguous()
        pairs.fill_(0x55555555); sub.fill_(0x55555555)
        torch.cuda.synchronize()
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=tiled); d.sync()
        pv = pairs if tiled else pairs.view(groups, R, n_pairs, frames, 2)
        sv = sub if tiled else sub.view(groups, R, frames)
        kv = peaks.view(groups, R, blocks, n_ch)
        assert bool((pv == pv[0:1]).all()), f"launch {c}: a row's pair words differ from row 0"
        assert bool((sv == sv[0:1]).all()), f"launch {c}: a row's sub words differ from row 0"
        assert bool((kv == kv[0:1]).all()), f"launch {c}: a row's peaks differ from row 0"
    plan = d.launch_plan()
    if fl: assert plan["packed_per_lane_values_and_bands" if bands else "packed_per_lane_values"] == groups and plan["packed_shared"] == plan["one_stream_per_lane_images"] == plan["latency_layout"] == 0, plan
    else: assert plan["q28_shared"] == 0 and plan["one_stream_per_lane_images"] == groups, plan
    if tiled:
        p0 = pairs[0].cpu().numpy(); s0 = sub[0].cpu().numpy()
    else:
        p0, s0 = pairs[:R].cpu().numpy(), sub[:R].cpu().numpy()
    k0 = peaks[:R].cpu().numpy().view(np.uint16)
    for s in (0, 1, R // 2 - 1, R // 2, R - 1):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-20 * 256); assert o.load_bulk(blob) == 0
        for req, wv, pl in table[s]: assert o.vendor_set(req, wv, pl) == 0
        rp, rs, rk, _ = o.process(base[s], blocks * calls, B, 16)
        last = rp[:, (calls - 1) * frames:, :]
        got = np.stack([np.stack([p0[2 * p, :, s], p0[2 * p + 1, :, s]], axis=-1) for p in range(n_pairs)]) if tiled else p0[s]
        assert np.array_equal(last, got), (s, np.argwhere(last != got)[:3].tolist())
        assert np.array_equal(rs[(calls - 1) * frames:], s0[:, s] if tiled else s0[s]), s
        assert np.array_equal(rk[(calls - 1) * blocks:], k0[s]), s
        for far in (s, s + R * (groups // 2), s + R * (groups - 1)): assert o.status() == d.status(far), far
    d.close()


@pytest.mark.auto_layout
@pytest.mark.parametrize("two_b", (False, True), ids=("config2", "config2b"))
def test_full_size_config2_latency_layout(two_b):
    """BASELINE config 2 at the size bench.py --config 2 times it: 4 096 streams (one workgroup of eight stream pairs on every CU), 48-frame
    packets, the firmware contract, DSPI_OUT_ENABLED_ONLY — taken to the latency layout by the library's own size rule (nothing forced),
    three launches of 200 packets.  Every workgroup (16 streams) gets the same input: all must produce workgroup 0's live words and
    peaks, the silent pairs and the sub must stay unwritten, and workgroup 0 is checked against the oracle over all three launches."""
    import torch
    fs, B, blocks, calls, S, R = 48000, 48, 200, 3, 4096, 16
    blob = WL.config2_blob(two_b)
    d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-10 * 256); assert d.load_bulk(blob) == 0
    base = WL.synth_pcm16(R, B * blocks * calls, fs)
    dev = torch.device("cuda", 0)
    groups, frames = S // R, B * blocks
    pairs = torch.empty((S, 4, frames, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, blocks, 11), dtype=torch.int16, device=dev)
    got = []
    for c in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat(groups, 1, 1).contiguous()
        pairs.fill_(0x55555555); sub.fill_(0x55555555)
        torch.cuda.synchronize()
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), enabled_only=True); d.sync()
        plan = d.launch_plan()
        assert plan["latency_layout"] == S // 16 and plan["packed_shared"] == 0, plan      # (one work item per workgroup of eight stream pairs)
        pv, kv = pairs.view(groups, R, 4, frames, 2), peaks.view(groups, R, blocks, 11)
        assert bool((pv[:, :, 0] == pv[0:1, :, 0]).all()), f"launch {c}: a workgroup's words differ from workgroup 0"
        assert bool((kv == kv[0:1]).all()), f"launch {c}: a workgroup's peaks differ from workgroup 0"
        assert bool((pairs[:, 1:] == 0x55555555).all()) and bool((sub == 0x55555555).all()), f"launch {c}: a silent output was written"
        got.append((pairs[:R, 0].cpu().numpy().copy(), peaks[:R].cpu().numpy().view(np.uint16).copy()))
    p0 = np.concatenate([g[0] for g in got], axis=1); k0 = np.concatenate([g[1] for g in got], axis=1)
    for s in (0, 1, 7, 8, R - 1):
        (rp, rs, rk, _), _ = oracle_run(W.F32_FMA, fs, -10 * 256, blob, base[s], blocks * calls, B, 16)
        assert np.array_equal(rp[0], p0[s]) and np.array_equal(rk, k0[s]), s
        assert int(np.abs(rp[1:]).max()) == 0 and int(np.abs(rs).max()) == 0
    d.close()


@pytest.mark.auto_layout
@pytest.mark.parametrize("flavor,S", [(W.F32_FMA, 1), (W.F32_FMA, 70), (1, 300), (0, 1), (0, 100)])
def test_small_host_calls_take_the_direct_path(flavor, S, monkeypatch):
    """The drop-in call as the firmware's main loop makes it (usb_audio.c:1326-1332): ONE packet per dspi_process on host buffers.  Such
    calls (<= 2 MB of buffers) skip the staged copies — the kernels read the packet from, and write to, a pinned host area — and must
    give what the staged path gives (DSPI_NO_DIRECT=1 on a second context): every word, peak, clip flag over 40 calls of one packet,
    with DSPI_OUT_ENABLED_ONLY, DSPI_OUT_I2S_SLOTS and DSPI_OUT_SPDIF in turn; stream 0 and S-1 against the oracle."""
    fl = int(flavor)
    fs, B = (96000, 96) if fl else (48000, 48)
    calls = 40
    blob = WL.full_chain_blob(fl)
    blob["outputs"][2]["enabled"] = 0; blob["outputs"][3]["enabled"] = 0          # a silent pair (ENABLED_ONLY)
    d, e = Dspi(flavor, S, device=0), Dspi(flavor, S, device=0)
    for x in (d, e): x.set_rate(fs); x.set_volume(-6 * 256); assert x.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * calls, fs, first_stream=19)          # (stream class 19: the full-scale square sets clip flags)
    got = []
    for c in range(calls):
        part = np.ascontiguousarray(pcm[:, c * B:(c + 1) * B])
        kw = dict(enabled_only=(c % 4 == 1), i2s_slots=(c % 4 == 2), spdif=(c % 4 == 3), clip=True)
        monkeypatch.delenv("DSPI_NO_DIRECT", raising=False)
        p0, s0, k0 = d.process_host(part, 1, B, **kw); c0 = d.last_clip.copy()
        monkeypatch.setenv("DSPI_NO_DIRECT", "1")
        p1, s1, k1 = e.process_host(part, 1, B, **kw); c1 = e.last_clip.copy()
        assert np.array_equal(p0, p1) and np.array_equal(s0, s1) and np.array_equal(k0, k1) and np.array_equal(c0, c1), c
        if c % 4 == 0: got.append((c, p0, s0, k0))
    monkeypatch.delenv("DSPI_NO_DIRECT", raising=False)
    for s_ in sorted({0, S - 1}):
        o = Oracle(flavor, detmath=True); o.set_rate(fs); o.set_volume(-6 * 256); assert o.load_bulk(blob) == 0
        rp, rs, rk, _ = o.process(pcm[s_], calls, B)
        for c, p0, s0, k0 in got:
            assert np.array_equal(rp[:, c * B:(c + 1) * B], p0[s_]) and np.array_equal(rs[c * B:(c + 1) * B], s0[s_]) and np.array_equal(rk[c], k0[s_, 0]), (s_, c)
        assert o.status() == d.status(s_)
    d.close(); e.close()


@pytest.mark.parametrize("flavor,tiled", [(W.F32_FMA, False), (1, True), (0, False), (0, True)])
def test_host_buffer_pipeline_chunks(flavor, tiled):
    """dspi_process on HOST buffers large enough to take the chunked pipeline (rows cut into chunks, H2D / kernels / D2H of consecutive
    chunks on three streams, the caller's arrays pinned for the call): every word, sub word and peak must equal what a second context
    produces from device buffers in one piece, over two calls, and sampled streams must equal the oracle."""
    import torch
    fs, B, blocks, S = 96000, 96, 20, 1024
    dev = torch.device("cuda", 0)
    dh, dd = Dspi(flavor, S, device=0), Dspi(flavor, S, device=0)
    for d in (dh, dd):
        d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(int(flavor))) == 0
    pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
    R = dh.tile_streams(); nt = (S + R - 1) // R
    N, P, C = dh.N, dh.P, dh.C
    frames = B * blocks
    host_out = None
    for c in range(2):
        part = np.ascontiguousarray(pcm[:, c * frames:(c + 1) * frames])
        host_out = dh.process_host(part, blocks, B, tiled=tiled, out=host_out)          # the second call reuses the arrays
        t_in = torch.from_numpy(part).to(dev)
        shape_p = (nt, 2 * P, frames, R) if tiled else (S, P, frames, 2)
        shape_s = (nt, frames, R) if tiled else (S, frames)
        tp = torch.zeros(shape_p, dtype=torch.int32, device=dev); ts = torch.zeros(shape_s, dtype=torch.int32, device=dev)
        tk = torch.zeros((S, blocks, C), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        dd.process_device(t_in.data_ptr(), blocks, B, 16, tp.data_ptr(), ts.data_ptr(), tk.data_ptr(), tiled=tiled); dd.sync()
        assert np.array_equal(host_out[0], tp.cpu().numpy()), f"call {c}: pair words of the chunked host path differ from the device path"
        assert np.array_equal(host_out[1], ts.cpu().numpy()) and np.array_equal(host_out[2].view(np.int16), tk.cpu().numpy()), c
    pairs, sub = (dh.untile(host_out[0], host_out[1]) if tiled else (host_out[0], host_out[1]))
    for s in (0, R - 1, R, 2 * R + 5, S // 2, S - 1):
        (rp, rs, rk, _), status = oracle_run(flavor, fs, -20 * 256, WL.full_chain_blob(int(flavor)), pcm[s], blocks * 2, B, 16)
        assert np.array_equal(rp[:, frames:, :], pairs[s]) and np.array_equal(rs[frames:], sub[s]) and np.array_equal(rk[blocks:], host_out[2][s]), s
        assert status == dh.status(s)
    dh.close(); dd.close()


@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
@pytest.mark.parametrize("tiled", (False, True))
def test_i2s_slot_words_fused_into_the_chain(flavor, tiled, monkeypatch):
    """DSPI_OUT_I2S_SLOTS: the chain itself writes the left-justified words of the slots that are I2S slots (per stream: streams 3
    and 70 switch another slot than everybody else, so the row takes the per-lane kernels) — the same words the two-call sequence
    gives (dspi_process, then dspi_i2s_encode by type), which tests/test_gpu_parity.py::test_i2s_slot_words pins to the reference's
    i2s_wrap_producer_give; also through the latency layout (float, a preset of its class)."""
    fs, B, blocks = 48000, 48, 40            # (the boot mute and the type switch's pipeline mute take the first packets)
    S = 300 if flavor else 75                # (float: a third row whose presets differ in a number only — the packed per-lane-value kernel)
    n_pairs = 4 if flavor else 2
    def setup(d, latency_class):
        d.set_rate(fs); d.set_volume(-5 * 256)
        assert d.load_bulk(_latency_blob(delays=(0.0, 0.3, 1.7, 0.0, 95.0, 0.05, 12.0, 0.0, 3.0)) if latency_class else WL.full_chain_blob(int(flavor))) == 0      # (delays within the 40 packets)
        d.vendor_get(W.REQ["SET_OUTPUT_TYPE"], 1 | (1 << 8), cap=1, stream=-1)                    # slot 1 -> I2S, every stream (DSPI_ALL_STREAMS)
        for s in (3, 70): d.vendor_get(W.REQ["SET_OUTPUT_TYPE"], 0 | (1 << 8), cap=1, stream=s)   # slot 0 too, two streams only
        if flavor and not latency_class: d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -7.5), stream=270)
    for latency_class in ((False, True) if flavor else (False,)):
        if latency_class: monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
        pcm = WL.synth_pcm16(S, B * blocks, fs)
        plain, fused = Dspi(flavor, S, device=0), Dspi(flavor, S, device=0)
        setup(plain, latency_class); setup(fused, latency_class)
        p0, s0, k0 = plain.process_host(pcm, blocks, B)
        p1, s1, k1 = fused.process_host(pcm, blocks, B, tiled=tiled, i2s_slots=True)
        if tiled: p1, s1 = fused.untile(p1, s1)
        want = p0.copy()
        for s in range(S):
            mask = 0b10 | (0b01 if s in (3, 70) else 0)
            for pr in range(n_pairs):
                if mask & (1 << pr): want[s, pr] = (want[s, pr].astype(np.uint32) << np.uint32(8)).astype(np.int32)
        assert np.array_equal(p1, want), (latency_class, np.argwhere(p1 != want)[:4].tolist())
        assert np.array_equal(s1, s0) and np.array_equal(k1, k0), latency_class
        assert int(np.abs(p0[0, 1, -B:]).max()) > 0 and int(np.abs(p0[3, 0, -B:]).max()) > 0
        if latency_class: assert fused.launch_plan()["latency_layout"] > 0
        elif flavor: assert fused.launch_plan()["packed_per_lane_values"] > 0
        plain.close(); fused.close()


@pytest.mark.auto_layout
def test_paired_presets_widen_the_size_rule():
    """The library's own choice: 4 096 streams with output EQ and no leveller take the packed kernel when they share a preset (the chip is
    filled enough) and the latency layout when every stream has its own (the alternative being the packed per-lane-filter kernel:
    13.9 against 22.6 ms per 200 packets) — dspi_capi.cpp rebuild_launch_lists.  Sampled streams against their oracles either way."""
    S, B, blocks, fs = 4096, 48, 4, 48000
    blob = WL.full_chain_blob(1); blob["leveller"]["enabled"] = 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    watch = (0, 1, 130, 2047, 4095)
    for distinct in (False, True):
        d = Dspi(W.F32_FMA, S, device=0); o = {s_: Oracle(W.F32_FMA, detmath=True) for s_ in watch}
        for x in [d] + list(o.values()):
            x.set_rate(fs); x.set_volume(-9 * 256); assert x.load_bulk(blob) == 0
        p_ = blob["eq"][0][1]
        for s_ in range(S if distinct else 0):
            reqs = [(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s_)),
                    (W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 1, int(p_["type"]), 0, float(p_["freq"]), float(p_["q"]), 1.0 + 0.001 * s_))]
            for req, wv, pl in reqs:
                assert d.vendor_set(req, wv, pl, stream=s_) == 0
                if s_ in o: assert o[s_].vendor_set(req, wv, pl) == 0
        pairs, sub, peaks = d.process_host(pcm, blocks, B)
        plan = d.launch_plan()
        if distinct: assert latency_plan(plan) and plan["latency_layout_paired"] == S // 4, plan
        else: assert plan["latency_layout"] == 0 and plan["packed_shared"] == S // 128, plan
        for s_ in watch:
            rp, rs, rk, _ = o[s_].process(pcm[s_], blocks, B)
            assert np.array_equal(rp, pairs[s_]) and np.array_equal(rs, sub[s_]) and np.array_equal(rk, peaks[s_]), (distinct, s_)
        d.close()


@pytest.mark.parametrize("distinct", (False, True), ids=("shared", "paired"))
@pytest.mark.parametrize("shape,fs,B", [(1, 48000, 48), (2, 96000, 96), (3, 44100, 45), (3, 48000, 7)])
def test_spdif_subframes_fused_into_the_chain(shape, fs, B, distinct, monkeypatch):
    """DSPI_OUT_SPDIF: the latency layout's output waves write the IEC 60958 subframes themselves — every shape of the layout, three calls
    so that the block position runs across call boundaries (and a set position) — the words dspi_process + dspi_spdif_encode give, which
    test_spdif_subframes pins to the reference's spdif_update_subframe.  A launch that is not on the latency layout refuses the flag.
    `paired`: every stream a preset of its own (one structure): the paired-preset instances of the kernels, with the encoder."""
    monkeypatch.setenv("DSPI_F32_LAYOUT", "skew")
    blob = _latency_blob() if shape == 1 else WL.full_chain_blob(1)
    if shape == 2: blob["leveller"]["enabled"] = 0
    blocks, S = (16 if B >= 44 else 120), 9
    plain, fused = Dspi(W.F32_FMA, S, device=0), Dspi(W.F32_FMA, S, device=0)
    for d in (plain, fused):
        d.set_rate(fs); d.set_volume(-6 * 256); assert d.load_bulk(blob) == 0
        for s_ in range(S if distinct else 0):
            assert d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.25 * s_), stream=s_) == 0
            assert d.vendor_set(W.REQ["SET_OUTPUT_GAIN"], s_ % 9, struct.pack("<f", -1.0 - 0.5 * s_), stream=s_) == 0
    assert fused.spdif_block_pos(77) == 77
    pos = 77
    pcm = WL.synth_pcm16(S, B * blocks * 3, fs)
    for c in range(3):
        part = np.ascontiguousarray(pcm[:, c * B * blocks:(c + 1) * B * blocks])
        p0, s0, k0 = plain.process_host(part, blocks, B)
        want, nxt = plain.spdif_host(p0, pos)
        p1, s1, k1 = fused.process_host(part, blocks, B, spdif=True)
        assert fused.launch_plan()["latency_layout"] > 0 and (fused.launch_plan()["latency_layout_paired"] > 0) == distinct
        assert np.array_equal(p1, want), (c, np.argwhere(p1 != want)[:4].tolist())
        assert np.array_equal(s1, s0) and np.array_equal(k1, k0), c
        pos = nxt
        assert fused.spdif_block_pos() == pos
    assert int(np.abs(p0[0, 0, -B:]).max()) > 0
    plain.close(); fused.close()


@pytest.mark.auto_layout
@pytest.mark.parametrize("flavor,S,B", [(W.F32_FMA, 70, 48), (W.F32_FMA, 4096, 45), (W.F32_FMA, 65536, 48), (0, 150, 48), (0, 16384, 44)],
                         ids=("f32-70-packed", "f32-4096", "f32-65536", "q28-150", "q28-16384"))
def test_spdif_flag_on_every_context(flavor, S, B, monkeypatch):
    """DSPI_OUT_SPDIF is a property of the output, not of the stream count: on launches the latency layout does not serve, dspi_process
    runs the chain into a scratch buffer row chunk by row chunk and the subframe encoder from there (sample_encoding.h:27-47) — the words
    of the two-call sequence, the block position carried across calls, any flavour, any size, device buffers; two streams get a preset of
    their own (mixed kernels in one launch).  Sampled streams against the reference-pinned encoder of the oracle."""
    import torch
    fl = int(flavor)
    if S == 70: monkeypatch.setenv("DSPI_F32_LAYOUT", "packed")
    fs = 44100 if B in (44, 45) else 48000
    blocks, calls = 6, 2
    P = 4 if fl else 2
    dev = torch.device("cuda", 0)
    plain, fused = Dspi(flavor, S, device=0), Dspi(flavor, S, device=0)
    for d in (plain, fused):
        d.set_rate(fs); d.set_volume(-6 * 256); assert d.load_bulk(WL.full_chain_blob(fl)) == 0
        for s_ in (3, S - 2): d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -9.5), stream=s_)
    assert fused.spdif_block_pos(100) == 100
    pos = 100
    frames = B * blocks
    base = WL.synth_pcm16(min(S, 256), frames * calls, fs)
    reps = (S + base.shape[0] - 1) // base.shape[0]
    for c in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat(reps, 1, 1)[:S].contiguous()
        words = torch.empty((S, P, frames, 2), dtype=torch.int32, device=dev); want = torch.empty((S, P, frames, 4), dtype=torch.int32, device=dev)
        got = torch.full((S, P, frames, 4), 0x55555555, dtype=torch.int32, device=dev)
        sub0 = torch.empty((S, frames), dtype=torch.int32, device=dev); sub1 = torch.empty((S, frames), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        plain.process_device(pcm.data_ptr(), blocks, B, 16, words.data_ptr(), sub0.data_ptr(), 0)
        nxt = plain.spdif_device(words.data_ptr(), frames, pos, want.data_ptr()); plain.sync()
        fused.process_device(pcm.data_ptr(), blocks, B, 16, got.data_ptr(), sub1.data_ptr(), 0, spdif=True); fused.sync()
        assert fused.launch_plan()["latency_layout"] == 0 or S == 4096      # (4 096 streams on a shared preset: the size rule may take the latency layout for most lanes)
        assert torch.equal(got, want), (c, torch.nonzero(got != want)[:3].tolist())
        assert torch.equal(sub0, sub1)
        for s_ in (0, 3, S // 2, S - 1):
            w_ = words[s_].cpu().numpy(); g_ = got[s_].cpu().numpy().view(np.uint32)
            for p_ in range(P):
                ref, n2 = orclib.spdif_encode(w_[p_], pos, fs)
                assert n2 == nxt and np.array_equal(ref, g_[p_]), (c, s_, p_)
        pos = nxt
        assert fused.spdif_block_pos() == pos
        del words, want, got, sub0, sub1, pcm
    plain.close(); fused.close()


def test_spdif_with_enabled_only_on_mixed_launches(monkeypatch):
    """DSPI_OUT_SPDIF served in two passes (chain into a scratch chunk, encoder from there) together with DSPI_OUT_ENABLED_ONLY.
    The encoder reads the WHOLE scratch, so every kernel of the launch — latency-layout workgroups next to packed / per-lane ones — must
    have written the silent pairs' zero words there: the subframes are then those of the call without the flag, silent pairs included
    (encoded silence, not whatever the scratch held).  The context: 2 200 streams with a leveller-on preset of their own each (too many for
    the latency layout: the packed kernels), the last 200 on the shared preset with ONE live pair (BASELINE config 2's: the latency layout,
    which honours DSPI_OUT_ENABLED_ONLY).  Device buffers pre-filled with garbage; two calls, so the first call's scratch is stale in the second."""
    import torch
    monkeypatch.delenv("DSPI_F32_LAYOUT", raising=False)
    fs, B, blocks, S, own = 48000, 48, 4, 2400, 2200
    dev = torch.device("cuda", 0)
    full = Oracle(W.F32_FMA); assert full.load_bulk(WL.full_chain_blob(1)) == 0
    full_blob = full.collect_bulk(); full.close()
    ctxs = []
    for _ in range(2):
        d = Dspi(W.F32_FMA, S, device=0)
        d.set_rate(fs); d.set_volume(-10 * 256); assert d.load_bulk(WL.config2_blob(False)) == 0
        for s_ in range(own): assert d.load_bulk(full_blob, stream=s_) == 0
        ctxs.append(d)
    plain, only = ctxs
    frames = B * blocks
    base = WL.synth_pcm16(256, frames * 2, fs)
    for c in range(2):
        part = torch.from_numpy(np.ascontiguousarray(base[:, c * frames:(c + 1) * frames])).to(dev)
        pcm = part.repeat((S + 255) // 256, 1, 1)[:S].contiguous()
        a = torch.full((S, 4, frames, 4), 0x5A5A5A5A, dtype=torch.int32, device=dev); b = torch.full((S, 4, frames, 4), 0x13571357, dtype=torch.int32, device=dev)
        sa = torch.empty((S, frames), dtype=torch.int32, device=dev); sb = torch.empty((S, frames), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        plain.process_device(pcm.data_ptr(), blocks, B, 16, a.data_ptr(), sa.data_ptr(), 0, spdif=True); plain.sync()
        only.process_device(pcm.data_ptr(), blocks, B, 16, b.data_ptr(), sb.data_ptr(), 0, spdif=True, enabled_only=True); only.sync()
        plan = only.launch_plan()
        others = sum(v for k, v in plan.items() if not k.startswith("latency_layout"))
        assert plan["latency_layout"] + plan["latency_layout_paired"] > 0 and others > 0, plan      # a mixed launch, hence two passes
        assert torch.equal(a, b), (c, torch.nonzero(a != b)[:3].tolist())
        ref, _ = orclib.spdif_encode(np.zeros((frames, 2), dtype=np.int32), (c * frames) % 192, fs)
        for s_ in (own + 1, S - 1):      # the shared preset's silent pairs carry the subframes of silence
            w = b[s_].cpu().numpy().view(np.uint32)
            assert np.array_equal(w[2], ref) and np.array_equal(w[3], ref), s_
    plain.close(); only.close()


@pytest.mark.both_layouts
def test_spdif_channel_status_follows_each_streams_rate():
    """the sample-rate byte of the IEC 60958 channel status (audio_spdif.c:250-256) belongs to the device.  A context whose
    streams run at three different rates: DSPI_OUT_SPDIF (fused on the latency layout, two-pass elsewhere) and dspi_spdif_encode both
    give every stream the status bits of ITS rate — checked against the reference-pinned encoder of the oracle, over two 192-frame blocks."""
    B, blocks, S = 48, 9, 6
    rates = {0: 48000, 1: 44100, 2: 96000, 3: 48000, 4: 96000, 5: 44100}
    d, e = Dspi(W.F32_FMA, S, device=0), Dspi(W.F32_FMA, S, device=0)
    for x in (d, e):
        x.set_volume(-6 * 256); assert x.load_bulk(WL.full_chain_blob(1)) == 0
        for s_, r in rates.items(): assert x.set_rate(r, stream=s_) == 0
    pcm = WL.synth_pcm16(S, B * blocks, 48000)
    p0, _, _ = d.process_host(pcm, blocks, B)
    sf, nxt = d.spdif_host(p0, 7)
    e.spdif_block_pos(7)
    p1, _, _ = e.process_host(pcm, blocks, B, spdif=True)
    assert (e.launch_plan()["latency_layout"] > 0) == on_latency_layout()
    assert e.spdif_block_pos() == nxt
    for s_, r in rates.items():
        for p_ in range(4):
            ref, n2 = orclib.spdif_encode(p0[s_, p_], 7, r)
            assert n2 == nxt and np.array_equal(ref, sf[s_, p_]), ("dspi_spdif_encode", s_, p_)
            assert np.array_equal(ref, p1[s_, p_]), ("DSPI_OUT_SPDIF", s_, p_)
    assert not np.array_equal(sf[0, 0, :40], orclib.spdif_encode(p0[0, 0], 7, 96000)[0][:40])      # (the rate does show in the first 40 frames of a block)
    d.close(); e.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", (1, W.F32_FMA), ids=("canonical", "fma"))
def test_boot_from_populated_flash_has_no_first_boot_mute(flavor):
    """DSPI_BOOT_POPULATED_FLASH: a context of devices that do NOT boot for the first time starts unmuted — the default context arms the
    512-sample preset mute the firmware's first boot arms by writing its directory (flash_storage.c:1097-1100, :347-348).  Checked
    against the reference's own boot path: the firmware build booted from a flash that holds a directory and the factory-default
    preset in slot 0 (no write, no mute) plays, from the first frame, what the flagged context plays."""
    fma = bool(getattr(flavor, "fma", False))
    if not orclib.ref_available(1, "fw", fma): pytest.skip("needs the firmware build under oracle/_ref")
    fs, B, blocks, S = 48000, 48, 12, 6
    slots = {0: Oracle(1, x86_casts=True).save_slot(0)}
    dump = W.flash_dump(W.flash_directory(default_slot=0, last_active_slot=0, slot_occupied=1), slots)
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    d = Dspi(flavor, S, device=0, populated_flash=True); assert d.set_rate(fs) == 0; d.set_volume(-6 * 256)
    muted = Dspi(flavor, S, device=0); assert muted.set_rate(fs) == 0; muted.set_volume(-6 * 256)
    pairs, sub, peaks = d.process_host(pcm, blocks, B)
    mp, _, _ = muted.process_host(pcm, blocks, B)
    assert not np.array_equal(mp, pairs)      # the default context fades into its first-boot mute, the flagged one does not
    for s in (0, S - 1):
        fw = Oracle(1, ref="fw", flash=dump, fma=fma); assert fw.set_rate(fs) == 0; fw.set_volume(-6 * 256)
        assert fw.collect_bulk() == d.collect_bulk(s)
        fp, fsub, fk, _ = fw.process(pcm[s], blocks, B)
        assert np.array_equal(fp, pairs[s]) and np.array_equal(fsub, sub[s]) and np.array_equal(fk, peaks[s]), s
        fw.close()
    d.close(); muted.close()


@pytest.mark.both_layouts
@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_flash_dump_boots_device_context(flavor):
    """SURVEY 8f-4 on the GPU: dspi_load_flash_dump on a DEVICE context (v2 directory, v1 directory, corrupt selected slot -> factory
    defaults, legacy "DSP1" sector -> migrated slot 0; flash_storage.c:1047-1105, :370-417, :997-1045), then audio.  Two readings of the
    call: on a running device (default context) it is a preset switch — every word from the first packet on against the restatement opened
    with the same dump (same preset-load mute), the packets after the mutes against the firmware build booted from it; on a context of
    devices with a populated flash that has played nothing it is the BOOT — every word from frame 0 against the restatement and the
    firmware build booted from the dump."""
    from test_flash_dump import make_slots
    fl = int(flavor)
    slots = make_slots(fl); occ = sum(1 << n for n in slots)
    D, F = W.flash_dump, W.flash_directory
    bad = dict(slots); b = bytearray(bad[4]); b[100] ^= 0x40; bad[4] = bytes(b)
    cases = [(D(F(default_slot=4, last_active_slot=9, slot_occupied=occ, master_volume_db=-17.0), slots), 4),
             (D(F(version=1, default_slot=9, slot_occupied=occ, master_volume_mode=1, names={9: "Night"}), slots), 9),
             (D(F(default_slot=4, slot_occupied=occ), bad), 16 + 4),
             (D(None, {}, W.legacy_sector_from_slot(slots[4], fl, version=7)), 32)]
    fs, B, warm, blocks, S = 48000, 48, 120, 20, 70
    fma = bool(getattr(flavor, "fma", False))
    # (Q28: the x86 build of the reference casts an out-of-range limiter quotient to INT_MIN where the MCU saturates — DESIGN.md section 5 —
    #  so the firmware-build leg runs for the float flavour only; tests/test_oracle_vs_fw.py pins the Q28 restatement with x86 casts)
    have_fw = fl == 1 and orclib.ref_available(fl, "fw", fma)
    pcm = WL.synth_pcm16(S, B * (warm + blocks), fs)
    for dump, want in cases:
        d = Dspi(flavor, S, device=0)
        assert d.set_rate(fs) == 0
        d.set_volume(-12 * 256)
        assert d.load_flash_dump(dump) == want
        outs = [d.process_host(np.ascontiguousarray(pcm[:, a * B:b_ * B]), b_ - a, B) for a, b_ in ((0, warm), (warm, warm + blocks))]
        for s in (0, 5, 64, S - 1):
            o = Oracle(flavor, detmath=True)
            assert o.set_rate(fs) == 0
            o.set_volume(-12 * 256)
            assert o.load_flash_dump(dump) == want
            rp, rs, rk, _ = o.process(pcm[s], warm + blocks, B)
            assert np.array_equal(rp, np.concatenate([x[0][s] for x in outs], axis=1)), (want, s)
            assert np.array_equal(rs, np.concatenate([x[1][s] for x in outs])) and np.array_equal(rk, np.concatenate([x[2][s] for x in outs])), (want, s)
            assert o.status() == d.status(s)
            assert o.collect_bulk() == d.collect_bulk(s)
            if have_fw and s in (0, 64):
                fw = Oracle(fl, ref="fw", flash=dump, fma=fma)
                assert fw.set_rate(fs) == 0
                fw.set_volume(-12 * 256)
                fw.process(pcm[s][:warm * B], warm, B)                     # boot mute and its trace in the delay lines run out
                fp, fsub, fk, _ = fw.process(pcm[s][warm * B:], blocks, B)
                assert np.array_equal(fp, outs[1][0][s]) and np.array_equal(fsub, outs[1][1][s]) and np.array_equal(fk, outs[1][2][s]), ("firmware build", want, s)
                fw.close()
            o.close()
        d.close()
        # The BOOT path: a context of devices with a populated flash that has not played anything boots from the dump (preset_boot_load ->
        # apply_slot_to_live, flash_storage.c:1047-1082: no mute unless the boot itself writes the flash, no line zeroing) — every word
        # FROM THE FIRST FRAME against the restatement booted from the dump and against the firmware build booted from it.
        db = Dspi(flavor, S, device=0, populated_flash=True)
        assert db.load_flash_dump(dump) == want
        assert db.set_rate(fs) == 0
        db.set_volume(-12 * 256)
        bp, bs, bk = db.process_host(np.ascontiguousarray(pcm[:, :blocks * B]), blocks, B)
        for s in (0, 5, 64, S - 1):
            ob = Oracle(flavor, detmath=True, flash=dump)
            assert ob.boot_selection == want and ob.set_rate(fs) == 0
            ob.set_volume(-12 * 256)
            rp, rs, rk, _ = ob.process(pcm[s][:blocks * B], blocks, B)
            assert np.array_equal(rp, bp[s]) and np.array_equal(rs, bs[s]) and np.array_equal(rk, bk[s]), ("boot path", want, s)
            assert ob.status() == db.status(s) and ob.collect_bulk() == db.collect_bulk(s)
            if s == 0: ob0 = ob
            else: ob.close()
            if have_fw and s in (0, 64):
                fw = Oracle(fl, ref="fw", flash=dump, fma=fma)
                assert fw.set_rate(fs) == 0
                fw.set_volume(-12 * 256)
                fp, fsub, fk, _ = fw.process(pcm[s][:blocks * B], blocks, B)
                assert np.array_equal(fp, bp[s]) and np.array_equal(fsub, bs[s]) and np.array_equal(fk, bk[s]), ("firmware build, from frame 0", want, s)
                fw.close()
        # once audio has run the same call is a preset switch on a running device again (preset_load: mute, lines zeroed)
        assert db.load_flash_dump(dump) == want and ob0.load_flash_dump(dump) == want
        ap, asub, ak = db.process_host(np.ascontiguousarray(pcm[:, blocks * B:(blocks + 12) * B]), 12, B)
        rp, rs, rk, _ = ob0.process(pcm[0][blocks * B:(blocks + 12) * B], 12, B)
        assert np.array_equal(rp, ap[0]) and np.array_equal(rs, asub[0]) and np.array_equal(rk, ak[0]), ("preset switch after the boot", want)
        ob0.close(); db.close()


@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_pdm_sub_output(flavor):
    """the sigma-delta consumer of the sub output (dspi_pdm_modulate) is bit-exact against the CPU
    restatement: fed by the chain's own sub words, both layouts, state carried over three calls, restart of one stream,
    inputs beyond the hard limiter."""
    fs, B, blocks = 48000, 48, 10
    S = 150 if flavor else 90
    chain = Dspi(flavor, S, device=0); chain.set_rate(fs); chain.set_volume(-4 * 256); assert chain.load_bulk(WL.full_chain_blob(flavor)) == 0
    pcm = WL.synth_pcm16(S, B * blocks * 3, fs)
    d_sm, d_t = Dspi(flavor, S, device=0), Dspi(flavor, S, device=0)
    o = [PdmOracle() for _ in range(S)]
    rng = np.random.default_rng(5)
    for c in range(3):
        _, sub, _ = chain.process_host(np.ascontiguousarray(pcm[:, c * blocks * B:(c + 1) * blocks * B]), blocks, B)
        sub = sub.copy()
        sub[3] = rng.integers(-(1 << 30), 1 << 30, size=sub.shape[1], dtype=np.int64).astype(np.int32)     # beyond +-1.8: limiter
        sub[7, ::5] = -(1 << 31)
        if c == 2:
            d_sm.pdm_restart(11); d_t.pdm_restart(11); o[11].restart()
        w_sm = d_sm.pdm_host(sub)
        R = d_t.tile_streams(); nt = (S + R - 1) // R
        sub_t = np.zeros((nt * R, sub.shape[1]), dtype=np.int32); sub_t[:S] = sub
        sub_t = np.ascontiguousarray(sub_t.reshape(nt, R, -1).transpose(0, 2, 1))
        w_t = d_t.pdm_host(sub_t, tiled=True)                                  # [tile][frame][8][R]
        w_t = w_t.transpose(0, 3, 1, 2).reshape(nt * R, sub.shape[1], 8)[:S]
        assert np.array_equal(w_sm, w_t), c
        for s_ in range(S):
            assert np.array_equal(o[s_].run(sub[s_]), w_sm[s_]), (c, s_)
    for x in (chain, d_sm, d_t): x.close()


@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_spdif_subframes(flavor):
    """S/PDIF subframe encoding of the chain's own pair words (dspi_spdif_encode) is bit-exact against the
    oracle (which is pinned to the reference's spdif_update_subframe): both layouts, block position carried over calls,
    all three sample rates."""
    B, blocks = 48, 9                                # 432 frames: crosses the 192-frame block twice
    S = 140 if flavor else 70
    for fs in (48000, 96000, 44100):
        d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-3 * 256); assert d.load_bulk(WL.full_chain_blob(flavor)) == 0
        dt = Dspi(flavor, S, device=0); dt.set_rate(fs)
        pcm = WL.synth_pcm16(S, B * blocks * 2, fs)
        pos = 5
        for c in range(2):
            pairs, _, _ = d.process_host(np.ascontiguousarray(pcm[:, c * blocks * B:(c + 1) * blocks * B]), blocks, B)
            sf, nxt = d.spdif_host(pairs, pos)
            R = dt.tile_streams(); nt = (S + R - 1) // R
            P, F = pairs.shape[1], pairs.shape[2]
            pt = np.zeros((nt * R, 2 * P, F), dtype=np.int32)
            pt[:S] = pairs.transpose(0, 1, 3, 2).reshape(S, 2 * P, F)
            pt = np.ascontiguousarray(pt.reshape(nt, R, 2 * P, F).transpose(0, 2, 3, 1))
            sft, nxt_t = dt.spdif_host(pt, pos, tiled=True)                       # [tile][pair][frame][4][R]
            sft = sft.transpose(0, 4, 1, 2, 3).reshape(nt * R, P, F, 4)[:S]
            assert nxt == nxt_t == (pos + F) % 192 and np.array_equal(sf, sft)
            for s_ in (0, 1, S // 2, S - 1):
                for p_ in range(P):
                    ref, n2 = orclib.spdif_encode(pairs[s_, p_], pos, fs)
                    assert n2 == nxt and np.array_equal(ref, sf[s_, p_]), (fs, c, s_, p_)
            # an odd frame count takes the one-frame-per-lane kernel (even counts: two frames per lane): same subframes
            sf_odd, n_odd = d.spdif_host(np.ascontiguousarray(pairs[:, :, :F - 1]), pos)
            assert n_odd == (pos + F - 1) % 192 and np.array_equal(sf_odd, sf[:, :, :F - 1])
            pos = nxt
        d.close(); dt.close()


@pytest.mark.parametrize("flavor", [1, 0])
def test_i2s_slot_words(flavor):
    """slots switched to I2S (REQ_SET_OUTPUT_TYPE) take the chain's pair words left-justified (dspi_i2s_encode,
    audio_i2s_multi.c:217-226): bit-exact against the oracle (pinned to the reference's producer-give), both layouts; only the
    selected pairs are written; DSPI_I2S_PAIRS_BY_TYPE follows output_types[] (explicit mask otherwise)."""
    B, blocks, fs = 45, 7, 44100                    # 315 frames: odd count (the stream-major kernel works per frame)
    S = 140 if flavor else 70
    P = 4 if flavor else 2
    d = Dspi(flavor, S, device=0); d.set_rate(fs); d.set_volume(-3 * 256); assert d.load_bulk(WL.full_chain_blob(flavor)) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    words, m = d.i2s_host(np.zeros((S, P, B, 2), dtype=np.int32))
    assert m == 0 and not words.any()                                        # every slot is S/PDIF: nothing to encode
    assert d.vendor_get(W.REQ["SET_OUTPUT_TYPE"], 0x0101, 1, -1) == b"\x00"
    pairs, _, _ = d.process_host(pcm, blocks, B)                              # (the switch's mute envelope is in these words)
    F = pairs.shape[2]
    keep = np.full(pairs.shape, 0xDEADBEEF, dtype=np.uint32)
    words, m = d.i2s_host(pairs, out=keep.copy())
    assert m == 0b10
    for s_ in range(S):
        assert np.array_equal(words[s_, 1], orclib.i2s_frames(pairs[s_, 1])), s_
    assert np.array_equal(np.delete(words, 1, axis=1), np.delete(keep, 1, axis=1))
    full, m = d.i2s_host(pairs, pair_mask=(1 << P) - 1)
    assert m == (1 << P) - 1 and np.array_equal(full, pairs.view(np.uint32) << 8)
    R = d.tile_streams(); nt = (S + R - 1) // R
    pt = np.zeros((nt * R, 2 * P, F), dtype=np.int32)
    pt[:S] = pairs.transpose(0, 1, 3, 2).reshape(S, 2 * P, F)
    pt = np.ascontiguousarray(pt.reshape(nt, R, 2 * P, F).transpose(0, 2, 3, 1))         # [tile][output][frame][R]
    wt, m = d.i2s_host(pt, tiled=True)
    assert m == 0b10
    wt = wt.transpose(0, 3, 1, 2).reshape(nt * R, P, 2, F).transpose(0, 1, 3, 2)[:S]
    assert np.array_equal(wt[:, 1], words[:, 1]) and not wt[:, 0].any()
    even = np.ascontiguousarray(pairs[:, :, :F - 1])                       # an even frame count takes the two-frames-per-lane kernel
    we, m = d.i2s_host(even, pair_mask=0b101 if P > 2 else 0b01, out=keep[:, :, :F - 1].copy())
    sel = [0, 2] if P > 2 else [0]
    assert np.array_equal(we[:, sel], even[:, sel].view(np.uint32) << 8) and np.array_equal(np.delete(we, sel, axis=1), np.delete(keep[:, :, :F - 1], sel, axis=1))
    assert d.L.dspi_i2s_encode(d.h, pairs.ctypes.data, F, 1 << P, words.ctypes.data, 0) == -10    # DSPI_E_INVAL: no such pair
    d.close()


def test_per_band_taps_both_contracts():
    """SURVEY.md section 8d parity procedure: taps after every band (dspi_debug_eq_taps, the production sample loop).  Both GPU
    contracts against the oracle tap for tap — the reference's dsp_process_channel_block compiled without / with contraction where
    oracle/_ref is present — and the one-step difference between the contracts bounded per stage (tools/ulp_report.py prints the
    full table, profiles/r02_ulp_per_stage.md)."""
    fs, n = 96000, 6000
    rng = np.random.default_rng(3)
    x = (rng.integers(-16384, 16385, n) / 32768.0 * 0.7079).astype(np.float32)
    blob = WL.full_chain_blob(1)
    use_ref = orclib.ref_available(1, "ref") and orclib.ref_available(1, "ref", True)
    for fma in (False, True):
        d = Dspi(1, 3, device=0, fma=fma); o = Oracle(1, ref=use_ref, fma=fma)
        for z in (d, o):
            assert z.set_rate(fs) == 0
            z.set_volume(-20 * 256)
            assert z.load_bulk(blob) == 0
        for ch in (0, 1, 2, 5, 9, 10):
            taps, other = d.eq_taps(x, ch, stream=1)
            assert np.array_equal(taps.view(np.uint32), o.eq_taps(x, ch).view(np.uint32)), f"taps of channel {ch}, fma={fma}"
            if not fma:      # canonical trajectory, firmware-contract one-step result beside it
                for b in range(10):
                    peak = float(np.abs(taps[b + 1]).max())
                    if peak > 0:
                        assert float(np.abs(taps[b + 1] - other[b]).max()) <= 2.0 * float(np.spacing(np.float32(peak))), (ch, b)
        d.close()


@pytest.mark.parametrize("flavor", FLAVORS_WITH_KERNEL)
def test_refused_calls_leave_the_context_as_it_was(flavor):
    """dspi_process refuses what process_audio_packet could never be handed — no buffer, a packet longer than the firmware's largest
    (usb_audio.c:273-276, :588: 192 frames), a word size that is neither of the two alt settings (usb_descriptors.c), zero packets, flag bits
    this ABI does not define, subframes in a layout they do not have — with DSPI_E_INVAL and WITHOUT touching the streams: the calls
    around the refused ones give the oracle's words as if those had never been made (first-boot mute, filter state, delay lines, meters)."""
    import ctypes as C
    from dspi_amd.host import _Out, OUT_TILED, OUT_SPDIF
    S, fs, B, blocks = 3, 48000, 48, 6
    blob = WL.full_chain_blob(flavor)
    d = Dspi(flavor, S, device=0)
    assert d.set_rate(fs) == 0
    d.set_volume(-20 * 256)
    assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * blocks, fs)
    half = blocks // 2
    first = d.process_host(np.ascontiguousarray(pcm[:, :half * B]), half, B, 16)
    F = half * B
    pairs = np.zeros((S, d.P, F, 4), dtype=np.uint32); sub = np.zeros((S, F), dtype=np.int32); peaks = np.zeros((S, half, d.C), dtype=np.uint16)
    out = _Out(pairs.ctypes.data, sub.ctypes.data, peaks.ctypes.data, None)
    nxt = np.ascontiguousarray(pcm[:, half * B:])
    E_INVAL = -10
    call = lambda pcm_ptr, depth, nb, bl, o, fl: d.L.dspi_process(d.h, pcm_ptr, depth, nb, bl, o, fl)
    assert call(None, 16, half, B, C.byref(out), 0) == E_INVAL                       # no input
    assert call(nxt.ctypes.data, 16, half, B, None, 0) == E_INVAL                    # no dspi_out
    assert call(nxt.ctypes.data, 20, half, B, C.byref(out), 0) == E_INVAL            # neither 16- nor 24-bit
    assert call(nxt.ctypes.data, 16, 0, B, C.byref(out), 0) == E_INVAL               # no packets
    assert call(nxt.ctypes.data, 16, half, 0, C.byref(out), 0) == E_INVAL            # empty packets
    assert call(nxt.ctypes.data, 16, 1, 193, C.byref(out), 0) == E_INVAL             # longer than the firmware's largest packet
    assert call(nxt.ctypes.data, 16, half, B, C.byref(out), 1 << 30) == E_INVAL      # an undefined flag bit
    assert call(nxt.ctypes.data, 16, half, B, C.byref(out), OUT_SPDIF | OUT_TILED) == E_INVAL
    assert b"DSPI_OUT_SPDIF" in d.L.dspi_last_error(d.h)
    assert not pairs.any() and not sub.any() and not peaks.any()                     # nothing was written either
    second = d.process_host(nxt, half, B, 16)
    for s in range(S):
        o2 = Oracle(flavor, detmath=True); assert o2.set_rate(fs) == 0; o2.set_volume(-20 * 256); assert o2.load_bulk(blob) == 0
        rp, rs, rk, _ = o2.process(pcm[s], blocks, B, 16)
        got_p = np.concatenate([first[0][s], second[0][s]], axis=1); got_s = np.concatenate([first[1][s], second[1][s]]); got_k = np.concatenate([first[2][s], second[2][s]])
        assert np.array_equal(got_p, rp), f"stream {s}: pair words differ after refused calls"
        assert np.array_equal(got_s, rs) and np.array_equal(got_k, rk)
    d.close()
