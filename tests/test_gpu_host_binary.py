"""The C host (dspi_amd/csrc/dspi_host.c -> dspi_host, plain C over include/dspi.h) run as a process on the GPU box: preset slot file
in, PCM file in, pair words out, against the oracle fed the same bytes.  Needs an MI355X."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import has_gpu
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no GPU")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dspi_amd", "csrc", "dspi_host")


@pytest.mark.parametrize("flavor,load", [(1, "slot"), (1, "bulk"), (W.F32_FMA, "slot"), (0, "slot")])
def test_dspi_host_against_oracle(tmp_path, flavor, load):
    fs, B, blocks, calls, vol_db = 48000, 48, 20, 3, -20
    ref = Oracle(flavor); assert ref.load_bulk(WL.full_chain_blob(flavor)) == 0
    image = ref.save_slot(4) if load == "slot" else ref.collect_bulk()
    pcm = WL.synth_pcm16(1, B * blocks, fs, first_stream=3)[0]
    (tmp_path / "preset.bin").write_bytes(image)
    (tmp_path / "pcm.raw").write_bytes(np.ascontiguousarray(pcm).tobytes())
    r = subprocess.run([HOST, "-f", ("f32fma" if getattr(flavor, "fma", False) else "f32") if flavor else "q28", "-s", "70", "-r", str(fs), "-b", str(B), "-n", str(blocks), "-c", str(calls),
                        "-P" if load == "slot" else "-B", str(tmp_path / "preset.bin"), "-i", str(tmp_path / "pcm.raw"), "-o", str(tmp_path / "pairs.raw"), "-v", str(vol_db)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert ("preset_load -> 0" if load == "slot" else "bulk_params_apply -> 0") in r.stdout
    o = Oracle(flavor, detmath=True)
    assert o.set_rate(fs) == 0
    o.set_volume(vol_db * 256)
    assert (o.load_slot(image) if load == "slot" else o.load_bulk(image)) == 0
    for _ in range(calls):
        pairs, sub, peaks, _ = o.process(pcm, blocks, B)
    got = np.frombuffer((tmp_path / "pairs.raw").read_bytes(), dtype=np.int32).reshape(pairs.shape)      # stream 0, last call
    assert np.array_equal(got, pairs)
    m = re.search(r"stream 0 status \((\d+) bytes\): peaks((?: \d+)+) clip 0x([0-9a-f]+)", r.stdout)
    st = o.status()
    assert m and int(m.group(1)) == len(st)
    assert [int(v) for v in m.group(2).split()] == np.frombuffer(st[:o.C * 2], dtype="<u2").tolist()
    assert int(m.group(3), 16) == int.from_bytes(st[-2:], "little")


def _xorshift_pcm(stream, frames):
    """dspi_host's synthetic input for GLOBAL stream index `stream` (xorshift32 seeded 0x9E3779B9 ^ s * 2654435761, -6 dBFS)."""
    x = (0x9E3779B9 ^ ((stream * 2654435761) & 0xFFFFFFFF)) & 0xFFFFFFFF
    if not x: x = 1
    out = np.empty(frames * 2, dtype=np.int16)
    for i in range(frames * 2):
        x ^= (x << 13) & 0xFFFFFFFF; x ^= x >> 17; x ^= (x << 5) & 0xFFFFFFFF
        out[i] = ((x >> 16) % 32769) - 16384
    return out.reshape(frames, 2)


@pytest.mark.parametrize("flavor,scaling", [(W.F32_FMA, "weak"), (0, "strong")])
def test_dspi_host_node_mode_through_rccl(tmp_path, flavor, scaling):
    """dspi_host -g 1: the node-level run of the thin C host (one context + feeder thread per GPU, device buffers, thread barriers around the
    timed calls) with its ONE collective on a real communicator — ncclCommInitAll + ncclAllReduce(sum frames, max seconds) over RCCL at
    world size 1 (the only size a one-GPU box has; the partition at N > 1 is tests/test_host_binary_cpu.py).  The JSON line carries
    bench.py's keys; device 0's first stream after the last call is checked against the oracle."""
    fl = int(flavor)
    fs, B, blocks, calls, warm, S = (96000, 96, 6, 3, 2, 300) if fl else (48000, 48, 8, 2, 1, 200)
    ref = Oracle(flavor); assert ref.load_bulk(WL.full_chain_blob(fl)) == 0
    (tmp_path / "bulk.bin").write_bytes(ref.collect_bulk())
    r = subprocess.run([HOST, "-g", "1", "-S", scaling, "-w", str(warm), "-f", "f32fma" if fl else "q28", "-s", str(S), "-r", str(fs), "-b", str(B), "-n", str(blocks),
                        "-c", str(calls), "-B", str(tmp_path / "bulk.bin"), "-v", "-20", "-o", str(tmp_path / "pairs.raw")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["steps"] == calls and d["warmup"] == warm and d["scaling"] == scaling and d["dry_run"] is False
    assert d["dist"]["backend"].startswith("rccl") and d["dist"]["world_size"] == 1
    assert d["config"]["shards"] == [[0, S]] and d["config"]["frames"] == S * blocks * B * calls
    assert d["value"] > 0 and abs(d["value"] - d["config"]["frames_per_s"] * d["config"]["channels"]) <= 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - d["config"]["seconds"] / calls * 1e3) < 1e-3
    o = Oracle(flavor, detmath=True)
    assert o.set_rate(fs) == 0
    o.set_volume(-20 * 256)
    assert o.load_bulk(ref.collect_bulk()) == 0
    pcm = _xorshift_pcm(0, blocks * B)
    for _ in range(warm + calls):
        pairs, _, _, _ = o.process(pcm, blocks, B)
    got = np.frombuffer((tmp_path / "pairs.raw").read_bytes(), dtype=np.int32).reshape(pairs.shape)
    assert np.array_equal(got, pairs)
    # the line's own window on the words: a checksum per device over its first stream's pair words (FNV-1a), recomputed here from the ORACLE's words
    cs = d["checked_streams"]
    assert len(cs) == 1 and cs[0]["device"] == 0 and cs[0]["stream"] == 0 and cs[0]["pair_words"] == pairs.size
    h = 2166136261
    for w in pairs.astype(np.int32).view(np.uint32).reshape(-1).tolist(): h = ((h ^ w) * 16777619) & 0xFFFFFFFF
    assert cs[0]["fnv1a"] == h
    roof = d["roofline"]
    assert roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and roof["achieved"] > 0
    assert len(d["affinity"]) == 1 and d["affinity"][0]["source"].startswith("pci")


def test_one_packet_calls_have_no_dropout_class_outliers(tmp_path):
    """the driver's run saw one 10 ms call among 3 000 one-packet calls.  10 000 calls per flavour, steady state: no call may
    take longer than 500 us (the packet carries 1 000 us of audio), and none may have reached the blocking wait.  Two retries: the outliers that
    were located (the calling thread off its core for 0.5-10 ms, ~1 call in 10^5, in either way of waiting) are the box's scheduler, not this
    library's to fix — three failures in a row would be something else."""
    import sys
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("a latency bound needs the GPU to itself: under pytest-xdist the other workers' kernels share it (run this file serially)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_realtime
    for fname, flavor, fs, B in (("f32fma", W.F32_FMA, 96000, 96), ("q28", 0, 48000, 48)):
        worst = None
        for attempt in range(3):
            r = bench_realtime.run(fname, flavor, 1, fs, B, 10000, 1000, check=(attempt == 0))
            worst = r
            if r["max_us"] <= 500.0 and r["n_over_packet"] == 0: break
        assert worst["max_us"] <= 500.0 and worst["n_over_packet"] == 0, worst
        assert worst["direct_path"]["calls"] == 10000 and worst["direct_path"]["blocking_waits"] == 0, worst["direct_path"]
        assert sum(worst["hist_log2_us"]["counts"]) == worst["calls"] - worst["first_calls"]
