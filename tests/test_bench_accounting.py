"""bench.py's byte accounting and input generator, on CPU: the algorithmic bytes per frame of every chain config (SURVEY.md section 8d:
both the HBM-resident-delay-line figure the roofline fraction uses and the launch-span figure), the profile lookup by kernel variant,
and the stream classes of the synthetic mix."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("dspi_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes():
    w = bench.chain_workload("3")
    a = bench.algorithmic_bytes(w, 50 * 96)
    assert a["resident"] == 104.0                 # 4 in + 8 x 4 S/PDIF words + 4 sub + 8 x 8 delayed outputs (write + read)
    assert 55.0 < a["span"] < 65.0                # delays shorter than the 4 800-frame launch count only dly/T of their 8 bytes
    # the strict figure (roofline.frac): 40 + 4 * (sum of min(dly, T) over the 8 delayed outputs + 8 * min(4096, T)) / T with the lines'
    # whole history kept — 48 + 96 + 192 + 480 + 960 + 1920 + 3840 + 4096 delay samples at 96 kHz
    assert abs(a["exact"] - (40.0 + 4.0 * (11632 + 8 * 4096) / 4800.0)) < 1e-9 and 76.9 < a["exact"] < 77.1
    assert a["span"] < a["exact"] < a["resident"]
    big, one = bench.algorithmic_bytes(w, 10 ** 9), bench.algorithmic_bytes(w, 1)
    assert big["span"] < 41.0 and big["exact"] < 41.0 and one["span"] == one["exact"] == one["resident"] == 104.0
    assert bench.algorithmic_bytes(bench.chain_workload("2"), 2000 * 48) == dict(exact=12.0, resident=12.0, span=12.0)
    q = bench.algorithmic_bytes(bench.chain_workload("5"), 50 * 48)
    assert q["resident"] == 56.0                  # 4 in + 4 x 4 words + 4 sub + 4 x 8 delayed outputs
    assert 24.0 < q["span"] < q["exact"] < 56.0   # 2 048-sample lines, 2 400-frame launches
    for name in ("perstream", "perstream_eq"):
        assert bench.algorithmic_bytes(bench.chain_workload(name), 50 * 96)["resident"] == 104.0


def test_profile_lookup_matches_the_variant():
    found = 0
    for key, contract, layout in (("chain3", "fma", "stream"), ("chain3", "fma", "tiled"), ("chain3", "canonical", "tiled"), ("chain5", "integer", "stream"),
                                  ("perstream", "fma", "stream")):
        p = bench.latest_profile(key, contract, layout)
        if p is None: continue
        found += 1
        t = json.load(open(os.path.join(ROOT, "profiles", p["source"])))
        assert t.get("kernel_key", "chain3") == key and t.get("contract", "canonical") == contract and t.get("out_layout", "stream") == layout
        assert 50.0 < p["hbm_bytes_per_frame"] < 400.0
    assert found >= 3
    assert bench.latest_profile("no-such-kernel", "fma", "stream") is None


def test_synthetic_mix_classes():
    fs, frames, S = 96000, 4800, 40
    pcm = bench.synth_device(torch, torch.device("cpu"), S, frames, fs, 7, True).numpy()
    assert pcm.shape == (S, frames, 2) and pcm.dtype == np.int16
    assert np.abs(pcm[0]).max() <= 16384 and np.abs(pcm[0]).max() > 12000            # white noise at -6 dBFS
    assert not pcm[18].any() and not pcm[38].any()                                    # digital silence
    assert set(np.unique(pcm[19, :, 0]).tolist()) == {-32768, 32767}                  # full-scale square
    assert np.abs(pcm[16, :frames // 2]).max() < 1100 < np.abs(pcm[16, frames // 2:]).max()      # bursts: -30 dBFS then -6 dBFS
    assert 8000 < np.abs(pcm[14]).max() <= 8231                                       # sweep at -12 dBFS
    shifted = bench.synth_device(torch, torch.device("cpu"), 4, frames, fs, 7, True, first_stream=16).numpy()
    assert not shifted[2].any()                                                       # classes follow the GLOBAL stream index (shards)
    noise = bench.synth_device(torch, torch.device("cpu"), S, frames, fs, 7, False).numpy()
    assert noise[18].any()


def test_source_fingerprint_tracks_the_kernel_sources(tmp_path):
    """Counter profiles carry the fingerprint of the sources they were taken from (tools/prof_summary.py); bench.py marks
    roofline.traffic stale when the tree it runs from has another one."""
    from dspi_amd import host
    a = host.source_fingerprint()
    assert len(a) == 16 and a == host.source_fingerprint()
    pkg = tmp_path / "pkg"; (pkg / "csrc").mkdir(parents=True); (tmp_path / "include").mkdir()
    (pkg / "csrc" / "k.hip").write_text("kernel"); (tmp_path / "include" / "api.h").write_text("api")
    f0 = host.source_fingerprint(pkg)
    (pkg / "csrc" / "k.hip").write_text("kernel, edited")
    f1 = host.source_fingerprint(pkg)
    (tmp_path / "include" / "api.h").write_text("api v2")
    assert len({f0, f1, host.source_fingerprint(pkg), a}) == 4
    p = bench.latest_profile("chain3", "fma", "stream")
    if p is not None: assert "src_sha16" in p
