#!/usr/bin/env python3
"""tools/ablate_power.py — does "time follows energy per frame" hold?  (run via gpurun)

Variants of BASELINE config 3 on the bench shape (65 536 streams, 50 packets of 96 frames per launch, firmware float contract), each
timed over 60 launches with socket power and shader clock sampled through librocm_smi64 (bench.PowerSampler): ms per launch, mean W,
mean MHz, J per launch, cycles = ms x MHz.  Prints a markdown table (profiles/r04_power_model.md is its output plus the reading)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import PowerSampler, synth_device
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

import json
from dspi_amd.host import source_fingerprint, LIB_PATH
import hashlib

S, NB, B, FS = 65536, 50, 96, 96000
dev = torch.device("cuda", 0)
OUT_JSON = os.environ.get("POWER_MODEL_JSON", "")       # e.g. gpurun_out/power_model_r06.json -> profiles/


def sample_idle(seconds=1.5):
    """socket power with the context created and nothing running: the static share of every figure below"""
    smi = PowerSampler(0); smi.start()
    t0 = time.perf_counter(); time.sleep(seconds); t1 = time.perf_counter()
    smi.stop()
    w = smi.window(t0, t1) if smi.ok else None
    return w["power_w"] if w else None


def sample_copy(seconds=2.5, gib=4):
    """joules per HBM byte of a plain streaming copy (torch's own copy kernel, 2 x `gib` GiB per pass: read + write), above idle"""
    n = gib << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_(); b.zero_()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize()
    smi = PowerSampler(0); smi.start()
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): b.copy_(a)
        torch.cuda.synchronize(); it += 20
    t1 = time.perf_counter()
    smi.stop()
    w = smi.window(t0 + 0.5, t1) if smi.ok else None      # (the first half second is the ramp)
    del a, b; torch.cuda.empty_cache()
    return (2.0 * n * it / (t1 - t0), w)


idle_w = sample_idle()
copy_bps, copy_w = sample_copy()
pcm = synth_device(torch, dev, S, NB * B, FS, 1234, True)
rows = []


def run(label, blob, tiled, outputs=True, steps=60, warm=8):
    d = Dspi(1, S, device=0, fma=True)
    d.set_rate(FS); d.set_volume(-20 * 256)
    assert d.load_bulk(blob) == 0
    frames = NB * B
    pairs = torch.empty(S * 8 * frames, dtype=torch.int32, device=dev) if outputs else None
    sub = torch.empty(S * frames, dtype=torch.int32, device=dev) if outputs else None
    peaks = torch.empty((S, NB, 11), dtype=torch.int16, device=dev)
    args = (pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()) if outputs else (0, 0, peaks.data_ptr())
    torch.cuda.synchronize()
    for _ in range(warm): d.process_device(pcm.data_ptr(), NB, B, 16, *args, tiled=tiled)
    d.sync()
    smi = PowerSampler(0); smi.start()
    t0 = time.perf_counter()
    for _ in range(steps): d.process_device(pcm.data_ptr(), NB, B, 16, *args, tiled=tiled)
    d.sync()
    t1 = time.perf_counter()
    smi.stop()
    w = smi.window(t0, t1) if smi.ok else None
    ms = (t1 - t0) / steps * 1e3
    rows.append((label, "tiled" if tiled else "stream", ms, w))
    pw, mhz = (w["power_w"], w["sclk_mhz"]) if w else (float("nan"), float("nan"))
    print(f"| {label} | {'tiled' if tiled else 'stream-major'} | {ms:.2f} | {pw:.0f} | {mhz:.0f} | {pw * ms / 1e3:.2f} | {ms * mhz / 1e3:.2f} |", flush=True)
    d.close()
    del pairs, sub, peaks
    torch.cuda.empty_cache()


print("| variant | words | ms / launch | mean W | mean MHz | J / launch | Mcycles / launch (ms x MHz) |\n|---|---|---|---|---|---|---|")
full = WL.full_chain_blob(1)
nodly = full.copy(); nodly["outputs"]["delay_ms"] = 0.0
nolev = full.copy(); nolev["leveller"]["enabled"] = 0
bare = nodly.copy(); bare["leveller"]["enabled"] = 0
for tiled in (True, False):
    run("full chain", full, tiled)
    run("no user delays (sub alignment only)", nodly, tiled)
    run("leveller off", nolev, tiled)
    run("no word buffers (pairs / sub null: meters only)", full, tiled, outputs=False)
    run("arithmetic only (no delays, leveller off, no word buffers)", bare, tiled, outputs=False)

# ---- the model bench.py's roofline.energy_floor_j is computed from ----
frames_per_launch = S * NB * B
arith = [r for r in rows if r[0].startswith("arithmetic only") and r[1] == "tiled" and r[3]]
if arith and copy_w and idle_w:
    a = arith[0]
    arith_j = a[3]["power_w"] * a[2] / 1e3
    j_per_byte = max(copy_w["power_w"] - idle_w, 0.0) / copy_bps
    model = {"what": "tools/ablate_power.py: energy model of BASELINE config 3 on the bench shape (one box, socket power through librocm_smi64 at ~100 Hz)",
             "src_sha16": source_fingerprint(), "lib_sha16": hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16],
             "idle_w": idle_w, "copy_bytes_per_s": copy_bps, "copy_w": copy_w["power_w"], "copy_sclk_mhz": copy_w["sclk_mhz"],
             "hbm_j_per_byte": j_per_byte, "hbm_pj_per_byte": j_per_byte * 1e12,
             "arith_j_per_launch": arith_j, "arith_j_per_frame": arith_j / frames_per_launch, "arith_ms": a[2], "arith_w": a[3]["power_w"], "arith_sclk_mhz": a[3]["sclk_mhz"],
             "frames_per_launch": frames_per_launch,
             "rows": [{"variant": r[0], "words": r[1], "ms": r[2], "power_w": r[3]["power_w"] if r[3] else None, "sclk_mhz": r[3]["sclk_mhz"] if r[3] else None,
                       "j_per_launch": (r[3]["power_w"] * r[2] / 1e3) if r[3] else None} for r in rows]}
    print(f"\nidle {idle_w:.0f} W; streaming copy {copy_bps / 1e12:.2f} TB/s at {copy_w['power_w']:.0f} W -> {j_per_byte * 1e12:.1f} pJ/B above idle; "
          f"arithmetic only {arith_j:.2f} J/launch = {arith_j / frames_per_launch * 1e9:.2f} nJ/frame")
    if OUT_JSON:
        json.dump(model, open(OUT_JSON, "w"), indent=1)
