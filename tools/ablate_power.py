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

S, NB, B, FS = 65536, 50, 96, 96000
dev = torch.device("cuda", 0)
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
