#!/usr/bin/env python3
"""tools/detmath_glibc_count.py — how often does glibc's log10f / powf change what the chain emits, against the correctly rounded functions of
include/dspi_detmath.h?  ("the count of packets where glibc's result differs".)  BASELINE config 3's preset, the firmware's
float contract, SURVEY 8d's stream classes; the oracle once with the header's functions and once with the platform libm in leveller.c:178 / :200 /
:206 (oracle/ref_math_hook.h routes them), same input: packets in which any output word differs, and the first such packet per stream.  CPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dspi_amd import workloads as WL
from orclib import Oracle

FS, B, PACKETS, STREAMS = 96000, 96, int(os.environ.get("PACKETS", "3000")), int(os.environ.get("STREAMS", "40"))
blob = WL.full_chain_blob(1)
pcm = WL.synth_pcm16(STREAMS, B * PACKETS, FS)
tot = diff = 0
first = []
worst = 0
for s in range(STREAMS):
    outs = []
    for det in (True, False):
        o = Oracle(1, detmath=det, fma=True); o.set_rate(FS); o.set_volume(-20 * 256); assert o.load_bulk(blob) == 0
        p, sub, pk, _ = o.process(pcm[s], PACKETS, B)
        outs.append((p, sub))
    d = (outs[0][0] != outs[1][0]).any(axis=(0, 2)).reshape(PACKETS, B).any(axis=1) | (outs[0][1] != outs[1][1]).reshape(PACKETS, B).any(axis=1)
    worst = max(worst, int(np.abs(outs[0][0].astype(np.int64) - outs[1][0].astype(np.int64)).max()))
    tot += PACKETS; diff += int(d.sum()); first.append(int(np.argmax(d)) if d.any() else -1)
print(f"{STREAMS} streams x {PACKETS} packets of {B} frames at {FS} Hz (classes by stream index % 20: noise, sweep, bursts, silence, square)")
print(f"packets in which glibc's libm changes an output word: {diff} of {tot} ({100.0 * diff / tot:.2f} %); largest word difference {worst} LSB of 24 bits")
print("first differing packet per stream (-1: none):", first)
