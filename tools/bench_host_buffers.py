#!/usr/bin/env python3
"""PCIe-inclusive rate of the chain: dspi_process on HOST buffers — pageable numpy arrays in and out, the
library stages them (hipMemcpy H2D, launch, D2H) — for BASELINE config 3's workload on a slice of streams.  This is never the
`value` of bench.py (inputs resident in HBM); it is what a caller that keeps its audio in host memory sees.
    python tools/bench_host_buffers.py [streams] [packets]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
packets = int(sys.argv[2]) if len(sys.argv) > 2 else 50
fs, B = 96000, 96
d = Dspi(W.F32_FMA, S, device=0)
d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(1)) == 0
rng = np.random.default_rng(1)
pcm = rng.integers(-16384, 16385, size=(S, packets * B, 2), dtype=np.int16)
rows = []
for what, kw in (("pairs + sub + peaks", {}), ("pairs + sub", {"want_peaks": False})):
    bufs = d.process_host(pcm, packets, B, **kw)    # warm-up: staging buffers, first touch of the delay lines and of the host arrays,
    ts = []                                         # which every later call reuses (as a host with its own buffers does)
    for _ in range(5):
        t = time.perf_counter(); d.process_host(pcm, packets, B, out=bufs); ts.append(time.perf_counter() - t)
    t = sorted(ts)[2]
    frames = S * packets * B
    rows.append({"outputs": what, "streams": S, "packets": packets, "s_per_call": t, "frames_per_s": frames / t, "samples_per_s": frames * 11 / t,
                 "host_bytes_per_frame": 4 + 36 + (22 / B if not kw else 0), "host_GB_per_s": frames * 40 / t / 1e9})
print(json.dumps(rows))
