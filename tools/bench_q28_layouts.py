#!/usr/bin/env python3
"""tools/bench_q28_layouts.py — where the Q28 latency layout (dspi_chain_q28_lat.inc) stops paying: ms per dspi_process for a grid of
(streams, packets per call) under DSPI_Q28_LAYOUT=lat and =chain, device buffers (run via gpurun)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi
S, NB = int(sys.argv[1]), int(sys.argv[2])
B, FS = 48, 48000
dev = torch.device("cuda", 0)
d = Dspi(0, S, device=0); d.set_rate(FS); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(0)) == 0
pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S, 2, NB * B, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, NB * B), dtype=torch.int32, device=dev)
peaks = torch.empty((S, NB, 7), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for _ in range(5): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
d.sync()
n = max(5, min(200, int(0.2 / (NB * 50e-6))))
t0 = time.perf_counter()
for _ in range(n): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
d.sync()
print(json.dumps({"ms": (time.perf_counter() - t0) / n * 1e3}))
''' % ROOT

rows = []
for S in (64, 256, 1024, 2048, 4096, 8192, 16384):
    for NB in (1, 10, 50):
        rec = {"streams": S, "packets_per_call": NB}
        for lay in ("lat", "chain"):
            r = subprocess.run([sys.executable, "-c", CHILD, str(S), str(NB)], capture_output=True, text=True, env=dict(os.environ, DSPI_Q28_LAYOUT=lay))
            rec[lay + "_ms"] = json.loads(r.stdout.strip().splitlines()[-1])["ms"] if r.returncode == 0 else None
        rows.append(rec); print(json.dumps(rec), flush=True)
