#!/usr/bin/env python3
"""I2S slots: the two-call sequence (dspi_process, then dspi_i2s_encode over the pair words) against DSPI_OUT_I2S_SLOTS (the chain
writes the left-justified words itself) on BASELINE config 3 with every slot an I2S slot; device buffers, ms per 50-packet launch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
S, fs, B, NB = 65536, 96000, 96, 50
dev = torch.device("cuda", 0)
F = NB * B
pcm = torch.randint(-16384, 16385, (S, F, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S, 4, F, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, F), dtype=torch.int32, device=dev)
words = torch.empty((S, 4, F, 2), dtype=torch.int32, device=dev)
peaks = torch.empty((S, NB, 11), dtype=torch.int16, device=dev)
d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(1)) == 0
for slot in range(4): d.vendor_get(W.REQ["SET_OUTPUT_TYPE"], slot | (1 << 8), cap=1, stream=-1)
def timed(f, n=6):
    f(); d.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    d.sync(); return (time.perf_counter() - t0) / n * 1e3
def two_calls():
    d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
    d.i2s_device(pairs.data_ptr(), F, 0, words.data_ptr())
def fused():
    d.process_device(pcm.data_ptr(), NB, B, 16, words.data_ptr(), sub.data_ptr(), peaks.data_ptr(), i2s_slots=True)
a, b = timed(two_calls), timed(fused)
print(json.dumps({"workload": "config 3, 65 536 streams x 50 packets, all four slots I2S", "two_calls_ms": a, "fused_ms": b,
                  "bytes_per_frame_saved": 64, "note": "two calls: chain writes 32 B/frame of pair words, dspi_i2s_encode reads them and writes 32 B/frame of slot words"}))
