#!/usr/bin/env python3
"""Latency layout (dspi_chain_skew.inc) on config 2's preset: time per launch against the packet length at a fixed number of frames —
separates the per-step cost of the skewed cascade from the per-packet cost of the frame-parallel phases.  DSPI_F32_LAYOUT=packed for
the packed kernel on the same work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
S = int(os.environ.get("S", 4096)); FR = 96000
dev = torch.device("cuda", 0)
for biq in (False, True):
    for B in (48, 96, 192, 24):
        blocks = FR // B
        d = Dspi(W.F32_FMA, S, device=0); d.set_rate(48000); d.set_volume(-10 * 256); assert d.load_bulk(WL.config2_blob(biq)) == 0
        pcm = torch.randint(-16384, 16385, (S, FR, 2), dtype=torch.int16, device=dev)
        pairs = torch.empty((S, 4, FR, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, FR), dtype=torch.int32, device=dev)
        peaks = torch.empty((S, blocks, 11), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()); d.sync()
        t0 = time.perf_counter()
        for _ in range(3): d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
        d.sync(); dt = (time.perf_counter() - t0) / 3
        print(f"{'all-biquad' if biq else 'svf+biquad'} B={B:3d} packets={blocks:5d}: {dt*1e3:7.2f} ms  plan {d.launch_plan()['latency_layout']}", flush=True)
        d.close()
