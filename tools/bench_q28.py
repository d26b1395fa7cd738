"""Throughput of the RP2040 Q28 chain on the GPU (BASELINE config 5 shape: 16 384 streams, 48 kHz, 48-frame packets,
7-channel chain).  Not the headline metric (bench.py is); reported in DESIGN.md §6 for completeness."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi

S, FS, B, NB = int(os.environ.get("S", 16384)), 48000, 48, 50
dev = torch.device("cuda", 0)
for tiled in (True, False):
    d = Dspi(0, S, device=0)
    d.set_rate(FS); d.set_volume(-20 * 256)
    assert d.load_bulk(WL.full_chain_blob(0)) == 0
    pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
    pairs = torch.empty((S * 4 * NB * B,), dtype=torch.int32, device=dev)
    sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, NB, 7), dtype=torch.int16, device=dev)
    for _ in range(2):
        d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=tiled)
    d.sync()
    t0 = time.perf_counter(); steps = 10
    for _ in range(steps):
        d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=tiled)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    fps = S * NB * B / dt
    print(f"Q28 7-ch chain, {S} streams, {'tiled' if tiled else 'stream-major'} words: {dt * 1e3:.2f} ms/launch, {fps:.3e} frames/s, "
          f"{fps * 7:.3e} samples/s, {fps / FS:.0f} real-time streams")
    d.close()
