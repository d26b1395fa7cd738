"""Throughput of the PDM sigma-delta consumer (dspi_pdm_modulate) on the bench shape: 65 536 streams x
2400 Q28 sub samples per call, tiled layouts.  4 B in + 32 B out per sample; ~2 800 integer VALU instructions per sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd.host import Dspi

S, F = int(os.environ.get("S", 65536)), int(os.environ.get("F", 2400))
dev = torch.device("cuda", 0)
d = Dspi(1, S, device=0)
R = d.tile_streams(); nt = (S + R - 1) // R
for tiled in (True, False):
    shape_in = (nt, F, R) if tiled else (S, F)
    sub = torch.randint(-(1 << 27), 1 << 27, shape_in, dtype=torch.int32, device=dev)
    words = torch.empty((nt * R if tiled else S) * F * 8, dtype=torch.int32, device=dev)
    for _ in range(2):
        d.pdm_device(sub.data_ptr(), F, words.data_ptr(), tiled=tiled)
    d.sync()
    t0 = time.perf_counter(); steps = 5
    for _ in range(steps):
        d.pdm_device(sub.data_ptr(), F, words.data_ptr(), tiled=tiled)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    sps = S * F / dt
    print(f"PDM modulator, {S} streams x {F} samples, {'tiled' if tiled else 'stream-major'}: {dt * 1e3:.2f} ms/call, {sps:.3e} samples/s "
          f"({sps / 48000:.0f} real-time 48 kHz subs), {sps * 36 / 1e9:.0f} GB/s algorithmic ({sps * 36 / 8e12:.3f} of 8 TB/s), "
          f"{sps * 256 / 1e12:.2f} T modulator steps/s")
d.close()
