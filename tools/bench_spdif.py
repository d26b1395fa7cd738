"""Bandwidth of the S/PDIF subframe encoder (dspi_spdif_encode) on the bench shape: 65 536 streams x 4 pairs x
2400 frames per call.  Pure byte work: 8 B in + 16 B out per frame and pair = 96 B per stream-frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd.host import Dspi

S, F = int(os.environ.get("S", 65536)), int(os.environ.get("F", 2400))
dev = torch.device("cuda", 0)
d = Dspi(1, S, device=0); d.set_rate(96000)
R = d.tile_streams(); nt = (S + R - 1) // R
for tiled in (True, False):
    n_in = (nt * R if tiled else S) * 4 * F * 2
    pairs = torch.randint(-(1 << 23), 1 << 23, (n_in,), dtype=torch.int32, device=dev)
    out = torch.empty(n_in * 2, dtype=torch.int32, device=dev)
    for _ in range(2):
        d.spdif_device(pairs.data_ptr(), F, 0, out.data_ptr(), tiled=tiled)
    d.sync()
    t0 = time.perf_counter(); steps = 10
    for _ in range(steps):
        d.spdif_device(pairs.data_ptr(), F, 0, out.data_ptr(), tiled=tiled)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    gbs = S * F * 96 / dt / 1e9
    print(f"S/PDIF encoder, {S} streams x 4 pairs x {F} frames, {'tiled' if tiled else 'stream-major'}: {dt * 1e3:.2f} ms/call, "
          f"{S * F / dt:.3e} stream-frames/s, {gbs:.0f} GB/s moved ({gbs / 8000:.3f} of the 8 TB/s HBM peak)")
d.close()
