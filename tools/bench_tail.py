"""Ragged packets (44.1 kHz: 44/45-frame packets take the TAIL kernel instantiations): throughput next to the 48-frame case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi

S, NB = int(os.environ.get("S", 65536)), 50
dev = torch.device("cuda", 0)
for fs, B in ((48000, 48), (44100, 45), (44100, 44)):
    d = Dspi(1, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256)
    assert d.load_bulk(WL.full_chain_blob(1)) == 0
    pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
    pairs = torch.empty((S * 8 * NB * B,), dtype=torch.int32, device=dev); sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, NB, 11), dtype=torch.int16, device=dev)
    for _ in range(2): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
    d.sync()
    t0 = time.perf_counter(); steps = 5
    for _ in range(steps): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"fs {fs}, {B}-frame packets: {dt * 1e3:.2f} ms/launch, {S * NB * B / dt:.3e} frames/s", flush=True)
    d.close()
