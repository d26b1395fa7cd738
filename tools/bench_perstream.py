"""Every stream its own preset: throughput of the per-lane-parameter float kernel.  S streams, each with its
own preamp (so: S parameter images), full chain, 96 kHz, 96-frame packets, tiled words."""
import os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

S, NB = int(os.environ.get("S", 16384)), 25
B, FS = int(os.environ.get("B", 96)), int(os.environ.get("FS", 96000))      # e.g. B=45 FS=44100 for ragged packets
dev = torch.device("cuda", 0)
d = Dspi(1, S, device=0); d.set_rate(FS); d.set_volume(-20 * 256)
assert d.load_bulk(WL.full_chain_blob(1)) == 0
pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S * 8 * NB * B,), dtype=torch.int32, device=dev); sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
peaks = torch.empty((S, NB, 11), dtype=torch.int16, device=dev)


def timed(label):
    for _ in range(2): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
    d.sync()
    t0 = time.perf_counter(); steps = 5
    for _ in range(steps): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label}: {dt * 1e3:.2f} ms/launch, {S * NB * B / dt:.3e} frames/s, {S * NB * B / dt * 11:.3e} samples/s", flush=True)


timed(f"{S} streams, one shared preset (packed kernel)")
t0 = time.perf_counter()
for s in range(S):
    d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s), stream=s)
print(f"host: {S} per-stream parameter images in {time.perf_counter() - t0:.1f} s", flush=True)
timed(f"{S} streams, {S} presets (per-lane kernel)")
d.close()
