"""Per-role busy / total cycles of the Q28 chain kernel's waves on BASELINE config 5 (development aid; needs the timing build:
make -C dspi_amd/csrc libdspi_mi355x_timing.so, DSPI_LIB=.../libdspi_mi355x_timing.so).  Roles of the seven-wave layout: 0 = left pass 1 +
hand-off (the lone wave of its SIMD), 1-5 = one output each, 6 = right pass 1."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi

S, FS, B, NB = int(os.environ.get("S", 16384)), 48000, 48, 50
dev = torch.device("cuda", 0)
d = Dspi(0, S, device=0)
d.set_rate(FS); d.set_volume(-20 * 256)
assert d.load_bulk(WL.full_chain_blob(0)) == 0
pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S * 4 * NB * B,), dtype=torch.int32, device=dev)
sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
peaks = torch.empty((S, NB, 7), dtype=torch.int16, device=dev)
args = (pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
for _ in range(3): d.process_device(*args, tiled=False)
d.sync()
buf = (ctypes.c_ulonglong * 84)()
d.L.dspi_debug_wave_timing(buf, 1)
steps = 10
t0 = time.perf_counter()
for _ in range(steps): d.process_device(*args, tiled=False)
d.sync()
dt = (time.perf_counter() - t0) / steps
d.L.dspi_debug_wave_timing(buf, 1)
nwg = (S + 63) // 64
print(f"config 5 on the timing build: {dt * 1e3:.2f} ms per launch, {nwg} workgroups")
for w in range(7):
    busy, tot = buf[2 * w] / nwg / steps, buf[2 * w + 1] / nwg / steps
    print(f"  role {w}: busy {busy / 1e6:.3f} Mticks of {tot / 1e6:.3f} per launch ({busy / max(tot, 1):.2f}); SIMD of workgroup 0's wave: {(buf[24 + w] >> 4) & 3}")
d.close()
