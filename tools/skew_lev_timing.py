"""Per-wave busy / total cycles of the latency layout's third shape (needs DSPI_LIB=.../libdspi_mi355x_timing.so for the cycle counters;
prints the time per launch with any build).  S, LEVELLER in the environment."""
import ctypes, os, sys, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/dspi_amd') else os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
S = int(os.environ.get("S", 256)); B, blocks, fs = 96, int(os.environ.get("BLOCKS", 200)), 96000
FR = B * blocks
dev = torch.device("cuda", 0)
blob = WL.full_chain_blob(1)
if os.environ.get("LEVELLER", "1") == "0": blob["leveller"]["enabled"] = 0
d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
pcm = torch.randint(-16384, 16385, (S, FR, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S, 4, FR, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, FR), dtype=torch.int32, device=dev)
peaks = torch.empty((S, blocks, 11), dtype=torch.int16, device=dev)
d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()); d.sync()
buf = (ctypes.c_ulonglong * 84)()
timing = hasattr(d.L, "dspi_debug_wave_timing")
if timing: d.L.dspi_debug_wave_timing(buf, 1)
t0 = time.perf_counter()
n = int(os.environ.get('N', 3))
for _ in range(n): d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
d.sync(); dt = (time.perf_counter() - t0) / n
if timing: d.L.dspi_debug_wave_timing(buf, 0)
nwg = (S + 3) // 4
print("ms/launch %.4f  plan %s" % (dt * 1e3, d.launch_plan()))
if timing: print("busy/total Mcyc per WG per launch: " + " ".join("w%d:%.4f/%.4f" % (w, buf[2 * w] / nwg / n / 1e6, buf[2 * w + 1] / nwg / n / 1e6) for w in range(9)))
