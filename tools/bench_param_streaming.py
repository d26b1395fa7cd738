import sys, os, time, struct
sys.path.insert(0, os.getcwd())
import torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
S = 65536
d = Dspi(W.F32_FMA, S, device=0); d.set_rate(96000); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(1)) == 0
dev = torch.device('cuda', 0)
pcm = torch.zeros((S, 96, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S, 4, 96, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, 96), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
d.process_device(pcm.data_ptr(), 1, 96, 16, pairs.data_ptr(), sub.data_ptr(), 0); d.sync()
t0 = time.perf_counter()
for s in range(S): d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s), stream=s)
t1 = time.perf_counter()
d.process_device(pcm.data_ptr(), 1, 96, 16, pairs.data_ptr(), sub.data_ptr(), 0); d.sync()
t2 = time.perf_counter()
for s in range(S): d.vendor_set(W.REQ["SET_OUTPUT_GAIN"], 3, struct.pack("<f", -1.0 - 0.0001 * s), stream=s)
t3 = time.perf_counter()
d.process_device(pcm.data_ptr(), 1, 96, 16, pairs.data_ptr(), sub.data_ptr(), 0); d.sync()
t4 = time.perf_counter()
print(f"65536 x vendor_set (clone + design): {t1-t0:.3f} s; first packet after (65536 images built, uploaded, tiled, lists): {t2-t1:.3f} s; second round of sets {t3-t2:.3f} s, packet after {t4-t3:.3f} s", d.launch_plan())
