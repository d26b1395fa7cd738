import sys, time, struct; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
S, NB, B, FS = 65536, 50, 48, 48000
dev = torch.device('cuda', 0)
pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
pairs = torch.empty((S, 2, NB * B, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, NB * B), dtype=torch.int32, device=dev); peaks = torch.empty((S, NB, 7), dtype=torch.int16, device=dev)
for mode in ('shared', 'perstream'):
    d = Dspi(0, S, device=0); d.set_rate(FS); d.set_volume(-20 * 256); assert d.load_bulk(WL.full_chain_blob(0)) == 0
    if mode == 'perstream':
        t0 = time.perf_counter()
        for s in range(S): d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s), stream=s)
        print('setup s', time.perf_counter() - t0)
    torch.cuda.synchronize()
    for _ in range(2): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
    d.sync(); t0 = time.perf_counter()
    for _ in range(5): d.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
    d.sync(); dt = (time.perf_counter() - t0) / 5
    print(mode, dt * 1e3, 'ms', S * NB * B / dt / 1e9, 'Gframes/s', d.launch_plan())
    d.close()
