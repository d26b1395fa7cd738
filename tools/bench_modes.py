"""Config-3 chain in every input mode: 16/24-bit words x packet lengths (96 kHz: 96, 48 kHz: 48, 44.1 kHz: 45/44 ragged)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi

S = int(os.environ.get("S", 65536))
FL = int(os.environ.get("FLAVOR", 1))      # 1 = RP2350 float chain, 0 = RP2040 Q28 chain
NP, NC = (8, 11) if FL else (4, 7)
dev = torch.device("cuda", 0)
for depth in (16, 24):
    for fs, B, NB in ((96000, 96, 25), (48000, 48, 50), (44100, 45, 50), (44100, 44, 50))[0 if FL else 1:]:
        d = Dspi(FL, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256)
        assert d.load_bulk(WL.full_chain_blob(FL)) == 0
        if depth == 16:
            pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
        else:
            pcm = torch.randint(0, 256, (S, NB * B, 6), dtype=torch.uint8, device=dev)
            pcm[:, :, 2] = torch.randint(0, 64, (S, NB * B), dtype=torch.uint8, device=dev) + 224      # small signed values: -32..31 in the top byte
            pcm[:, :, 5] = pcm[:, :, 2]
        pairs = torch.empty((S * NP * NB * B,), dtype=torch.int32, device=dev); sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
        peaks = torch.empty((S, NB, NC), dtype=torch.int16, device=dev)
        for _ in range(2): d.process_device(pcm.data_ptr(), NB, B, depth, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
        d.sync()
        t0 = time.perf_counter(); steps = 5
        for _ in range(steps): d.process_device(pcm.data_ptr(), NB, B, depth, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=True)
        d.sync()
        dt = (time.perf_counter() - t0) / steps
        print(f"{'float' if FL else 'Q28'} chain, {depth}-bit, fs {fs}, {B}-frame packets: {dt * 1e3:.2f} ms/launch, {S * NB * B / dt:.3e} frames/s", flush=True)
        d.close()
