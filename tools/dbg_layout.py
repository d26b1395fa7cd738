import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from dspi_amd import workloads as WL
from dspi_amd.host import Dspi
for (S, B, blocks, lev) in ((130, 48, 7, 1), (130, 48, 7, 0), (20, 45, 5, 1), (256, 96, 3, 1)):
    blob = WL.full_chain_blob(1); blob["leveller"]["enabled"] = lev
    outs = []
    for tiled in (False, True):
        d = Dspi(1, S, device=0); d.set_rate(48000); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
        pcm = WL.synth_pcm16(S, B * blocks, 48000)
        r = [d.process_host(pcm, blocks, B, tiled=tiled) for _ in range(2)]
        if tiled: r = [d.untile(p, s) + (k,) for (p, s, k) in r]
        outs.append(r); d.close()
    for c in range(2):
        for name, a, b in zip(("pairs", "sub"), outs[0][c][:2], outs[1][c][:2]):
            bad = np.argwhere(a != b)
            print(S, B, blocks, "lev", lev, "launch", c, name, "mismatches", len(bad), bad[:6].tolist(), "streams", sorted(set(bad[:, 0].tolist()))[:8], "frames", sorted(set(bad[:, 2 if name == "pairs" else 1].tolist()))[:12])
