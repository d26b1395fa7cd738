import os, sys
os.environ["DSPI_F32_LAYOUT"]="skew"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi
from test_gpu_parity import _latency_blob
fs,B,blocks,S=48000,48,6,4
pcm=WL.synth_pcm16(S,B*blocks,fs)
for name,kw in (("plain config2",None),("xfeed off loud off",dict(xfeed=False,loud=False)),("xfeed on loud off",dict(xfeed=True,loud=False)),("xfeed off loud on",dict(xfeed=False,loud=True)),("full",dict())):
    blob = WL.config2_blob(False) if kw is None else _latency_blob(**kw)
    if kw is not None and os.environ.get("NODELAY"): 
        for o in range(9): blob["outputs"][o]["delay_ms"]=0.0
    d=Dspi(1,S,device=0); d.set_rate(fs); d.set_volume(-7*256); assert d.load_bulk(blob)==0
    pairs,sub,peaks=d.process_host(pcm,blocks,B)
    print(name, d.launch_plan())
    for s in range(S):
        o=Oracle(1,detmath=True); o.set_rate(fs); o.set_volume(-7*256); o.load_bulk(blob)
        rp,rs,rk,_=o.process(pcm[s],blocks,B)
        bad=[(p,side) for p in range(4) for side in range(2) if not np.array_equal(rp[p,:,side],pairs[s][p,:,side])]
        print("  stream",s,"bad pair/side:",bad,"sub ok",np.array_equal(rs,sub[s]),"peaks ok",np.array_equal(rk,peaks[s]), "first bad frame", [int(np.argwhere(rp[p,:,sd]!=pairs[s][p,:,sd])[0][0]) for p,sd in bad][:8])
    d.close()
