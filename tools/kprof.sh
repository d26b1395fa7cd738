#!/bin/bash
# tools/kprof.sh <python script> — cycles, clock and VALU issue utilisation of every chain-kernel dispatch of a tool run
# (development aid; GRBM_GUI_ACTIVE / 8 XCDs = kernel cycles).  Env vars pass through, e.g.  S=65536 tools/kprof.sh tools/bench_perstream.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/kprof; rm -rf $OUT; mkdir -p $OUT
timeout 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $OUT -o pmc -- python "$@" > $OUT/run.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('gpurun_out/kprof/**/pmc_results.db', recursive=True)[0])
rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%chain_kernel%' order by dispatch_id").fetchall()
d = collections.OrderedDict()
for did, kn, n, v, dur in rows:
    r = d.setdefault(did, {'dur': dur, 'k': kn.split('(')[0][-40:]})
    r[n] = r.get(n, 0) + v
last = None
for i, (did, r) in enumerate(d.items()):
    cyc = r.get('GRBM_GUI_ACTIVE', 0) / 8
    key = (r['k'], round(r.get('SQ_INSTS_VALU', 0) / 1e7))
    if key == last: continue
    last = key
    print(f"{i:3d} {r['k']:40s} {r['dur']/1e6:7.3f} ms clk {cyc / r['dur']:.2f} GHz cyc {cyc/1e6:7.2f} M valu {r.get('SQ_INSTS_VALU',0)/1e9:.3f} G vmem rd/wr {r.get('SQ_INSTS_VMEM_RD',0)/1e6:.1f}/{r.get('SQ_INSTS_VMEM_WR',0)/1e6:.1f} M util {r.get('SQ_INSTS_VALU',0)*4.0/(1024*max(cyc,1)):.3f} waitinst {r.get('SQ_WAIT_INST_ANY',0)/max(1,r.get('SQ_WAVE_CYCLES',1)):.2f}")
PY
