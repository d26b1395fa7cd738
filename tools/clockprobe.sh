#!/bin/bash
# tools/clockprobe.sh — shader clock and issue counters per ablation variant (development aid).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/clk; rm -rf $OUT; mkdir -p $OUT
timeout 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $OUT -o pmc -- python tools/ablate.py > $OUT/run.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('gpurun_out/clk/**/pmc_results.db', recursive=True)[0])
rows = db.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%chain_kernel%' order by dispatch_id").fetchall()
d = collections.OrderedDict()
for did, n, v, dur in rows:
    d.setdefault(did, {'dur': dur})[n] = d.get(did, {}).get(n, 0) + v
for i, (did, r) in enumerate(d.items()):
    cyc = r.get('GRBM_GUI_ACTIVE', 0) / 8
    print(f"{i:3d} dur {r['dur']/1e6:7.3f} ms  clk {cyc / r['dur']:.3f} GHz  cyc {cyc/1e6:7.3f} M  valu {r.get('SQ_INSTS_VALU',0)/1e9:.3f} G  util {r.get('SQ_INSTS_VALU',0)*4.0/(1024*cyc):.3f}  waitinst/wave {r.get('SQ_WAIT_INST_ANY',0)/max(1,r.get('SQ_WAVE_CYCLES',1)):.3f}  wait/wave {r.get('SQ_WAIT_ANY',0)/max(1,r.get('SQ_WAVE_CYCLES',1)):.3f}")
PY
tail -20 $OUT/run.log
