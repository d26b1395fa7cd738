#!/bin/bash
# tools/probe/probe9_run.sh — probe9's kernels (known bytes: 2 GiB each) under rocprofv3's FETCH_SIZE / WRITE_SIZE, separate passes; prints
# reported bytes / known bytes per kernel.  Run via gpurun; output -> profiles/r03_fetch_write_calibration.md
cd "${GRAFT_REPO_ROOT:-/root/repo}/tools/probe"; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o probe9 probe9.hip || exit 1
./probe9
echo "| kernel | counter | bytes reported (avg per launch) | known bytes | reported / known |"
echo "|---|---|---|---|---|"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p9_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/p9_$c -o pmc -- ./probe9 > /tmp/p9_$c.log 2>&1
  python3 - $c <<'PY'
import sqlite3, glob, sys
c = sys.argv[1]
db = glob.glob(f'/tmp/p9_{c}/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
t = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('counters_collection')][0]
known = float(2 << 30)
for name, v in con.execute(f"select kernel_name, avg(value) from {t} where kernel_name like '%p9_%' and counter_name = ? group by kernel_name order by kernel_name", (c,)):
    reads = 'rd' in name
    if (c == 'FETCH_SIZE') != reads: continue      # (a read kernel under WRITE_SIZE and the reverse say nothing)
    print(f"| {name.split('(')[0]} | {c} | {v * 1024:.4g} | {known:.4g} | {v * 1024 / known:.3f} |")
PY
done
