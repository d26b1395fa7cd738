#!/bin/bash
# tools/probe/probe11_run.sh — config 2's access pattern alone under rocprofv3's FETCH_SIZE / WRITE_SIZE (separate passes), with and without
# the per-packet peak words; prints reported bytes / known bytes.  Run via gpurun; output -> profiles/r06_config2_calibration.md
cd "${GRAFT_REPO_ROOT:-/root/repo}/tools/probe"; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o probe11 probe11.hip || exit 1
echo "| peaks | counter | KiB reported (avg per launch) | bytes reported | known bytes | reported / known | B/frame reported |"
echo "|---|---|---|---|---|---|---|"
for pk in 1 0; do
  PEAKS=$pk ./probe11 > /tmp/p11_$pk.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p11_${pk}_$c
    PEAKS=$pk timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/p11_${pk}_$c -o pmc -- ./probe11 > /tmp/p11_${pk}_$c.log 2>&1
    python3 - $pk $c <<'PY'
import sqlite3, glob, sys, re
pk, c = sys.argv[1], sys.argv[2]
db = glob.glob(f'/tmp/p11_{pk}_{c}/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if x.startswith('counters_collection')][0]
v = con.execute(f"select avg(value) from {t} where kernel_name like '%c2_pattern%' and counter_name = ?", (c,)).fetchone()[0]
m = re.search(r"read (\d+) .* written (\d+) .* frames (\d+)", open(f'/tmp/p11_{pk}.txt').read())
rd, wr, fr = float(m.group(1)), float(m.group(2)), float(m.group(3))
known = rd if c == 'FETCH_SIZE' else wr
print(f"| {pk} | {c} | {v:.0f} | {v * 1024:.4g} | {known:.4g} | {v * 1024 / known:.3f} | {v * 1024 / fr:.3f} |")
PY
  done
done
cat /tmp/p11_1.txt /tmp/p11_0.txt
