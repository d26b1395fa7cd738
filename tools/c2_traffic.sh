#!/bin/bash
# tools/c2_traffic.sh — FETCH_SIZE / WRITE_SIZE of BASELINE config 2 (latency layout, first shape) with and without the per-packet peak array (run via gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in peaks nopeaks; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/c2_$v_$c; 
    if [ $v = nopeaks ]; then export DSPI_BENCH_NO_PEAKS=1; else unset DSPI_BENCH_NO_PEAKS; fi
    timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/c2_${v}_$c -o pmc -- python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-parity > /tmp/c2_${v}_$c.log 2>&1
    python3 - $v $c <<'PY'
import sqlite3,glob,sys
v,c=sys.argv[1],sys.argv[2]
db=glob.glob(f'/tmp/c2_{v}_{c}/**/*.db',recursive=True)[0]
con=sqlite3.connect(db)
for r in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%chain_kernel%' group by kernel_name, counter_name"): print(v, r[0][30:80], r[1], '%.5g KB = %.2f B/frame'%(r[2], r[2]*1024/393216000.0), r[3])
PY
  done
done
