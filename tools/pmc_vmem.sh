#!/bin/bash
# tools/pmc_vmem.sh — vector-memory instruction counts (SQ_INSTS_VMEM_RD / _WR / FLAT, SMEM, LDS) per launch of the packed chain kernel for the
# per-stream-preset workloads next to the shared preset (run via gpurun); how the spills of the per-lane-filter kernel were found
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
run() { tag=$1; shift
  rm -rf /tmp/pv_$tag; timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_LDS -d /tmp/pv_$tag -o pmc -- env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants $ARGS > /tmp/pv_$tag.log 2>&1
  python3 - $tag <<'PY'
import sqlite3,glob,sys
tag=sys.argv[1]
db=glob.glob(f'/tmp/pv_{tag}/**/*.db',recursive=True)[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
q="""select k.kernel_name, p.counter_name, avg(p.value), count(*) from counters_collection p join kernels k on 1=1 limit 0"""
# use the generic view
rows=c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%chain_kernel_pk%' group by kernel_name, counter_name").fetchall() if 'counters_collection' in tabs else []
for r in rows: print(tag, r[0][40:75], r[1], '%.4g'%r[2], r[3])
if not rows: print(tag,'tables',[t for t in tabs if 'pmc' in t or 'counter' in t][:10])
PY
}
ARGS="--config perstream_eq --out-layout tiled"
run one
run out5 DSPI_BENCH_EQ_CH=5
run all DSPI_DEBUG=1
ARGS="--config perstream --out-layout tiled"
run pv
ARGS="--out-layout tiled"
run shared
