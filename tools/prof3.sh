#!/bin/bash
# tools/prof3.sh <tag> — memory-side PMC passes (TLB, L2, EA/HBM stalls, request latencies) of the default bench workload.
set -u
TAG=${1:-x}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_m1 -o pmc -- $BENCH > $OUT/m1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum -d $OUT/pmc_m2 -o pmc -- $BENCH > $OUT/m2.log 2>&1
# (TCC_* passes removed: rocprofv3 aborted and then hung on this pool when TCC_HIT/TCC_EA0_* were requested)
timeout 240 rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum -d $OUT/pmc_m5 -o pmc -- $BENCH > $OUT/m5.log 2>&1
ls $OUT
