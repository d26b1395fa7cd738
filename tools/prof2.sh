#!/bin/bash
# tools/prof2.sh <tag> — second-level PMC passes (instruction issue / caches) of the default bench workload.
set -u
TAG=${1:-x}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/pmc_a -o pmc -- $BENCH > $OUT/a.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH -d $OUT/pmc_b -o pmc -- $BENCH > $OUT/b.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC -d $OUT/pmc_c -o pmc -- $BENCH > $OUT/c.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -d $OUT/pmc_d -o pmc -- $BENCH > $OUT/d.log 2>&1
ls $OUT
