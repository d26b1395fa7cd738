#!/bin/bash
# tools/prof.sh <tag> — rocprofv3 kernel trace + PMC passes of a bench.py workload (run via gpurun).
# BENCH_ARGS selects the variant, e.g. BENCH_ARGS="--contract canonical --out-layout stream" or "--config 5"; every
# pass measures the primary variant only (--no-variants).
set -u
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH_ARGS="${BENCH_ARGS:-} --no-cpu-baseline --no-variants --no-parity --no-configs --no-realtime"
BENCH="python bench.py --steps 3 --warmup 1 $BENCH_ARGS"
timeout 240 rocprofv3 -L > $OUT/counters.txt 2>&1
# the trace pass runs more steps so that the per-kernel average is the steady state bench.py times (the first launch of a
# process touches 10 GB of delay lines for the first time and is ~40 % slower)
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --steps 20 --warmup 3 $BENCH_ARGS > $OUT/trace.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
# the rocpd databases carry the whole code object (~50 MB each since the library holds both kernel families) and gpurun
# only returns 64 MiB: summarise on the box, keep the summary (+ traffic json) and drop the databases
mkdir -p gpurun_out/profsum
python tools/prof_summary.py $OUT gpurun_out/profsum/${TAG}_summary.md "${NOTE:-}"
find $OUT -name "*.db" -delete
du -sh $OUT
