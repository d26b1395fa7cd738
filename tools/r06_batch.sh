#!/bin/bash
# ad hoc batch of the round (run via gpurun): the counter profile of config 2b alone (tools/prof_all.sh's run r), then config 2b's bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/profsum gpurun_out/r06u
env BENCH_ARGS="--config 2b" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain2b STREAMS=4096 BLOCK_LEN=48 PACKETS_PER_LAUNCH=2000 ALGO_BYTES=12 KERNEL_LIKE="%chain_kernel_skew%" NOTE="BASELINE config 2b (config 2 with every band a biquad): 4 096 streams, master PEQ only, 2 000 packets per launch — the latency layout (dspi_chain_skew.inc), DSPI_OUT_ENABLED_ONLY" bash tools/prof.sh r06r > gpurun_out/prof_r06r.log 2>&1
tail -4 gpurun_out/prof_r06r.log
cp gpurun_out/profsum/r06r_summary.md gpurun_out/profsum/traffic_r06r.json profiles/
python bench.py --config 2b --no-cpu-baseline > gpurun_out/r06u/bench_config_2b.json 2>/dev/null
tail -c 1500 gpurun_out/r06u/bench_config_2b.json
