#!/bin/bash
# tools/prof_all.sh <round tag> — rocprofv3 trace + PMC summaries of every kernel the bench can drive (run via gpurun); the
# summaries land in gpurun_out/profsum/ and are copied to profiles/ by hand.
R=${1:-r04}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { # tag, bench args, env...
  local tag=$1 args=$2; shift 2
  env BENCH_ARGS="$args" "$@" bash tools/prof.sh ${R}${tag} > gpurun_out/prof_${R}${tag}.log 2>&1
  tail -4 gpurun_out/prof_${R}${tag}.log
}
run a "--contract fma --out-layout tiled" CONTRACT=fma OUT_LAYOUT=tiled KERNEL_KEY=chain3 NOTE="config 3, firmware float contract (FMA), tiled words"
run b "--contract fma --out-layout stream" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain3 NOTE="config 3, firmware float contract (FMA), stream-major words (the delay line as the hand-over between the waves of a pair): the bench default"
run c "--contract canonical --out-layout tiled" CONTRACT=canonical OUT_LAYOUT=tiled KERNEL_KEY=chain3 NOTE="config 3, canonical contract, tiled words (round-1 configuration)"
run d "--contract canonical --out-layout stream" CONTRACT=canonical OUT_LAYOUT=stream KERNEL_KEY=chain3 NOTE="config 3, canonical contract, stream-major words"
run e "--config 5" CONTRACT=integer OUT_LAYOUT=stream KERNEL_KEY=chain5 STREAMS=16384 BLOCK_LEN=48 ALGO_BYTES=56 KERNEL_LIKE="%chain_kernel<0%" NOTE="config 5: Q28 7-channel chain, 16 384 streams, 48 kHz"
run k "--config 5 --streams 65536" CONTRACT=integer OUT_LAYOUT=stream KERNEL_KEY=chain5_64k STREAMS=65536 BLOCK_LEN=48 ALGO_BYTES=56 KERNEL_LIKE="%chain_kernel<0%" NOTE="Q28 7-channel chain at 65 536 streams (two workgroups per CU)"
run q "--config 5 --streams 1024" CONTRACT=integer OUT_LAYOUT=stream KERNEL_KEY=chain5_1024 STREAMS=1024 BLOCK_LEN=48 ALGO_BYTES=56 KERNEL_LIKE="%chain_kernel_q28_lat%" NOTE="Q28 chain on 1 024 streams: the integer flavour's latency layout (dspi_chain_q28_lat.inc: one stream per workgroup, one lane per (channel, stage))"
run l "--config 2" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain2 STREAMS=4096 BLOCK_LEN=48 PACKETS_PER_LAUNCH=2000 ALGO_BYTES=12 KERNEL_LIKE="%chain_kernel_skew%" NOTE="BASELINE config 2: 4 096 streams, master PEQ only, 2 000 packets per launch — the latency layout (dspi_chain_skew.inc), DSPI_OUT_ENABLED_ONLY"
run r "--config 2b" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain2b STREAMS=4096 BLOCK_LEN=48 PACKETS_PER_LAUNCH=2000 ALGO_BYTES=12 KERNEL_LIKE="%chain_kernel_skew%" NOTE="BASELINE config 2b (config 2 with every band a biquad): 4 096 streams, master PEQ only, 2 000 packets per launch — the latency layout (dspi_chain_skew.inc), DSPI_OUT_ENABLED_ONLY"
run m "--config 2" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain2_packed STREAMS=4096 BLOCK_LEN=48 PACKETS_PER_LAUNCH=2000 ALGO_BYTES=12 KERNEL_LIKE="%chain_kernel_pk%" DSPI_F32_LAYOUT=packed NOTE="BASELINE config 2 forced onto the packed kernel (DSPI_F32_LAYOUT=packed): the round-2 path, for comparison"
run p "--config 3 --streams 512" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=chain3_512 STREAMS=512 KERNEL_LIKE="%chain_kernel_skew_lev%" NOTE="BASELINE config 3's preset on 512 streams: the latency layout's third shape (dspi_chain_skew_lev.inc: leveller on, output rows, rings between the groups)"
run f "--config perstream" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=perstream KERNEL_LIKE="%chain_kernel_pk%" NOTE="65 536 distinct presets with identical filters (preamp per stream): packed kernel, per-lane values, shared band coefficients; stream-major words"
run i "--config perstream_eq --out-layout tiled" CONTRACT=fma OUT_LAYOUT=tiled KERNEL_KEY=perstream_eq KERNEL_LIKE="%chain_kernel_pk%" NOTE="65 536 distinct presets, one master band differs per stream: packed kernel, the master channels' band coefficients per lane from the value tiles; tiled words"
run n "--config perstream_eq --out-layout tiled" CONTRACT=fma OUT_LAYOUT=tiled KERNEL_KEY=perstream_eq_all KERNEL_LIKE="%chain_kernel_pk%" DSPI_DEBUG=1 NOTE="65 536 distinct presets, EVERY band of every channel differs per stream (DSPI_DEBUG=1): the worst case of the per-lane-filter kernel; tiled words"
run o "--config perstream_eq" CONTRACT=fma OUT_LAYOUT=stream KERNEL_KEY=perstream_eq KERNEL_LIKE="%chain_kernel_pk%" NOTE="65 536 distinct presets, one master band differs per stream; stream-major words"
run j "--config i2s --out-layout stream" CONTRACT=integer OUT_LAYOUT=stream KERNEL_KEY=i2s PACKETS_PER_LAUNCH=25 BLOCK_LEN=96 ALGO_BYTES=64 KERNEL_LIKE="%i2s%" NOTE="I2S slot words, 65 536 streams x 4 pairs x 2 400 frames"
run g "--config pdm --out-layout tiled" CONTRACT=integer OUT_LAYOUT=tiled KERNEL_KEY=pdm PACKETS_PER_LAUNCH=25 BLOCK_LEN=96 ALGO_BYTES=36 KERNEL_LIKE="%pdm_kernel%" NOTE="PDM sigma-delta modulator, 65 536 streams x 2 400 samples (frames = sub samples)"
run h "--config spdif --out-layout stream" CONTRACT=integer OUT_LAYOUT=stream KERNEL_KEY=spdif PACKETS_PER_LAUNCH=25 BLOCK_LEN=96 ALGO_BYTES=96 KERNEL_LIKE="%spdif%" NOTE="S/PDIF subframe encoder, 65 536 streams x 4 pairs x 2 400 frames"
ls gpurun_out/profsum
