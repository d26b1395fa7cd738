#!/bin/bash
# tools/collect_final.sh <round tag> — after `gpurun -- 'bash tools/final_round.sh <tag>'`: copy what the round commits from gpurun_out/ (scratch)
# into profiles/ (tracked) under the round's names, and rebuild DESIGN.md's measurement tables from them (tools/design_tables.py).
R=${1:-r06}
cd "$(dirname "$0")/.."
O=gpurun_out/final_$R
cp gpurun_out/profsum/${R}[a-r]_summary.md gpurun_out/profsum/traffic_${R}[a-r].json profiles/
for f in default steps200 blocks200_tiled config3_512streams config_2 config_2b config_5 config_5_64k config_5_1024streams perstream perstream_eq perstream_eq_every_band pdm spdif i2s; do
  cp $O/bench_$f.json profiles/bench_${R}_$f.json
done
cp $O/realtime.json profiles/${R}_realtime.json
cp $O/realtime_q28_chain_kernel.json profiles/${R}_realtime_q28_chain_kernel.json
cp $O/q28_layouts.jsonl profiles/${R}_q28_layouts.jsonl
for f in small_contexts_leveller_on small_contexts_leveller_off small_contexts_per_stream_leveller_on small_contexts_per_stream_leveller_off; do cp $O/$f.jsonl profiles/${R}_$f.jsonl; done
[ -f $O/power_model.json ] && cp $O/power_model.json profiles/power_model_$R.json
grep "^{" $O/dspi_host_g1.json > profiles/${R}_dspi_host_g1.json
(tail -8 $O/gputest.log; echo; tail -2 $O/smoke.log) > profiles/gputest_${R}_final.log
[ -f $O/config2_calibration.md ] && cp $O/config2_calibration.md profiles/${R}_config2_calibration_raw.md
python tools/design_tables.py $R
tail -3 $O/gputest.log; tail -1 $O/smoke.log
