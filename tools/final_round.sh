#!/bin/bash
# tools/final_round.sh <round tag> — the round's closing measurements in one gpurun call: the whole -m gpu suite, smoke(), the default
# bench line (with the CPU baseline), every other bench configuration, the one-packet-per-call drop-in, small contexts, and the rocprofv3
# summaries of every kernel (tools/prof_all.sh).  Everything lands under gpurun_out/final_<tag>/ (+ gpurun_out/profsum/), copied to profiles/ by hand.
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final_$R; mkdir -p $O
(time python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/gputest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
# (first, so that every bench line below reads counter profiles and an energy model taken from THIS library: bench.py marks older ones stale)
# round 6: the energy model behind roofline.energy_floor_j (-> profiles/power_model_<tag>.json + <tag>_power_model.md), config 2's calibration
# kernel, the two-waves-per-SIMD arithmetic probe
POWER_MODEL_JSON=$O/power_model.json python tools/ablate_power.py > $O/power_model.md 2>&1
bash tools/probe/probe11_run.sh > $O/config2_calibration.md 2>&1
(cd tools/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o probe12 probe12.hip && ./probe12) > $O/probe12.md 2>&1
bash tools/prof_all.sh $R > $O/prof_all.log 2>&1
cp gpurun_out/profsum/${R}[a-r]_summary.md gpurun_out/profsum/traffic_${R}[a-r].json profiles/ 2>/dev/null
[ -f $O/power_model.json ] && cp $O/power_model.json profiles/power_model_$R.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 200 --no-cpu-baseline --no-variants > $O/bench_steps200.json 2>/dev/null
b() { local name=$1; shift; env "$@" > /dev/null 2>&1; }
run() { local name=$1; shift; "$@" > $O/bench_$name.json 2>/dev/null; }
run config_2 python bench.py --config 2 --no-cpu-baseline
run config_2b python bench.py --config 2b --no-cpu-baseline
run config_5 python bench.py --config 5 --no-cpu-baseline
run config_5_64k python bench.py --config 5 --streams 65536 --no-cpu-baseline
run config3_512streams python bench.py --config 3 --streams 512 --no-cpu-baseline --no-variants
run config_5_1024streams python bench.py --config 5 --streams 1024 --no-cpu-baseline
python tools/bench_q28_layouts.py > $O/q28_layouts.jsonl 2>/dev/null
run perstream python bench.py --config perstream --no-cpu-baseline
run perstream_eq python bench.py --config perstream_eq --no-cpu-baseline
DSPI_DEBUG=1 python bench.py --config perstream_eq --no-cpu-baseline > $O/bench_perstream_eq_every_band.json 2>/dev/null
run pdm python bench.py --config pdm --out-layout tiled
run spdif python bench.py --config spdif
run i2s python bench.py --config i2s
run blocks200_tiled python bench.py --out-layout tiled --blocks-per-step 200 --no-cpu-baseline --no-variants
python tools/bench_realtime.py --calls 30000 --presets config3,config3_leveller_off,config2 --out $O/realtime.json > $O/realtime.log 2>&1
DSPI_Q28_LAYOUT=chain python tools/bench_realtime.py --calls 3000 --flavors q28 --out $O/realtime_q28_chain_kernel.json > $O/realtime_q28_chain_kernel.log 2>&1
# the thin C host's node-level mode on the one GPU of this box: one context + feeder thread, the reduction through ncclCommInitAll / ncclAllReduce
python -c "
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from orclib import Oracle
from dspi_amd import workloads as WL
o = Oracle(1); assert o.load_bulk(WL.full_chain_blob(1)) == 0
open('$O/config3_bulk.bin', 'wb').write(o.collect_bulk())
"
dspi_amd/csrc/dspi_host -g 1 -f f32fma -s 65536 -r 96000 -b 96 -n 50 -c 20 -w 5 -v -20 -B $O/config3_bulk.bin > $O/dspi_host_g1.json 2> $O/dspi_host_g1.err
python tools/bench_small_contexts.py > $O/small_contexts_leveller_on.jsonl 2>/dev/null
LEVELLER=0 python tools/bench_small_contexts.py > $O/small_contexts_leveller_off.jsonl 2>/dev/null
SIZES=16,128,512,1024,2048,4096 PERSTREAM=1 python tools/bench_small_contexts.py > $O/small_contexts_per_stream_leveller_on.jsonl 2>/dev/null
SIZES=512,1024,2048,4096 LEVELLER=0 PERSTREAM=1 python tools/bench_small_contexts.py > $O/small_contexts_per_stream_leveller_off.jsonl 2>/dev/null
tail -3 $O/gputest.log; tail -2 $O/smoke.log; ls $O gpurun_out/profsum | head -80
