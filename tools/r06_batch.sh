#!/bin/bash
# ad hoc batch of the round (run via gpurun): soak runs of configs 5 and 2 (thousands of launches, then the oracle replays them all for the checked streams)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06y
python bench.py --config 5 --steps 3000 --warmup 10 --no-cpu-baseline --no-variants > gpurun_out/r06y/bench_soak_q28.json 2>/dev/null
python bench.py --config 2 --steps 300 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/r06y/bench_soak_config2.json 2>/dev/null
for f in gpurun_out/r06y/bench_soak_q28.json gpurun_out/r06y/bench_soak_config2.json; do
  python - $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["steps"], round(d["ms_per_step"], 3), d.get("parity_checked"), d.get("parity_launches_replayed"))
PY
done
