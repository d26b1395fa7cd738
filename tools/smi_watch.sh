#!/bin/bash
# tools/smi_watch.sh <bench args...> — power, shader clock and temperature (rocm-smi, 10 Hz) while bench.py runs 400 steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
( for i in $(seq 80); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.load(sys.stdin)['card0']
    keys=[k for k in d if any(t in k.lower() for t in ('power','sclk','junction','mclk'))]
    print(' | '.join('%s=%s'%(k[:40],d[k]) for k in keys))
except Exception as e: print('smi parse',e)
"; sleep 0.1; done ) > gpurun_out/smi_watch.log 2>&1 &
W=$!
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-variants "$@" > gpurun_out/smi_bench.json 2>/dev/null
wait $W
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/smi_bench.json') if l.startswith('{')][-1]); print('ms/step', d['ms_per_step'])"
sort gpurun_out/smi_watch.log | uniq -c | sort -rn | head -12
