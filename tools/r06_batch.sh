cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06z; mkdir -p $O
(python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -5) > $O/gputest.log 2>&1
python tools/bench_realtime.py --calls 20000 --streams 1,16,128 --flavors f32fma --presets config3,config3_leveller_off,config2 --no-check > $O/rt.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06z/rt.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["preset"], r["streams"], "p50 %.1f p99 %.1f max %.1f" % (r["p50_us"], r["p99_us"], r["max_us"]))
PY
tail -3 $O/gputest.log
