cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06m; mkdir -p $O
for rep in 1 2 3; do
  for mode in query flag; do
    DSPI_DIRECT_POLL=$mode python tools/bench_realtime.py --calls 30000 --streams 1,16,128 --flavors f32fma --no-check > $O/rt_${mode}_$rep.jsonl 2>&1
  done
done
python - <<'PY'
import json, glob
for mode in ("query", "flag"):
    for f in sorted(glob.glob(f"gpurun_out/r06m/rt_{mode}_*.jsonl")):
        for l in open(f):
            if l.startswith("{"):
                r = json.loads(l)
                print(mode, r["streams"], "p50 %.1f p99 %.1f p99.9 %.1f p99.99 %.1f max %.1f over %d" % (r["p50_us"], r["p99_us"], r["p99_9_us"], r["p99_99_us"], r["max_us"], r["n_over_packet"]), r["direct_path"]["max_enqueue_us"], r["direct_path"]["max_wait_us"])
PY
