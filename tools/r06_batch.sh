cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06r; mkdir -p $O
NODE=$(cat /sys/bus/pci/devices/$(python -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); b=ctypes.create_string_buffer(64); h.hipDeviceGetPCIBusId(b,64,0); print(b.value.decode().lower())")/numa_node 2>/dev/null)
CPUS=$(cat /sys/devices/system/node/node${NODE:-0}/cpulist 2>/dev/null)
FIRST=$(echo $CPUS | sed 's/[-,].*//')
echo "gpu numa node $NODE cpus $CPUS first $FIRST" > $O/info.txt
for rep in 1 2 3 4 5 6; do
  python tools/bench_realtime.py --calls 30000 --streams 1 --flavors f32fma --no-check > $O/rt_free_$rep.jsonl 2>&1
  taskset -c $((FIRST+4)) python tools/bench_realtime.py --calls 30000 --streams 1 --flavors f32fma --no-check > $O/rt_pinned_$rep.jsonl 2>&1
done
python - <<'PY'
import json, glob
for mode in ("free", "pinned"):
    for f in sorted(glob.glob(f"gpurun_out/r06r/rt_{mode}_*.jsonl")):
        for l in open(f):
            if l.startswith("{"):
                r = json.loads(l)
                print(mode, "p50 %.1f p99 %.1f p99.9 %.1f p99.99 %.1f max %.1f" % (r["p50_us"], r["p99_us"], r["p99_9_us"], r["p99_99_us"], r["max_us"]), r["hist_log2_us"]["counts"][3:9], r["direct_path"]["max_wait_us"])
PY
cat $O/info.txt
