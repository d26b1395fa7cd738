cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06y; mkdir -p $O /tmp/prelib
cp dspi_amd/csrc/libpre_lev.so /tmp/prelib/libdspi_mi355x.so
readelf -d dspi_amd/csrc/dspi_host | grep -i "rpath\|runpath" > $O/info.txt
for rep in 1 2 3; do
  LD_LIBRARY_PATH=/tmp/prelib python tools/bench_realtime.py --calls 20000 --streams 1,128 --flavors f32fma --presets config3,config3_leveller_off --no-check > $O/rt_pre_$rep.jsonl 2>&1
  python tools/bench_realtime.py --calls 20000 --streams 1,128 --flavors f32fma --presets config3,config3_leveller_off --no-check > $O/rt_new_$rep.jsonl 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06y/rt_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l); print(f.split("rt_")[1][:6], r["preset"], r["streams"], "p50 %.1f p99 %.1f max %.1f" % (r["p50_us"], r["p99_us"], r["max_us"]))
PY
cat $O/info.txt
