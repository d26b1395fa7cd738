cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06i; mkdir -p $O
bash tools/ab_bench.sh "libr05.so libinl.so libnostep2.so" 3 > $O/ab_default.log 2>&1
bash tools/ab_bench.sh "libr05.so libinl.so" 2 --out-layout tiled > $O/ab_tiled.log 2>&1
bash tools/ab_bench.sh "libr05.so libinl.so" 2 --config perstream > $O/ab_perstream.log 2>&1
bash tools/ab_bench.sh "libr05.so libinl.so" 2 --config perstream_eq --out-layout tiled > $O/ab_perstream_eq_tiled.log 2>&1
(cd tools/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o probe12 probe12.hip && ./probe12) > $O/probe12.md 2>&1
for f in $O/ab_*.log; do echo $f; cat $f; done; cat $O/probe12.md
