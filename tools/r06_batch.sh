cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06j; mkdir -p $O
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -12) > $O/gputest.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 3 > $O/ab_default.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --out-layout tiled > $O/ab_tiled.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --config perstream > $O/ab_perstream.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --config perstream_eq --out-layout tiled > $O/ab_perstream_eq_tiled.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --config 5 > $O/ab_config5.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 1 --config 2 > $O/ab_config2.log 2>&1
tail -4 $O/gputest.log; for f in $O/ab_*.log; do echo $f; cat $f; done
