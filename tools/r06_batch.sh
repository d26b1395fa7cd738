cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06o; mkdir -p $O
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "latency or config2" 2>&1 | tail -6) > $O/gputest_latency.log 2>&1
bash tools/ab_bench.sh "libpre_peak.so libdspi_mi355x.so" 2 --config 2 > $O/ab_config2.log 2>&1
bash tools/ab_bench.sh "libpre_peak.so libdspi_mi355x.so" 2 --config 2b > $O/ab_config2b.log 2>&1
tail -4 $O/gputest_latency.log; cat $O/ab_*.log
