cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06v; mkdir -p $O
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | tail -5) > $O/gputest_parity_fuzz.log 2>&1
bash tools/ab_bench.sh "libpre_tune.so libtune1.so libdspi_mi355x.so" 2 --config 2 > $O/ab_config2.log 2>&1
bash tools/ab_bench.sh "libpre_tune.so libtune1.so libdspi_mi355x.so" 2 --config 2b > $O/ab_config2b.log 2>&1
tail -3 $O/gputest_parity_fuzz.log; cat $O/ab_*.log
