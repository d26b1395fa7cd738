cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06s; mkdir -p $O
(DSPI_FUZZ_SEEDS=300 DSPI_Q28_LAYOUT=lat python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | tail -4) > $O/fuzz_300_seeds_q28_latency_layout.log 2>&1
cat $O/fuzz_300_seeds_q28_latency_layout.log
