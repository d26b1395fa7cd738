cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06q; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default_driver_args.json 2> $O/bench.err
tail -c 300 $O/bench.err
