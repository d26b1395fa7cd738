mkdir -p gpurun_out/r06d
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "latency_layout or config2" 2>&1 | tail -15) > gpurun_out/r06d/gputest_latency.log 2>&1
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/r06d/gputest.log 2>&1
bash tools/ab_bench.sh "libsk1.so libdspi_mi355x.so" 2 --config 2 > gpurun_out/r06d/ab_config2.log 2>&1
bash tools/ab_bench.sh "libsk1.so libdspi_mi355x.so" 2 --config 2b > gpurun_out/r06d/ab_config2b.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --config perstream_eq --out-layout tiled > gpurun_out/r06d/ab_perstream_eq_tiled.log 2>&1
bash tools/ab_bench.sh "libr05.so libdspi_mi355x.so" 2 --config perstream > gpurun_out/r06d/ab_perstream.log 2>&1
tail -4 gpurun_out/r06d/gputest_latency.log; tail -4 gpurun_out/r06d/gputest.log; cat gpurun_out/r06d/ab_*.log
