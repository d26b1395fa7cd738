#!/bin/bash
# ad hoc batch of the round (run via gpurun): same-box A/B of two library builds on configs 2 / 2b + the latency layout parity tests (LIBS="libA.so libB.so")
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06z
LIBS="${LIBS:-libpre_w2.so libdspi_mi355x.so}"
for c in 2 2b; do echo "== config $c"; bash tools/ab_bench.sh "$LIBS" 2 --config $c --no-side-runs; done > gpurun_out/r06z/ab.txt 2>&1
cat gpurun_out/r06z/ab.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "latency or config2 or one_stream or paired or spdif or one_packet or small_host" -n 4 2>&1 | tail -3
