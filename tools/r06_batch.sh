#!/bin/bash
# ad hoc batch of the round (run via gpurun): probe13 + same-box A/B of latency-layout builds on configs 2 / 2b
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06r
(cd tools/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o probe13 probe13.hip && timeout 300 ./probe13) > gpurun_out/r06r/probe13.txt 2>&1
LIBS="${LIBS:-libpre_exp.so libdspi_mi355x.so}"
for c in 2 2b; do echo "== config $c"; bash tools/ab_bench.sh "$LIBS" 2 --config $c --no-side-runs; done > gpurun_out/r06r/ab.txt 2>&1
cat gpurun_out/r06r/probe13.txt gpurun_out/r06r/ab.txt
