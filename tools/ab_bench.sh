#!/bin/bash
# A/B timing of library builds through bench.py on the same box (box-to-box spread is +-4 %, so never compare across gpurun calls):
#   tools/ab_bench.sh "libA.so libB.so ..." [rounds] [bench args...]      libs relative to dspi_amd/csrc/
LIBS=$1; N=${2:-2}; shift 2
for i in $(seq $N); do
  for l in $LIBS; do
    printf "%-26s" $l
    DSPI_LIB=$PWD/dspi_amd/csrc/$l timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms/step  kernel %.3f ms  %.3e %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['unit']))"
  done
done
