#!/bin/bash
# A/B timing of two builds of the library on the same box: tools/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for l in $A $B; do
    printf "%-28s" $(basename $l); QUICK=1 DSPI_LIB=$PWD/$l timeout 100 python tools/ablate.py 2>&1 | tail -1
  done
done
