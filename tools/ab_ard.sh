#!/bin/bash
# Same-box A/B of library variants on the likelihood grid: tools/ab_ard.sh <reps> <lib1.so> <lib2.so> ...
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    GPBO_LIB=$PWD/$v timeout -k 10 200 python tools/ard_time.py $ARD_SIZES 2>/dev/null | tail -1
  done
done
