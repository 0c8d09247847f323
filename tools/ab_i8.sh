#!/bin/bash
# Same-box A/B of library variants on the int8-sliced screen: tools/ab_i8.sh <reps> <lib1.so> <lib2.so> ...
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    export GPBO_LIB=$PWD/$v   # the installed library is never touched (_lib.LIB_PATH)
    echo -n "$v: "; timeout -k 10 200 python tools/bench_i8.py ${AB_N:-4096} 3 ${AB_MODE:-i8raw} 2>/dev/null | tail -1
  done
done
