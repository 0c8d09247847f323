#!/bin/bash
# Same-box A/B of library variants on the factorisation: tools/ab_fact.sh <reps> <lib1.so> <lib2.so> ...   (sizes: AB_SIZES)
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    export GPBO_LIB=$PWD/$v   # the installed library is never touched (_lib.LIB_PATH)
    echo "$v: $(timeout -k 10 200 python tools/bench_factorise.py ${AB_SIZES:-2048 4096 8192} 2>/dev/null | grep 'N=' | tr '\n' ';')"
  done
done
