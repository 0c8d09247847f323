#!/bin/bash
# Same-box A/B of library variants on the factorisation: tools/ab_fact.sh <reps> <lib1.so> <lib2.so> ...   (sizes: AB_SIZES)
reps=$1; shift
cp bayesian_optimisation_amd/libgpbo.so /tmp/libgpbo_orig.so
for rep in $(seq $reps); do
  for v in "$@"; do
    cp $v bayesian_optimisation_amd/libgpbo.so
    echo "$v: $(timeout -k 10 200 python tools/bench_factorise.py ${AB_SIZES:-2048 4096 8192} 2>/dev/null | grep 'N=' | tr '\n' ';')"
  done
done
cp /tmp/libgpbo_orig.so bayesian_optimisation_amd/libgpbo.so
