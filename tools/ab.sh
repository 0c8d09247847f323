#!/bin/bash
# Same-box A/B of library variants: tools/ab.sh <reps> <lib1.so> <lib2.so> ... ; each variant is copied over
# libgpbo.so and bench.py is run; the original library is restored at the end.
reps=$1; shift
cp bayesian_optimisation_amd/libgpbo.so /tmp/libgpbo_orig.so
for rep in $(seq $reps); do
  for v in "$@"; do
    cp $v bayesian_optimisation_amd/libgpbo.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_scoring_only'],4), round(d['ms_per_step']-d['ms_per_step_scoring_only'],4))"
  done
done
cp /tmp/libgpbo_orig.so bayesian_optimisation_amd/libgpbo.so
