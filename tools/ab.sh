#!/bin/bash
# Same-box A/B of library variants: tools/ab.sh <reps> <lib1.so> <lib2.so> ... ; each variant is loaded through GPBO_LIB
# (the installed libgpbo.so is never overwritten) and bench.py is run.
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    export GPBO_LIB=$PWD/$v   # the installed library is never touched (_lib.LIB_PATH)
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['ms_per_step_scoring_only'],4), round(d['ms_per_step']-d['ms_per_step_scoring_only'],4))"
  done
done
