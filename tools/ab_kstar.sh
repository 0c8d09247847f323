#!/bin/bash
# Same-box A/B of library variants, K(X*,X) kernel view: tools/ab_kstar.sh <reps> <lib1.so> <lib2.so> ...
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    export GPBO_LIB=$PWD/$v   # the installed library is never touched (_lib.LIB_PATH)
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps ${AB_STEPS:-4} --warmup 1 $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'step', round(d['ms_per_step'],4), 'kstar', d['kstar_roofline']['avg_launch_ms'], d['kstar_roofline']['frac'], 'sigma', d['roofline']['avg_launch_ms'], 'argmax', d['argmax_index'])"
  done
done
