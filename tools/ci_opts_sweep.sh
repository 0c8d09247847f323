#!/bin/bash
# on the GPU box: factorisation time under different schedule options (GPBO_CI_OPTS = win,far_k,far_kind,defer+1)
for o in "$@"; do
  echo "== GPBO_CI_OPTS=$o"
  GPBO_CI_OPTS=$o timeout -k 10 120 python tools/bench_factorise.py $CI_SIZES 2>&1 | grep "N="
done
