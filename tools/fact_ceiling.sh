#!/bin/bash
# What ANY scheduler of the factorisation's tiles could reach (timing-only diagnostics build ab_libs/ci_diag.so = cholinv.hip with
# -DGPBO_DIAGNOSTICS; WRONG results in modes 1 and 2): tools/fact_ceiling.sh [N ...]
#   shipping library | diagnostics build, plan as shipped | GPBO_CI_CEILING=1: NEAR + PAIR launches without their filler tiles (the
#   critical path) | GPBO_CI_CEILING=2: every update tile in one launch, no dependencies, no PAIR workgroups (the work)
for n in ${@:-4096 8192}; do
  echo "shipping:        $(timeout -k 10 200 python tools/bench_factorise.py $n 2>/dev/null | tail -1)"
  echo "diagnostics:     $(GPBO_LIB=$PWD/ab_libs/ci_diag.so timeout -k 10 200 python tools/bench_factorise.py $n 2>/dev/null | tail -1)"
  echo "critical path:   $(GPBO_LIB=$PWD/ab_libs/ci_diag.so GPBO_CI_CEILING=1 timeout -k 10 200 python tools/bench_factorise.py $n 2>/dev/null | tail -1)"
  echo "work, 1 launch:  $(GPBO_LIB=$PWD/ab_libs/ci_diag.so GPBO_CI_CEILING=2 timeout -k 10 200 python tools/bench_factorise.py $n 2>/dev/null | tail -1)"
done
