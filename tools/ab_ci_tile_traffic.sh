#!/bin/bash
# Where do an update tile's fixed 14 us go?  Same-box timing of tile launches (tools/bench_ci_jobs.py) with the C tile's stores
# and / or loads compiled out - timing-only builds, wrong results by design:
#   bash tools/build_variant.sh ci_nostore cholinv "-DGPBO_CI_DIAG_NO_STORE"
#   bash tools/build_variant.sh ci_nocload cholinv "-DGPBO_CI_DIAG_NO_CLOAD"
#   bash tools/build_variant.sh ci_noc cholinv "-DGPBO_CI_DIAG_NO_CLOAD -DGPBO_CI_DIAG_NO_STORE"
#   cp bayesian_optimisation_amd/libgpbo.so ab_libs/ci_base.so
for v in ab_libs/ci_base.so ab_libs/ci_nostore.so ab_libs/ci_nocload.so ab_libs/ci_noc.so; do
  export GPBO_LIB=$PWD/$v   # the installed library is never touched (_lib.LIB_PATH)
  echo "== $v"; CI_NTILES=1,64,256,512 timeout -k 10 120 python tools/bench_ci_jobs.py 4096 2>/dev/null | grep "big128 K=256\|big128 K=128"
done
