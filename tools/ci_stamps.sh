#!/bin/bash
# on the GPU box: cycle stamps of the PAIR workgroups (timing build ab_libs/cistamp.so = cholinv.hip with -DGPBO_DIAGNOSTICS)
export GPBO_LIB=$PWD/ab_libs/cistamp.so   # the installed library is never touched (_lib.LIB_PATH)
for N in "$@"; do GPBO_CI_STAMPS=1 timeout -k 10 120 python tools/fact_profile_one.py $N 2>&1 | grep -i "stamps" | tail -1; done
