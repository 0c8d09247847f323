#!/bin/bash
# on the GPU box: cycle stamps of the PAIR workgroups (timing build ab_libs/cistamp.so = cholinv.hip with -DGPBO_DIAGNOSTICS)
cp bayesian_optimisation_amd/libgpbo.so /tmp/libgpbo_orig.so
cp ab_libs/cistamp.so bayesian_optimisation_amd/libgpbo.so
for N in "$@"; do GPBO_CI_STAMPS=1 timeout -k 10 120 python tools/fact_profile_one.py $N 2>&1 | grep -i "stamps" | tail -1; done
cp /tmp/libgpbo_orig.so bayesian_optimisation_amd/libgpbo.so
