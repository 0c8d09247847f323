#!/bin/bash
# on the GPU box: cycle stamps inside the update tiles (timing build ab_libs/cistamp.so): args = lines of "KIND K NTILES [Np]"
export GPBO_LIB=$PWD/ab_libs/cistamp.so   # the installed library is never touched (_lib.LIB_PATH)
for cfg in "$@"; do echo "== $cfg"; GPBO_CI_STAMPS=1 timeout -k 10 120 python tools/ci_tile_one.py $cfg 2>&1 | grep -i "stamps" | tail -1; done
