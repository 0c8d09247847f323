#!/bin/bash
# Same-box A/B of library variants on the ARD likelihood grid over a list of sizes: tools/ab_ard_n.sh "<N ...>" <lib1.so> <lib2.so> ...
NS=$1; shift
for d in 2 8; do
  for v in "$@" "$@"; do
    GPBO_LIB=$PWD/$v ARD_D=$d python tools/ard_time.py $NS 2>/dev/null | tail -n 1
  done
done
