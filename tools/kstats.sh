#!/bin/bash
# on the GPU box: per-kernel average durations of bench.py under rocprofv3 (tag = $1, extra env via caller)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG/t_kernel_stats.csv")))[:4]:
    print("$TAG", r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
