#!/bin/bash
# on the GPU box: one rocprofv3 --pmc pass over bench.py, per-kernel means. usage: pmc.sh TAG "COUNTERS" [bench args]
TAG=$1; CNT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/p_counter_collection.csv")):
    k = r['Kernel_Name']
    short = 'sigma' if 'sigma_acq' in k else ('kstar' if 'kstar' in k else ('potrf_diag' if 'potrf_diag' in k else None))
    if short: agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    for c, v in d.items():
        print("$TAG", k, c, 'n=%d'%len(v), 'mean=%.6g'%(sum(v)/len(v)))
PY
