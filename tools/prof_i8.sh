#!/bin/bash
# kernel trace of tools/bench_i8.py on the GPU box -> gpurun_out/prof_i8_$1
TAG=${1:-a}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_i8_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_i8.py "$@" > $OUT/log.txt 2>&1
echo rc=$?
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"], "%.3f ms avg" % (float(r["AverageNs"])/1e6), r["Percentage"])
PY
tail -3 $OUT/log.txt
