#!/bin/bash
# on the GPU box: kernel trace of three factorisations at N, every cholinv_kernel launch matched with its plan entry
N=${1:-4096}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ci_$N; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/fact_profile_one.py $N > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/cholinv_trace.py $N $OUT/t_kernel_trace.csv | tee $OUT/summary.txt
