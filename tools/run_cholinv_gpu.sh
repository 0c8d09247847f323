#!/bin/bash
# on the GPU box: correctness of the fused factorisation, then its time beside the round-2 chain on the same box
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -k 10 420 python -m pytest tests/test_gpu_cholinv.py -x -q > $OUT/cholinv_test.log 2>&1; rc=$?
tail -15 $OUT/cholinv_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/bench_factorise.py 512 2048 4096 8192 2>&1 | tee $OUT/fact_new.log && \
GPBO_FACTOR_OLD=1 timeout -k 10 200 python tools/bench_factorise.py 512 2048 4096 8192 2>&1 | tee $OUT/fact_old.log
