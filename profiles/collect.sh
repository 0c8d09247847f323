#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + PMC passes for bench.py; output under gpurun_out/prof_$TAG.
# Usage: [TRACE_ONLY=1] bash profiles/collect.sh TAG [bench args...]
set -uo pipefail
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-also $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
if [ "${TRACE_ONLY:-0}" = "1" ]; then exit 0; fi
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | awk '{print $1}')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
ls -R $OUT | head -50
