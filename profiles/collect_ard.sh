#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the ARD likelihood grid alone (three calls of 2,500 cells at N, tools/ard_profile_one.py):
# rocprofv3 kernel trace + stats, then PMC passes (separate runs); per-call totals -> gpurun_out/prof_TAG/summary.txt
# Usage: bash profiles/collect_ard.sh TAG N [d] [likelihood]
set -uo pipefail
TAG=$1; N=$2; D=${3:-2}; MODE=${4:-reference}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/ard_profile_one.py $N $D $MODE > $OUT/trace.log 2>&1
echo "trace rc=$?"
if [ "${TRACE_ONLY:-0}" != "1" ]; then
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
  name=$(echo $pass | awk '{print $1}')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $REPO/tools/ard_profile_one.py $N $D $MODE > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
fi
cd $REPO && python3 profiles/summarise_ard.py $OUT $N "" $D | tee $OUT/summary.txt
