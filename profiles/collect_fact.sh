#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the factorisation alone (three gpbo_factorise_f64 calls at N, tools/fact_profile_one.py):
# rocprofv3 kernel trace + stats, then PMC passes (separate runs); totals per factorisation -> gpurun_out/prof_TAG/summary.
# Usage: bash profiles/collect_fact.sh TAG N
set -uo pipefail
TAG=$1; N=$2
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/fact_profile_one.py $N > $OUT/trace.log 2>&1
echo "trace rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
  name=$(echo $pass | awk '{print $1}')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $REPO/tools/fact_profile_one.py $N > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
cd $REPO && python3 profiles/summarise_fact.py $OUT $N | tee $OUT/summary.txt
