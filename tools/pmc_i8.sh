#!/bin/bash
# PMC passes over tools/bench_i8.py -> per-kernel means
TAG=${1:-a}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_i8_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  name=$(echo $pass | awk '{print $1}')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o pmc -- python3 $REPO/tools/bench_i8.py "$@" > $OUT/$name.log 2>&1
  echo "pmc $name rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    if "i8" in k or "slices" in k:
        for c,v in sorted(agg[k].items()):
            print(k.ljust(42), c.ljust(28), len(v), "%.6g" % (sum(v)/len(v)))
PY
