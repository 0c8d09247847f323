#!/bin/bash
# on the GPU box: PMC passes over one update-tile configuration: tools/ci_tile_pmc.sh KIND K NTILES [Np]
ARGS="$*"; TAG=$(echo $ARGS | tr ' ' '_')
OUT=$GRAFT_REPO_ROOT/gpurun_out/ci_pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | awk '{print $1}')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ci_tile_one.py $ARGS > $OUT/$name.log 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/ci_tile_one.py $ARGS > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'cholinv_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
durs = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open("$OUT/trace/t_kernel_trace.csv")) if 'cholinv_kernel' in r['Kernel_Name']]
print("config $ARGS: launch %.1f us (n=%d)" % (sum(durs) / len(durs), len(durs)))
for c, v in sorted(agg.items()): print("  %-28s mean %.6g" % (c, sum(v) / len(v)))
PY
