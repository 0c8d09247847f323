#!/bin/bash
# on the GPU box: where the time of one factorisation goes (kernel trace of tools/fact_profile_one.py N, three factorisations)
N=${1:-8192}
OUT=$GRAFT_REPO_ROOT/gpurun_out/fact_$N; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/tools/fact_profile_one.py $N > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# classify GEMM launches by grid shape: phase = potrf (before the first init_w_kernel of a factorisation) or trtri (after)
agg = collections.defaultdict(lambda: [0, 0.0])
phase = "potrf"
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "kxx_kernel" in n: phase = "potrf"
    if "init_w_kernel" in n: phase = "trtri"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (phase, n[:28])
    agg[key][0] += 1; agg[key][1] += dur
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:6s} {k[1]:30s} launches/3 {v[0]/3:7.1f}  ms per factorisation {v[1]/3e3:7.3f}  {100*v[1]/tot:5.1f} %")
print("sum of kernel durations per factorisation: %.3f ms" % (tot / 3e3))
PY
