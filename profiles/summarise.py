#!/usr/bin/env python3
"""Turn the raw output of profiles/collect.sh (gpurun_out/prof_TAG/) into the committed summaries:
    profiles/<out>_kernel_stats.csv   copy of rocprofv3's kernel stats
    profiles/<out>_pmc_summary.csv    per-launch means of every counter, per kernel
    profiles/pmc_sigma_acq.json       HBM bytes per launch of the dominant kernel (read by bench.py)
usage: python profiles/summarise.py gpurun_out/prof_TAG r01 [bench_line.json]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(here, f"{out}_kernel_stats.csv"))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ("sigma_acq_kernel", "kstar_mu_kernel", "potrf_diag_kernel", "gemm_f64_kernel", "kxx_kernel", "utv_kernel",
        "uv_kernel")
with open(os.path.join(here, f"{out}_pmc_summary.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc passes (one pass per counter group, profiles/collect.sh), bench.py --steps 5 --warmup 2,\n")
    fo.write("# MI355X; per-launch means; a sigma/kstar launch = one chunk of 2^17 candidates, N=512, d=8\n")
    fo.write("kernel,counter,launches,mean_per_launch\n")
    for k in keep:
        for c, v in sorted(agg.get(k, {}).items()):
            fo.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")

s = agg["sigma_acq_kernel"]
fetch = sum(s["FETCH_SIZE"]) / len(s["FETCH_SIZE"])
write = sum(s["WRITE_SIZE"]) / len(s["WRITE_SIZE"])
N, cands = 512, 131072
js = {
    "kernel": "sigma_acq_kernel",
    "source": f"profiles/{out}_pmc_summary.csv",
    "FETCH_SIZE_KiB": fetch,
    "WRITE_SIZE_KiB": write,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "candidates_per_launch": cands,
    "algorithmic_bytes_per_launch": cands * 8 * (N + 5),  # K*^T slab once + mu partial slices read + outputs
}
json.dump(js, open(os.path.join(here, "pmc_sigma_acq.json"), "w"), indent=1)
if len(sys.argv) > 3:
    line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
    open(os.path.join(here, f"{out}_bench_line.json"), "w").write(line)
print(open(os.path.join(here, f"{out}_pmc_summary.csv")).read())
