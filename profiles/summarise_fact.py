#!/usr/bin/env python3
"""Per-factorisation totals from profiles/collect_fact.sh (three factorisations per run): kernel time by kernel, counters
of cholinv_kernel summed over its launches.  usage: python profiles/summarise_fact.py gpurun_out/prof_TAG N [out_prefix]
With out_prefix: writes profiles/<out_prefix>_kernel_stats.csv (rocprofv3's stats) and <out_prefix>_pmc_summary.csv."""
import collections, csv, glob, os, shutil, sys
src, N = sys.argv[1], int(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else None
here = os.path.dirname(os.path.abspath(__file__))
REPS = 3
rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
t = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    t[k][0] += 1; t[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
lines = [f"# N={N}: per factorisation (means over {REPS} calls of gpbo_factorise_f64), MI355X, rocprofv3 kernel trace"]
tot = sum(v[1] for v in t.values()) / REPS
for k, v in sorted(t.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"kernel_time,{k},launches={v[0] / REPS:.1f},us={v[1] / REPS:.1f},share={v[1] / REPS / tot:.3f}")
lines.append(f"kernel_time,TOTAL,,us={tot:.1f},")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]) / REPS
for k in ("cholinv_kernel", "kxx_kernel", "transpose_w_kernel"):
    for c, v in sorted(agg.get(k, {}).items()):
        lines.append(f"counter_sum_per_factorisation,{k},{c},{v:.6g}")
c = agg.get("cholinv_kernel", {})
if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
    flop = 2.0 * N ** 3 / 3.0
    mfma_cycles_needed = flop / 2048.0 * 64.0 / 1024.0   # fp64 16x16x4 MFMAs of 64 cycles over 1024 SIMDs
    lines.append(f"derived,cholinv_kernel,mfma_busy_cycles_per_simd,{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0:.6g}")
    lines.append(f"derived,cholinv_kernel,mfma_cycles_per_simd_of_2N^3/3_flop,{mfma_cycles_needed:.6g}")
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    lines.append(f"derived,cholinv_kernel,fabric_bytes_per_factorisation,{(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024:.6g}")
print("\n".join(lines))
if out:
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(here, f"{out}_kernel_stats.csv"))
    open(os.path.join(here, f"{out}_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
