#!/usr/bin/env python3
"""Turn the raw output of profiles/collect.sh (gpurun_out/prof_TAG/) into the committed summaries:
    profiles/<out>_kernel_stats.csv   copy of rocprofv3's kernel stats
    profiles/<out>_pmc_summary.csv    per-launch means of every counter, per kernel
    profiles/pmc_sigma_acq.json       HBM bytes per launch of the dominant kernel (read by bench.py)
usage: python profiles/summarise.py gpurun_out/prof_TAG r02 [bench_line.json] [N d dtype candidates_per_launch]
pmc_sigma_acq.json is keyed by workload shape ("N=4096,d=8,dtype=f64,candidates_per_launch=131072"); bench.py
replays the entry that matches the shape it runs (counters cannot be read from inside an un-profiled run)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_source_hash  # noqa: E402

src, out = sys.argv[1], sys.argv[2]
N, d, dtype, cands = 4096, 8, "f64", 131072
if len(sys.argv) > 4:
    N, d, dtype, cands = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
w = {"f32": 4, "i8": 5, "i8c": 3}.get(dtype, 8)   # bytes of K* per entry (int8: five slices, coarse screen: three)
kern = {"f32": "sigma_acq_f32_kernel", "i8": "sigma_i8_kernel", "i8c": "sigma_i8c_kernel"}.get(dtype, "sigma_acq_kernel")
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(here, f"{out}_kernel_stats.csv"))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ("sigma_acq_kernel", "kstar_mu_kernel", "sigma_acq_f32_kernel", "sigma_i8_kernel", "sigma_i8c_kernel", "kstar_slices_kernel", "kstar_mu_mfma_kernel", "bound_select_kernel",
        "split_finish_kernel", "u_slices_kernel", "u_colscale_kernel", "potrf_diag_kernel", "gemm_f64_kernel", "kxx_kernel", "utv_kernel",
        "uv_kernel", "cholinv_kernel", "transpose_w_kernel", "fps_coop_kernel", "gather_obs_kernel", "qei_kernel")
with open(os.path.join(here, f"{out}_pmc_summary.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc passes (one pass per counter group, profiles/collect.sh), bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also,\n")
    fo.write(f"# MI355X; per-launch means; a sigma/kstar launch = one chunk of {cands} candidates, N={N}, d={d}, {dtype}\n")
    fo.write("kernel,counter,launches,mean_per_launch\n")
    for k in keep:
        for c, v in sorted(agg.get(k, {}).items()):
            fo.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")

s = agg[kern]
if "FETCH_SIZE" not in s:   # TRACE_ONLY collection: kernel stats only
    if len(sys.argv) > 3:
        line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
        open(os.path.join(here, f"{out}_bench_line.json"), "w").write(line)
    print(f"{out}: kernel stats only (no PMC passes)")
    sys.exit(0)
fetch = sum(s["FETCH_SIZE"]) / len(s["FETCH_SIZE"])
write = sum(s["WRITE_SIZE"]) / len(s["WRITE_SIZE"])
path = os.path.join(here, "pmc_sigma_acq.json")
try:
    shapes = json.load(open(path))
    if "kernel" in shapes:  # round-1 layout (one un-keyed entry, N=512)
        shapes = {"N=512,d=8,dtype=f64,candidates_per_launch=131072": shapes}
except Exception:  # noqa: BLE001
    shapes = {}
shapes[f"N={N},d={d},dtype={dtype},candidates_per_launch={cands}"] = {
    "kernel": kern,
    "source": f"profiles/{out}_pmc_summary.csv",
    "FETCH_SIZE_KiB": fetch,
    "WRITE_SIZE_KiB": write,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "candidates_per_launch": cands,
    "algorithmic_bytes_per_launch": cands * w * (N + 5),  # K*^T slab once + mu partial slices read + outputs
    # the kernel sources these counters were collected on (profiles/source_hash.py): bench.py flags the replay as stale
    # when today's sources differ
    "kernel_source_hash": kernel_source_hash(dtype),
}
ks = agg.get({"i8": "kstar_slices_kernel", "i8c": "kstar_slices_kernel"}.get(dtype, "kstar_mu_kernel"), {})
if "SQ_INSTS_VALU" in ks:
    # the K(X*,X) build beside its stores: vector instructions issued per launch (wave instructions; x64 = lane instructions)
    shapes[f"N={N},d={d},dtype={dtype},candidates_per_launch={cands}"]["kstar_valu_wave_instructions_per_launch"] = \
        sum(ks["SQ_INSTS_VALU"]) / len(ks["SQ_INSTS_VALU"])
qk = agg.get("qei_kernel", {})
if "FETCH_SIZE" in qk and "WRITE_SIZE" in qk:
    # the qEI stage of BASELINE config 5: HBM bytes per launch (its algorithmic bytes: the rows of V, 8 N per candidate)
    shapes[f"N={N},d={d},dtype={dtype},candidates_per_launch={cands}"]["qei_hbm_bytes_per_launch"] = \
        sum(qk["FETCH_SIZE"]) / len(qk["FETCH_SIZE"]) * 1024 * 2 + sum(qk["WRITE_SIZE"]) / len(qk["WRITE_SIZE"]) * 1024
json.dump(shapes, open(path, "w"), indent=1)
if len(sys.argv) > 3:
    line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
    open(os.path.join(here, f"{out}_bench_line.json"), "w").write(line)
print(open(os.path.join(here, f"{out}_pmc_summary.csv")).read())
