#!/usr/bin/env python3
"""Per-call totals from profiles/collect_ard.sh (three calls of 2,500 cells per run): kernel time by kernel, counters of the
grid kernel per call.  usage: python profiles/summarise_ard.py gpurun_out/prof_TAG N [out_prefix]
With out_prefix: writes profiles/<out_prefix>_kernel_stats.csv (rocprofv3's stats) and <out_prefix>_pmc_summary.csv."""
import collections, csv, glob, os, shutil, sys
src, N = sys.argv[1], int(sys.argv[2])
out = (sys.argv[3] or None) if len(sys.argv) > 3 else None
d = int(sys.argv[4]) if len(sys.argv) > 4 else 2
here = os.path.dirname(os.path.abspath(__file__))
REPS, G = 3, 2500
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
t = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = short(r["Kernel_Name"])
    t[k][0] += 1; t[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
lines = [f"# N={N}, {G} cells: per call (means over {REPS} calls of DeviceGP.nlml_grid), MI355X, rocprofv3 kernel trace"]
for k, v in sorted(t.items(), key=lambda kv: -kv[1][1]):
    if "nlml" in k:
        lines.append(f"kernel_time,{k},launches={v[0] / REPS:.1f},us={v[1] / REPS:.1f}")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]) / REPS
main = [k for k in t if "nlml_fused" in k or "nlml_grid_kernel" in k or "nlml_wave_kernel" in k]
main = max(main, key=lambda k: t[k][1]) if main else None
if main:
    us = t[main][1] / REPS
    flop = G * N ** 3 / 3.0
    lines.append(f"derived,{main},flop_per_call_G*N^3/3,{flop:.6g}")
    lines.append(f"derived,{main},TFLOP/s,{flop / us / 1e6:.4g}")
    lines.append(f"derived,{main},frac_of_78.6,{flop / us / 1e6 / 78.6:.4g}")
    c = agg.get(main, {})
    for cn, v in sorted(c.items()):
        lines.append(f"counter_sum_per_call,{main},{cn},{v:.6g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        lines.append(f"derived,{main},mfma_busy_share_of_cycles,{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0):.4g}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fabric = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        Nf = (N + 63) // 64 * 64
        alg = G * Nf * (Nf + 16) * 8.0      # the factor written once and read once (lower triangle, twice)
        # what the 64-column left-looking algorithm itself moves when only the current panel is on chip: every panel reads
        # the earlier columns of the rows at and below it once (own rows), and the factor is written once
        blocked = G * 8.0 * (sum((Nf + 16 - 64 * j) * 64 * j for j in range(Nf // 64)) + Nf * (Nf + 16) / 2.0)
        lines.append(f"derived,{main},fabric_bytes_per_call,{fabric:.6g}")
        lines.append(f"derived,{main},algorithmic_bytes_per_call_(factor_written_once_read_once),{alg:.6g}")
        lines.append(f"derived,{main},fabric_over_algorithmic,{fabric / alg:.4g}")
        lines.append(f"derived,{main},blocked_algorithm_bytes_per_call_(own_rows_read_once_per_panel_+_factor_written_once),{blocked:.6g}")
        lines.append(f"derived,{main},fabric_over_blocked_algorithm,{fabric / blocked:.4g}")
    if "TCC_HIT_sum" in c:
        lines.append(f"derived,{main},l2_hit_rate,{c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4g}")
print("\n".join(lines))
if out and main and "FETCH_SIZE" in agg.get(main, {}):
    import json
    sys.path.insert(0, here)
    from source_hash import kernel_source_hash
    path = os.path.join(here, "pmc_ard.json")
    try:
        shapes = json.load(open(path))
    except Exception:  # noqa: BLE001
        shapes = {}
    c = agg[main]
    Nf = (N + 63) // 64 * 64
    shapes[f"N={N},d={d},cells={G}"] = {
        "kernel": main, "source": f"profiles/{out}_pmc_summary.csv",
        "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1",
        "fabric_bytes_per_call": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
        "algorithmic_bytes_per_call": G * Nf * (Nf + 16) * 8.0,
        "algorithmic_bytes_definition": "the cell's factor (lower triangle) written once and read once",
        "blocked_algorithm_bytes_per_call": G * 8.0 * (sum((Nf + 16 - 64 * j) * 64 * j for j in range(Nf // 64)) + Nf * (Nf + 16) / 2.0),
        "mfma_busy_share_of_cycles": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c else None,
        "kernel_time_us_profiled": t[main][1] / REPS,
        "hash_key": "ard_wave" if "nlml_wave" in main else "ard",
        "kernel_source_hash": kernel_source_hash("ard_wave" if "nlml_wave" in main else "ard"),
    }
    json.dump(shapes, open(path, "w"), indent=1)
if out:
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(here, f"{out}_kernel_stats.csv"))
    open(os.path.join(here, f"{out}_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
