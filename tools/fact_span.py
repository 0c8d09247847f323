"""From a rocprofv3 kernel trace of bench.py: GPU-side span of one factorisation (kxx ... uv) against the sum of
its kernel durations, i.e. how much of the factorisation is dispatch gaps.  usage: fact_span.py t_kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
spans = []
cur = None
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "kxx_kernel" in n:
        cur = dict(start=s, busy=0, n=0, by={})
    if cur is not None:
        cur["busy"] += e - s
        cur["n"] += 1
        key = n.split("(")[0].split("::")[-1][:28]
        cur["by"][key] = cur["by"].get(key, 0) + (e - s)
        if "uv_kernel" in n and "utv" not in n:
            cur["end"] = e
            spans.append(cur)
            cur = None
for sp in spans[-3:]:
    print(f"launches {sp['n']}, span {(sp['end']-sp['start'])/1e3:.1f} us, kernel time {sp['busy']/1e3:.1f} us")
    print("   ", {k: round(v / 1e3, 1) for k, v in sorted(sp["by"].items(), key=lambda kv: -kv[1])})
