"""Match the cholinv_kernel launches of a rocprofv3 kernel trace with the launch plan (gpbo_cholinv_plan) and say where
the time of one factorisation goes: python tools/cholinv_trace.py N trace.csv   (written by tools/cholinv_trace.sh)"""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cholinv_sim import get_plan, PANEL, SMALL, BIG

N = int(sys.argv[1])
Np = (N + 127) // 128 * 128
plan = get_plan(Np)
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ci = [r for r in rows if "cholinv_kernel" in r["Kernel_Name"]]
nrep = len(ci) // len(plan)
assert nrep * len(plan) == len(ci), (len(ci), len(plan))
others = collections.defaultdict(float)
for r in rows:
    if "cholinv_kernel" not in r["Kernel_Name"]:
        others[r["Kernel_Name"].split("(")[0][-40:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / nrep
last = ci[-len(plan):]  # the last factorisation of the run
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
gaps = 0.0
for i, (r, l) in enumerate(zip(last, plan)):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k0 = {PANEL: "panel", SMALL: "small", BIG: "big"}[int(l[0][0])]
    if k0 != "panel":
        k0 += ("-narrow" if int(l[0][6]) - int(l[0][5]) == 64 else "-near") + f"(K={int(l[0][4])})"
    far = int(l[2][1])
    key = k0 + ("+far" if far else "")
    agg[key][0] += 1; agg[key][1] += dur; agg[key][2] += far
    if i:
        gaps += (int(r["Start_Timestamp"]) - int(last[i - 1]["End_Timestamp"])) / 1e3
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"N={N}: {len(plan)} launches, span {span/1e3:.3f} ms, sum of durations {sum(v[1] for v in agg.values())/1e3:.3f} ms, gaps {gaps/1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} launches {v[0]:4d}  total {v[1]/1e3:7.3f} ms  mean {v[1]/v[0]:7.2f} us  far tiles/launch {v[2]/v[0]:7.1f}")
for k, v in sorted(others.items(), key=lambda kv: -kv[1]):
    print(f"  other: {k:40s} {v/1e3:7.3f} ms per factorisation")
if "-v" in sys.argv:
    for i, (r, l) in enumerate(zip(last, plan)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(i, [int(x) for x in l[0][:2]], "far", int(l[2][1]), f"{dur:.2f} us")
