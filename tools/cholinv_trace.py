"""Match the cholinv_kernel launches of a rocprofv3 kernel trace with the launch plan (gpbo_cholinv_plan) and say where
the time of one factorisation goes: python tools/cholinv_trace.py N trace.csv   (written by tools/cholinv_trace.sh)"""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cholinv_sim import get_plan, SMALL, BIG, BIG256

N = int(sys.argv[1])
Np = (N + 127) // 128 * 128
L, T = get_plan(Np)
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ci = [r for r in rows if "cholinv_kernel" in r["Kernel_Name"]]
nrep = len(ci) // len(L)
assert nrep * len(L) == len(ci), (len(ci), len(L))
others = collections.defaultdict(float)
for r in rows:
    if "cholinv_kernel" not in r["Kernel_Name"]:
        others[r["Kernel_Name"].split("(")[0][-40:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / nrep
last = ci[-len(L):]  # the last factorisation of the run
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, (r, l) in enumerate(zip(last, L)):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t = T[l[2]:l[2] + l[3]]
    units = float((t[:, 2].astype(np.int64) * np.where(t[:, 0] == SMALL, 0.25, np.where(t[:, 0] == BIG, 1, 2))).sum()) / 128
    key = "pair" if l[1] > 0 else "near"
    if l[1] > 0 and l[3] > 0:
        key += "+fill(<=256 units)" if units <= 256 else "+fill(<=512 units)" if units <= 512 else "+fill(>512 units)"
    agg[key][0] += 1; agg[key][1] += dur; agg[key][2] += l[3]; agg[key][3] += units
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"N={N}: {len(L)} launches, span {span/1e3:.3f} ms, sum of durations {sum(v[1] for v in agg.values())/1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:26s} launches {v[0]:4d}  total {v[1]/1e3:7.3f} ms  mean {v[1]/v[0]:7.2f} us  tiles/launch {v[2]/v[0]:7.1f}  "
          f"128^3-units/launch {v[3]/v[0]:7.1f}")
for k, v in sorted(others.items(), key=lambda kv: -kv[1]):
    print(f"  other: {k:40s} {v/1e3:7.3f} ms per factorisation")
if "-v" in sys.argv:
    for i, (r, l) in enumerate(zip(last, L)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        t = T[l[2]:l[2] + l[3]]
        print(i, list(l), "K:", sorted(set(t[:, 2].tolist())), f"{dur:.2f} us")
