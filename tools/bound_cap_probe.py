"""Time of the exact-bound route on the headline workload by acquisition and by the survivor cap (DeviceGP.screen_cap):
python tools/bound_cap_probe.py [N] [log2 M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import ard_length_scales, rff_objective, sobol_points

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 21)
d = 8
dev = torch.device("cuda", 0)
ls = ard_length_scales(d)
X = sobol_points(0, N, d)
y = rff_objective(X, ls)
Xs = sobol_points(N, M, d)
gp = DeviceGP(dev)
Xd, yd, Xsd = gp._dev(X), gp._dev(y), gp._dev(Xs)
gp.factorise(Xd, yd, ls, check=False, order="fps")
fb = float(np.min(y))
for cap in (0, M // 16):
    gp.screen_cap = cap
    for name, kw in (("lcb 1", dict(acquisition="lcb", explore=1.0)), ("lcb 4", dict(acquisition="lcb", explore=4.0)),
                     ("lcb 10", dict(acquisition="lcb", explore=10.0)), ("lcb 25", dict(acquisition="lcb", explore=25.0)), ("lcb 100", dict(acquisition="lcb", explore=100.0)),
                     ("ei", dict(acquisition="ei", f_best=fb, xi=0.0))):
        r = gp.score_bound(Xsd, **kw)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            r = gp.score_bound(Xsd, **kw)
        ms = (time.perf_counter() - t) / 3 * 1e3
        s = gp.last_screen
        print(f"cap={cap or 'default'} {name}: {ms:8.2f} ms  survivors {s.get('survivors')} rescored {s.get('rescored')} fallback {s.get('fallback')} idx {r.best_idx}", flush=True)
