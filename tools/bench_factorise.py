"""Time of gpbo_factorise_f64 alone (median of repeated groups): python tools/bench_factorise.py [N ...]
GPBO_FACTOR_OLD=1 selects the round-2 chain (potrf + trtri) for an A/B on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

for N in [int(a) for a in sys.argv[1:]] or [512]:
    X, y, Xs, ls = make_problem(N, 512, 8)
    gp = DeviceGP()
    Xd, yd = gp._dev(X), gp._dev(y)
    for _ in range(5):
        gp.factorise(Xd, yd, ls, check=False)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            gp.factorise(Xd, yd, ls, check=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / 10)
    print(f"N={N}: factorise median {np.median(ts)*1e3:.4f} ms  min {min(ts)*1e3:.4f} ms", flush=True)
