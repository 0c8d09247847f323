"""Time of the farthest-point order of the observations (gpbo_fps_order_f64: selection + gather) the bound route's
factorisation starts with: python tools/fps_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
for N, d in ((2048, 8), (4096, 8), (8192, 8), (8192, 16), (8192, 4), (6000, 6)):
    X, y, Xs, ls = make_problem(N, 512, d)
    gp = DeviceGP()
    Xd, yd = gp._dev(X), gp._dev(y)
    ta, tf = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        gp.factorise(Xd, yd, ls, check=False)
        torch.cuda.synchronize(); ta.append(time.perf_counter() - t)
        torch.cuda.synchronize(); t = time.perf_counter()
        gp.factorise(Xd, yd, ls, check=False, order="fps")
        torch.cuda.synchronize(); tf.append(time.perf_counter() - t)
    print(f"N={N} d={d}: factorise {min(ta)*1e3:.3f} ms, with the farthest-point order ({gp.bound_prefix()} members) "
          f"{min(tf)*1e3:.3f} ms: order {1e3*(min(tf)-min(ta)):.3f} ms", flush=True)
