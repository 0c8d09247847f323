"""Time of the bound route's observation subset (farthest-point sampling + gather + its own factorisation): python tools/fps_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
for N, d in ((4096, 8), (8192, 8), (8192, 4), (6000, 6)):
    X, y, Xs, ls = make_problem(N, 512, d)
    gp = DeviceGP()
    gp.factorise(X, y, ls)
    Np = gp.Np
    J = max(128, (Np // 16) // 128 * 128); J2 = 4 * J
    ts = []
    for _ in range(5):
        gp._bound_subset = None   # rebuild
        torch.cuda.synchronize(); t = time.perf_counter()
        gp._ensure_bound_subset(J, J2)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"N={N} d={d}: subset (FPS + gather + factor of {J2}) {min(ts)*1e3:.3f} ms", flush=True)
