"""Time of the ARD likelihood grid (2,500 cells, d=2) at several N: python tools/bench_ard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

a1, a2 = np.linspace(0.05, 3.0, 50), np.linspace(0.05, 3.0, 50)
cells = np.stack(np.meshgrid(a1, a2, indexing="ij"), -1).reshape(-1, 2)
gp = DeviceGP()
for N in (32, 64, 128, 176, 200, 512, 1024):
    X, y, _, _ = make_problem(N, 8, 2)
    gp.nlml_grid(X, y, cells[:50])
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = gp.nlml_grid(X, y, cells)
    dt = time.perf_counter() - t
    print(f"N={N}: 2,500 cells in {dt*1e3:.2f} ms  (finite cells: {int(np.isfinite(out).sum())})", flush=True)
