"""Time of one appended observation (gpbo_append_f64) against a full factorisation of the same size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

for N in (512, 2048, 4096, 8192):
    d = 8
    X, y, Xs, ls = make_problem(N + 40, 512, d)
    gp = DeviceGP()
    Xd, yd = gp._dev(X), gp._dev(y)
    for _ in range(2):
        gp.factorise(Xd[:N], yd[:N], ls, check=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        gp.factorise(Xd[:N], yd[:N], ls, check=False)
    torch.cuda.synchronize()
    t_full = (time.perf_counter() - t) / 3
    for i in range(N, N + 8):
        gp.append(Xd[i], yd[i:i + 1], check=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(N + 8, N + 40):
        gp.append(Xd[i], yd[i:i + 1], check=False)
    torch.cuda.synchronize()
    t_app = (time.perf_counter() - t) / 32
    print(f"N={N}: factorise {t_full*1e3:.3f} ms, append {t_app*1e3:.3f} ms per row ({t_full/t_app:.0f}x)", flush=True)
