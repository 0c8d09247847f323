import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
N = 4096
X, y, Xs, ls = make_problem(N + 40, 512, 8)
gp = DeviceGP()
Xd, yd = gp._dev(X), gp._dev(y)
for _ in range(3):
    gp.factorise(Xd[:N], yd[:N], ls, check=False)
torch.cuda.synchronize()
ts = []
for i in range(N, N + 40):
    torch.cuda.synchronize()
    t = time.perf_counter()
    gp.append(Xd[i], yd[i:i + 1], check=False)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "allocated", torch.cuda.memory_allocated() / 1e6)
