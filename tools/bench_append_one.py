"""One size of tools/bench_append.py for profiling: python tools/bench_append_one.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
N = int(sys.argv[1])
X, y, Xs, ls = make_problem(N + 24, 512, 8)
gp = DeviceGP()
Xd, yd = gp._dev(X), gp._dev(y)
gp.factorise(Xd[:N], yd[:N], ls, check=False)
for i in range(N, N + 24):
    gp.append(Xd[i], yd[i:i + 1], check=False)
torch.cuda.synchronize()
