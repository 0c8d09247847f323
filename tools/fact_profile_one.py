"""Three factorisations at one size, for a kernel trace: python tools/fact_profile_one.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
N = int(sys.argv[1])
X, y, Xs, ls = make_problem(N, 512, 16)
gp = DeviceGP()
Xd, yd = gp._dev(X), gp._dev(y)
for _ in range(3):
    gp.factorise(Xd, yd, ls, check=False)
torch.cuda.synchronize()
