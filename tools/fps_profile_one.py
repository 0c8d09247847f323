import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
X, y, Xs, ls = make_problem(8192, 512, 16)
gp = DeviceGP()
for _ in range(3):
    gp.factorise(X, y, ls, order="fps")
torch.cuda.synchronize()
