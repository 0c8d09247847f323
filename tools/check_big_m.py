import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
X, y, Xs, ls = make_problem(100, 1 << 24, 8)
gp = DeviceGP().factorise(X, y, ls)
Xsd = gp._dev(Xs)
t = time.perf_counter(); r = gp.score(Xsd, dense=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
acq = r.acq.cpu().numpy()
print("M=2^24 N=100:", dt, r.best_idx, int(np.flatnonzero(acq == acq.max())[0]), r.nan_count, r.best_val == acq.max())
r2 = gp.score(Xsd[(1 << 24) - 1000:], idx_offset=(1 << 24) - 1000)
print("tail shard:", r2.best_idx, (1 << 24) - 1000 + int(np.argmax(acq[-1000:])))
