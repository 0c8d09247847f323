"""PCIe-inclusive rate of one BO step at BASELINE config 2 (d=8, N=512, M=2^20): candidates and observations start
as host NumPy arrays and the dense mu / sigma / acquisition the reference exposes come back to the host.
bench.py's `value` is the HBM-resident rate; this script gives the number quoted beside it in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

X, y, Xs, ls = make_problem(512, 1 << 20, 8)
gp = DeviceGP()
def step(dense):
    gp.factorise(X, y, ls, check=False)
    r = gp.score(Xs, dense=dense)                       # Xs is a host array: H2D copy inside
    if dense:
        return r.best_idx, r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    return r.best_idx
for dense in (False, True):
    for _ in range(3):
        step(dense)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        out = step(dense)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"host-resident inputs, dense outputs back={dense}: {dt*1e3:.2f} ms per step -> {(1<<20)/dt:.3e} candidates/s")
