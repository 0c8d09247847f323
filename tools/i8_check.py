# int8-sliced variance screen: accuracy against the oracle / the fp64 kernels, then timing at the headline shape
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O
for N, M, d, chunk in [(100, 1000, 3, 512), (256, 2048, 8, 1024), (700, 5000, 8, 2048), (2048, 4096, 8, 4096)]:
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score_i8(Xs, dense=True, idx_offset=3)
    r64 = gp.score(Xs, dense=True, idx_offset=3)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    print(N, M, d, "dmu vs fp64 kernels", np.abs(r.mu.cpu().numpy() - r64.mu.cpu().numpy()).max(),
          "dsigma vs fp64 kernels %.3g" % np.abs(r.sigma.cpu().numpy() - r64.sigma.cpu().numpy()).max(),
          "vs oracle %.3g" % np.abs(r.sigma.cpu().numpy() - sig_o).max(), "idx", r.best_idx == r64.best_idx, gp.last_screen, flush=True)
if len(sys.argv) > 1:
    N, M, d = 4096, 1 << 19, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP().factorise(X, y, ls)
    Xd = gp._dev(Xs)
    for name, fn in [("i8", gp.score_i8), ("f64", gp.score)]:
        fn(Xd)
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(Xd)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(name, "%.1f ms" % (dt * 1e3), "%.3g cand/s" % (M / dt), r.best_idx, r.best_val, gp.last_screen if name == "i8" else "", flush=True)
