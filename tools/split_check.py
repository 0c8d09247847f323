# column-split fp64 launch against the plain one: a few candidates, N = 4096
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
for N, M, d in [(4096, 1 << 15, 8), (8192, 1 << 15, 16)]:
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=1 << 14).factorise(X, y, ls)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        r32 = gp.score_f32(Xs)
        torch.cuda.synchronize(); t32 = time.perf_counter() - t
    t = time.perf_counter()
    r64 = gp.score(Xs)
    torch.cuda.synchronize(); t64 = time.perf_counter() - t
    print(N, M, "f32-screened %.1f ms" % (t32 * 1e3), "plain fp64 %.1f ms" % (t64 * 1e3), gp.last_screen, r32.best_idx == r64.best_idx, r32.best_val - r64.best_val)
