"""Does replaying the factorisation's launches from a captured HIP graph shorten the gaps between its dependent kernels?
python tools/bench_factorise_graph.py [N ...]   (stream launches vs torch.cuda.CUDAGraph replay of the same call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem


def timed(fn, reps=10, groups=15):
    ts = []
    for _ in range(groups):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / reps)
    return np.median(ts) * 1e3


for N in [int(a) for a in sys.argv[1:]] or [512]:
    X, y, Xs, ls = make_problem(N, 512, 8)
    gp = DeviceGP()
    Xd, yd = gp._dev(X), gp._dev(y)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            gp.factorise(Xd, yd, ls, check=False)   # plan uploaded, workspaces allocated
        torch.cuda.synchronize()
        plain = timed(lambda: gp.factorise(Xd, yd, ls, check=False))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            gp.factorise(Xd, yd, ls, check=False)
        torch.cuda.synchronize()
        graph = timed(g.replay)
    print(f"N={N}: stream launches {plain:.4f} ms, graph replay {graph:.4f} ms", flush=True)
