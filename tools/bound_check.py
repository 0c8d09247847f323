# prefix-bound screen (DeviceGP.score_bound): the selected point against the plain fp64 pass on assorted problems, then
# timing at the headline shape.  usage: python tools/bound_check.py [time]
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

for N, M, d, chunk in [(256, 2048, 8, 1024), (700, 5000, 8, 2048), (2048, 40000, 8, 4096), (1000, 60000, 16, 8192),
                       (2500, 70000, 8, 1 << 15), (4096, 1 << 17, 8, 1 << 16)]:
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls, order="fps")
    for kw in (dict(acquisition="lcb"), dict(acquisition="lcb", explore=1.0), dict(acquisition="lcb", explore=12.0),
               dict(acquisition="ei", f_best=float(y.min()))):
        r = gp.score_bound(Xs, idx_offset=3, **kw)
        scr = dict(gp.last_screen)
        r64 = gp.score(Xs, idx_offset=3, **kw)
        print(N, M, d, kw.get("acquisition"), kw.get("explore", ""), "idx", r.best_idx == r64.best_idx,
              "dval %.2g" % abs(r.best_val - r64.best_val), "nan", r.nan_count == r64.nan_count, scr, flush=True)

if len(sys.argv) > 1:
    N, M, d = 4096, 1 << 21, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP().factorise(X, y, ls, order="fps")
    Xd = gp._dev(Xs)
    for name, fn in [("bound", gp.score_bound), ("bound J=512 one level", lambda P: gp.score_bound(P, prefix=512, prefix2=0)),
                     ("bound J=256 one level", lambda P: gp.score_bound(P, prefix=256, prefix2=0)),
                     ("bound 128/512", lambda P: gp.score_bound(P, prefix=128, prefix2=512)),
                     ("bound 256/1536", lambda P: gp.score_bound(P, prefix=256, prefix2=1536)),
                     ("bound 384/1536", lambda P: gp.score_bound(P, prefix=384, prefix2=1536)),
                     ("bound 256/1024", lambda P: gp.score_bound(P, prefix=256, prefix2=1024)), ("bound again", gp.score_bound),
                     ("i8c", gp.score_i8c), ("f64", gp.score)]:
        fn(Xd)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 1 if name == "f64" else 3
        for _ in range(reps):
            r = fn(Xd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        print(name, "%.1f ms per 2^21" % (dt * 1e3), "%.4g cand/s" % (M / dt), r.best_idx, r.best_val,
              gp.last_screen if name != "f64" else "", flush=True)
