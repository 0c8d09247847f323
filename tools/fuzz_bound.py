"""Randomised sweep of the exact prefix bound (DeviceGP.score_bound) against the plain fp64 pass: random sizes, feature
counts, length scales, objectives (rough / nearly flat), observation orders (random / sorted along an axis / clustered;
factorised in arrival or in farthest-point order), ill-conditioned histories (points on a line, duplicated rows),
acquisitions (LCB weights 0 .. 30, EI), prefix lengths, duplicated and NaN candidates.  The bar: the same index, the same
NaN count, the value within 1e-9 relative (both values are fp64 kernels' - the fused launch of the plain pass and the
column-split launch that re-scores the survivors add |v|^2 up in different orders, and sigma^2 = c - |v|^2 cancels: with
sigma ~ 1e-2 and a weight of 30 the two differ by 4e-12 relative in one of 4,246 cases; 1e-12 holds for weights <= 4).
usage: python tools/fuzz_bound.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bayesian_optimisation_amd import DeviceGP

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_fail = n_pruned = n_fallback = 0
while time.time() < t_end:
    n_cases += 1
    d = int(rng.integers(1, 17))
    N = int(rng.integers(129, 3000))
    M = int(rng.integers(1000, 200000))
    ls = np.exp(rng.uniform(np.log(0.1), np.log(3.0), d))
    X = rng.uniform(0, 1, (N, d))
    kind = rng.integers(0, 3)
    if kind == 1:
        X = X[np.argsort(X[:, 0])]
    elif kind == 2:
        X[: N // 2] = 0.5 + 0.05 * rng.standard_normal((N // 2, d))
    elif rng.random() < 0.15:   # hundreds of points on a line + duplicated rows: cond(K) ~ 1e7 at the reference's jitter
        t = rng.uniform(0, 1, N)
        X = 0.5 + np.outer(t - 0.5, rng.standard_normal(d) * 0.3)
        X[rng.integers(0, N, N // 10)] = X[rng.integers(0, N, N // 10)]
    Xs = rng.uniform(-0.1, 1.1, (M, d))
    if rng.random() < 0.3:      # candidates ON observations (smallest variances)
        k = min(M, N) // 2
        Xs[:k] = X[rng.integers(0, N, k)]
    if rng.random() < 0.25:  # raw physical units: large coordinates, length scales to match
        sc = float(rng.choice([1e2, 1e3, 5e4]))
        X, Xs, ls = X * sc, Xs * sc, ls * sc * float(rng.choice([0.05, 0.3, 1.0]))
    y = rng.standard_normal(N) * rng.choice([1e-6, 0.1, 1.0, 30.0]) + rng.choice([0.0, 5.0])
    for i in rng.integers(0, M, int(rng.integers(0, 4))):
        Xs[i, rng.integers(0, d)] = np.nan
    if rng.random() < 0.3:
        Xs[rng.integers(0, M, 5)] = Xs[rng.integers(0, M)]
    if rng.random() < 0.5:
        kw = dict(acquisition="lcb", explore=float(rng.choice([0.0, 0.5, 4.0, 30.0])))
    else:
        kw = dict(acquisition="ei", f_best=float(y.min()), xi=float(rng.choice([0.0, 0.01])))
    tag = f"d={d} N={N} M={M} order={kind} {kw}"
    try:
        order = str(rng.choice(["fps", "fps", "arrival"]))   # the factorisation's order of the observations (both exact)
        gp = DeviceGP(chunk=int(rng.choice([4096, 1 << 15, 1 << 17]))).factorise(X, y, ls, order=order)
        J = int(rng.choice([0, 128, 256]))
        args = dict(prefix=J, prefix2=int(rng.choice([0, 2 * J, 4 * J]))) if J else {}
        if J and 2 * J > gp.Np:
            args = {}
        if args.get("prefix2", 0) > gp.Np:
            args["prefix2"] = 0
        r = gp.score_bound(Xs, idx_offset=5, **kw, **args)
        st = dict(gp.last_screen)
        r64 = gp.score(Xs, idx_offset=5, **kw)
        n_fallback += bool(st.get("fallback"))
        n_pruned += (not st.get("fallback")) and st.get("rescored", M) < M // 4
        if r.best_idx != r64.best_idx or r.nan_count != r64.nan_count or \
                abs(r.best_val - r64.best_val) > 1e-9 * max(1.0, abs(r64.best_val)):
            n_fail += 1
            print("FAIL", tag, (r.best_idx, r.best_val, r.nan_count), (r64.best_idx, r64.best_val, r64.nan_count), st, flush=True)
    except Exception as exc:  # noqa: BLE001
        n_fail += 1
        print("FAIL", tag, f"{type(exc).__name__}: {exc}", flush=True)
    if n_cases % 25 == 0:
        print(f"... {n_cases} cases, {n_fail} failures, {n_pruned} pruned to < M/4, {n_fallback} fell back", flush=True)
print(f"fuzz_bound: {n_cases} cases, {n_fail} failures, {n_pruned} pruned to < M/4, {n_fallback} fell back to the plain pass")
