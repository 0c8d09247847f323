"""Randomised sweep of the three variance screens (DeviceGP.score_f32 / score_i8 / score_i8c: a reduced-cost pass over every
candidate, then the fp64 kernels on whatever could still win) against the plain fp64 pass: random sizes, feature counts,
length scales, objectives, observation layouts (uniform / sorted / clustered / on a line with duplicated rows), raw physical
units, candidates ON observations, duplicated and NaN candidates, LCB weights 0 .. 30 and EI.  The screens' tolerance is
VERIFIED per call, not proven - this sweep looks for an input on which the verification passes and
the decision is wrong all the same.  The bar: the same index, the same NaN count, the value within 1e-9 relative (the
re-scoring launch adds |v|^2 up in another order than the fused launch of the plain pass).
usage: python tools/fuzz_screens.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bayesian_optimisation_amd import DeviceGP, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_fail = n_fallback = 0
by_mode = {"f32": 0, "i8": 0, "i8c": 0}
while time.time() < t_end:
    n_cases += 1
    d = int(rng.integers(1, 17))
    N = int(rng.integers(2, 3000)) if rng.random() < 0.8 else int(rng.integers(3000, 6000))
    M = int(rng.integers(300, 150000))
    ls = np.exp(rng.uniform(np.log(0.1), np.log(3.0), d))
    X = rng.uniform(0, 1, (N, d))
    kind = int(rng.integers(0, 4))
    if kind == 1:
        X = X[np.argsort(X[:, 0])]
    elif kind == 2:
        X[: N // 2] = 0.5 + 0.05 * rng.standard_normal((N // 2, d))
    elif kind == 3 and rng.random() < 0.5:   # points on a line + duplicated rows: cond(K) ~ 1e7 at the reference's jitter
        t = rng.uniform(0, 1, N)
        X = 0.5 + np.outer(t - 0.5, rng.standard_normal(d) * 0.3)
        if N >= 10:
            X[rng.integers(0, N, N // 10)] = X[rng.integers(0, N, N // 10)]
    Xs = rng.uniform(-0.1, 1.1, (M, d))
    if rng.random() < 0.3:      # candidates ON observations (smallest variances)
        k = min(M, N) // 2
        Xs[:k] = X[rng.integers(0, N, k)]
    if rng.random() < 0.25:     # raw physical units
        sc = float(rng.choice([1e2, 1e3, 5e4]))
        X, Xs, ls = X * sc, Xs * sc, ls * sc * float(rng.choice([0.05, 0.3, 1.0]))
    y = rng.standard_normal(N) * rng.choice([1e-6, 0.1, 1.0, 30.0]) + rng.choice([0.0, 5.0])
    if rng.random() < 0.2:      # a nearly flat acquisition: thousands of candidates within the screen's tolerance of the best
        y = np.full(N, float(rng.choice([0.0, 3.0]))) + 1e-9 * rng.standard_normal(N)
    for i in rng.integers(0, M, int(rng.integers(0, 4))):
        Xs[i, rng.integers(0, d)] = np.nan
    if rng.random() < 0.3:
        Xs[rng.integers(0, M, 5)] = Xs[rng.integers(0, M)]
    if rng.random() < 0.5:
        kw = dict(acquisition="lcb", explore=float(rng.choice([0.0, 0.5, 4.0, 30.0])))
    else:
        kw = dict(acquisition="ei", f_best=float(y.min()), xi=float(rng.choice([0.0, 0.01])))
    mode = str(rng.choice(["f32", "i8", "i8c"]))
    tag = f"{mode} d={d} N={N} M={M} layout={kind} {kw}"
    try:
        gp = DeviceGP(chunk=int(rng.choice([4096, 1 << 15, 1 << 17]))).factorise(X, y, ls)
        if mode != "f32" and gp.Np > _lib.I8_MAX_N:
            mode = "f32"
        by_mode[mode] += 1
        if mode == "f32":
            gp.prepare_f32()
            r = gp.score_f32(Xs, idx_offset=7, **kw)
        else:
            gp.prepare_i8()
            r = gp.score_i8(Xs, idx_offset=7, **kw) if mode == "i8" else gp.score_i8c(Xs, idx_offset=7, **kw)
        st = dict(gp.last_screen)
        r64 = gp.score(Xs, idx_offset=7, **kw)
        n_fallback += bool(st.get("fallback"))
        if r.best_idx != r64.best_idx or r.nan_count != r64.nan_count or \
                abs(r.best_val - r64.best_val) > 1e-9 * max(1.0, abs(r64.best_val)):
            n_fail += 1
            print("FAIL", tag, (r.best_idx, r.best_val, r.nan_count), (r64.best_idx, r64.best_val, r64.nan_count), st, flush=True)
    except Exception as exc:  # noqa: BLE001
        n_fail += 1
        print("FAIL", tag, f"{type(exc).__name__}: {exc}", flush=True)
    if n_cases % 25 == 0:
        print(f"... {n_cases} cases, {n_fail} failures, {n_fallback} fell back to the plain pass", flush=True)
print(f"fuzz_screens: {n_cases} cases ({by_mode}), {n_fail} failures, {n_fallback} fell back to the plain pass (seed {seed})")
sys.exit(1 if n_fail else 0)
