"""Randomised parity sweep of the HIP path against the CPU oracle (test infrastructure, like tests/):
random N, M, d, length scales, chunk sizes, acquisition kinds, shard offsets, the append route and the
host-pointer route.  usage: python tools/fuzz_parity.py [seconds] [seed]
Prints one line per failure and a summary; exit code 1 if anything disagreed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the first host-pointer call: a process that uses both routes imports PyTorch first)

from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd import host_binding as H
from oracle import gp_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_fail = 0
worst = dict(mu=0.0, sigma=0.0)


def first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


while time.time() < t_end:
    d = int(rng.integers(1, 17))
    N = int(rng.choice([1, 2, 3, rng.integers(4, 64), rng.integers(60, 70), rng.integers(120, 136), rng.integers(136, 700)]))
    M = int(rng.choice([1, rng.integers(2, 600), rng.integers(500, 1100), rng.integers(1100, 6000)]))
    chunk = int(rng.choice([512, 1024, 2048, 4096]))
    ls = np.exp(rng.uniform(np.log(0.08), np.log(3.0), d))
    scale = float(rng.choice([1.0, 1e-3, 1e4]))
    X = rng.uniform(0, 1, (N, d))
    if N > 3 and rng.random() < 0.2:
        X[1] = X[0]                                     # duplicated observation
    Xs = rng.uniform(-0.1, 1.1, (M, d))
    if rng.random() < 0.2:
        Xs[: min(M, N)] = X[: min(M, N)]                # candidates on top of observations (sigma ~ 0)
    y = scale * rng.standard_normal(N)
    kind = "lcb" if rng.random() < 0.6 else "ei"
    kw = dict(explore=float(rng.choice([4.0, 1.0, 0.0, 10.0]))) if kind == "lcb" else dict(
        f_best=float(y.min()), xi=float(rng.choice([0.0, 0.01])))
    route = rng.choice(["device", "append", "host"])
    off = 0 if route == "host" else int(rng.choice([0, 7, 1 << 33]))
    if Xs.shape == X.shape:
        off = 0  # the shape-coincidence rule speaks about the FULL candidate set: not a shard of a larger one
    tag = f"d={d} N={N} M={M} chunk={chunk} {kind} {kw} off={off} route={route} scale={scale}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", n_cases, tag, flush=True)
    try:
        mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
        acq_o = O.lcb(mu_o, sig_o, kw["explore"]) if kind == "lcb" else O.expected_improvement(mu_o, sig_o, kw["f_best"], kw["xi"])
        if route == "host":
            r = H.select_next(X, y, ls, Xs, acquisition=kind, chunk=chunk, **kw)
            mu, sig, acq, bi, nanc = r["mu"], r["sigma"], r["acq"], r["best_idx"], r["nan_count"]
        else:
            gp = DeviceGP(chunk=chunk)
            if route == "append" and N >= 2:
                n0 = int(rng.integers(1, N))
                gp.factorise(X[:n0], y[:n0], ls)
                for i in range(n0, N):
                    gp.append(X[i], y[i])
            else:
                # (farthest-point order of the observations, round 4: same posterior; not with the N == M quirk, which is
                #  keyed on the arrival index)
                gp.factorise(X, y, ls, order="fps" if (Xs.shape != X.shape and rng.random() < 0.4) else "arrival")
            # the reference's shape-coincidence jitter (point_selector.py:173), which the oracle applies by itself
            q = gp.score(Xs, acquisition=kind, dense=True, idx_offset=off, diag_add=1e-4 if Xs.shape == X.shape else 0.0, **kw)
            mu, sig, acq = q.mu.cpu().numpy(), q.sigma.cpu().numpy(), q.acq.cpu().numpy()
            bi, nanc = q.best_idx, q.nan_count
        ys = max(1.0, float(np.abs(y).max()))
        e_mu, e_sig = float(np.max(np.abs(mu - mu_o))) / ys, float(np.max(np.abs(sig - sig_o)))
        worst["mu"], worst["sigma"] = max(worst["mu"], e_mu), max(worst["sigma"], e_sig)
        # the stated fp64 tolerance of the path: random 1-D problems with hundreds of points reach
        # cond(K) ~ 1e7, beyond the order-tighter bound the fixed test problems meet
        tol_mu, tol_sig = 1e-9, 1e-8
        if d <= 2 and N > 100:  # hundreds of points on a line or in a plane: cond(K) ~ N / 1e-4, the two Cholesky
            tol_mu = 3e-9       # routes (oracle, device) then differ by up to ~cond eps each
        if route == "append":  # hundreds of columns built through the explicit inverse: cond(L) eps each
            tol_mu, tol_sig = 5e-9, 5e-8
        ok = nanc == 0 and e_mu <= tol_mu and e_sig <= tol_sig and bi == off + first_argmax(acq)
        top2 = np.sort(acq_o)[-2:] if M > 1 else np.array([-np.inf, acq_o[0]])
        if ok and top2[1] - top2[0] > 1e-7 * ys:
            ok = bi - off == first_argmax(acq_o)
        if not ok:
            n_fail += 1
            print(f"FAIL {tag}: dmu={e_mu:.3g} dsigma={e_sig:.3g} idx={bi - off} oracle={first_argmax(acq_o)} nan={nanc}", flush=True)
    except Exception as exc:  # noqa: BLE001
        n_fail += 1
        print(f"ERROR {tag}: {type(exc).__name__}: {exc}", flush=True)
    n_cases += 1
    if n_cases % 50 == 0:
        print(f"... {n_cases} cases, {n_fail} failures, worst dmu/|y| {worst['mu']:.3g}, worst dsigma {worst['sigma']:.3g}", flush=True)

print(f"fuzz: {n_cases} cases, {n_fail} failures, worst dmu/|y| {worst['mu']:.3g}, worst dsigma {worst['sigma']:.3g}")
sys.exit(1 if n_fail else 0)
