"""A fixed-seed slice of the randomised sweeps in tools/ (fuzz_parity.py, fuzz_dropin.py run tens of thousands of cases
by hand): random sizes, feature counts, length scales, acquisition kinds and shard offsets, fp64 / fp32-screened /
int8-screened routes against the oracle.  Small enough for the driver-run suite."""
import numpy as np
import pytest

from bayesian_optimisation_amd import DeviceGP
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _case(rng):
    d = int(rng.integers(1, 17))
    N = int(rng.integers(1, 400))
    M = int(rng.integers(1, 5000))
    X = rng.uniform(0, 1, (N, d))
    Xs = rng.uniform(0, 1, (M, d))
    if rng.random() < 0.2 and N > 3:               # duplicated observation rows (the reference meets them: the grid is finite)
        X[rng.integers(0, N)] = X[rng.integers(0, N)]
    if rng.random() < 0.2:                          # candidates on top of observations
        k = min(N, M, 5)
        Xs[:k] = X[:k]
    ls = np.exp(rng.uniform(np.log(0.1), np.log(2.0), d))
    y = np.sin(X @ rng.standard_normal(d) * 3.0) * rng.choice([1.0, 50.0, 1e7]) + 0.01 * rng.standard_normal(N)
    return X, y, Xs, ls


@pytest.mark.parametrize("seed", range(24))
def test_random_case_all_routes_vs_oracle(seed):
    rng = np.random.default_rng(9000 + seed)
    X, y, Xs, ls = _case(rng)
    chunk = int(rng.choice([512, 1024, 4096]))
    off = int(rng.integers(0, 1 << 40))
    kind = "lcb" if rng.random() < 0.6 else "ei"
    kw = dict(acquisition="lcb", explore=float(rng.choice([1.0, 4.0, 7.5]))) if kind == "lcb" else \
        dict(acquisition="ei", f_best=float(y.min()), xi=float(rng.choice([0.0, 0.01])))
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    acq_o = O.lcb(mu_o, sig_o, kw["explore"]) if kind == "lcb" else O.expected_improvement(mu_o, sig_o, kw["f_best"], kw["xi"])
    ys = max(1.0, float(np.abs(y).max()))
    top2 = np.sort(acq_o)[-2:] if len(acq_o) > 1 else np.array([-np.inf, acq_o[0]])
    clear = top2[1] - top2[0] > 1e-7 * ys
    want = int(np.flatnonzero(acq_o == acq_o.max())[0])
    # (sigma tolerance; the coarse int8 screen is bounded on the variance instead: |dsigma^2| <= 1e-3)
    for route, sig_tol in (("score", 1e-8), ("score_f32", 5e-3), ("score_i8", 1e-8), ("score_i8c", None)):
        r = getattr(gp, route)(Xs, dense=True, idx_offset=off, **kw)
        assert r.nan_count == 0
        assert np.max(np.abs(r.mu.cpu().numpy() - mu_o)) <= 2e-9 * ys, route
        if sig_tol is None:
            assert np.max(np.abs(r.sigma.cpu().numpy() ** 2 - sig_o ** 2)) <= 1e-3, route
        else:
            assert np.max(np.abs(r.sigma.cpu().numpy() - sig_o)) <= sig_tol, route
        if clear:
            assert r.best_idx == off + want, route
    # the same case factorised in farthest-point order (what the exact-bound route runs on): same posterior, same point
    gf = DeviceGP(chunk=chunk).factorise(X, y, ls, order="fps")
    r = gf.score(Xs, dense=True, idx_offset=off, **kw)
    assert (gf.order == "fps") == (X.shape[0] > 128)
    assert np.max(np.abs(r.mu.cpu().numpy() - mu_o)) <= 2e-9 * ys and np.max(np.abs(r.sigma.cpu().numpy() - sig_o)) <= 1e-8
    rb = gf.score_bound(Xs, idx_offset=off, **kw)
    assert rb.best_idx == r.best_idx and rb.nan_count == 0
    if clear:
        assert r.best_idx == off + want
    Xa, ya = gf.observations_host()
    assert np.array_equal(Xa, X) and np.array_equal(ya, y)
