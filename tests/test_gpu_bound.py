"""Prefix-bound screen (DeviceGP.score_bound: gpbo_posterior_prefix_f64 + gpbo_bound_select_f64): branch and bound on the
variance reduction of the first J observations - everything in fp64, no tolerance anywhere.

|U^T k_c|^2 over the first J components is a lower bound of the whole sum, so the acquisition computed from it is an UPPER
bound of acq_func_eval_c (point_selector.py:98,204: both acquisitions increase with sigma); candidates whose bound is below
an exact value cannot be the maximum nor tie with it.  The bar here is the plain fp64 pass itself: same index (the LOWEST
among ties, point_selector.py:207), same NaN count, value equal up to the rounding of the column-split launch that
re-scores the survivors (1e-12 relative, as for the other screens), and the oracle's first arg-max."""
import ctypes as C

import numpy as np
import pytest

from bayesian_optimisation_amd import DeviceGP, _lib
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


def _same(r, r64, scale=1.0):
    assert r.best_idx == r64.best_idx and r.nan_count == r64.nan_count
    assert abs(r.best_val - r64.best_val) <= 1e-12 * max(scale, abs(r64.best_val))


@pytest.mark.parametrize("N,M,d,chunk", [(256, 2048, 8, 1024), (300, 7001, 3, 2048), (700, 5000, 8, 2048),
                                         (2048, 40000, 8, 4096), (1000, 60000, 16, 8192), (2500, 70000, 8, 1 << 15)])
def test_bound_screen_selects_the_fp64_point(N, M, d, chunk):
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    ys = max(1.0, float(np.abs(y).max()))
    for kw, acq_o in ((dict(acquisition="lcb", explore=4.0), O.lcb(mu_o, sig_o, 4)),
                      (dict(acquisition="lcb", explore=0.0), O.lcb(mu_o, sig_o, 0)),
                      (dict(acquisition="lcb", explore=15.0), O.lcb(mu_o, sig_o, 15)),
                      (dict(acquisition="ei", f_best=float(y.min()), xi=0.0),
                       O.expected_improvement(mu_o, sig_o, float(y.min()), 0.0))):
        r = gp.score_bound(Xs, idx_offset=11, **kw)
        st = dict(gp.last_screen)
        r64 = gp.score(Xs, idx_offset=11, **kw)
        _same(r, r64, ys)
        top2 = np.sort(acq_o)[-2:]
        if top2[1] - top2[0] > 1e-7 * ys:
            assert r.best_idx == 11 + _first_argmax(acq_o), kw
        assert st["mode"] == "bound" and st["prefix"] % 128 == 0 and 2 * st["prefix"] <= gp.Np
        if M >= 40000 and kw.get("explore", 4.0) <= 4.0:   # the bound prunes: a few thousand rows meet the fp64 kernels again
            assert not st["fallback"] and st["rescored"] < M // 4, st


def test_bound_is_an_upper_bound_of_the_fp64_acquisition_for_every_candidate():
    """The first pass alone (C ABI): mean bit-identical to the plain pass, sigma_ub >= sigma, acq_ub >= acq, for every prefix."""
    import torch

    X, y, Xs, ls = make_problem(1500, 9000, 8)
    Xs = Xs.copy()
    Xs[:60] = X[::25]          # candidates ON observations: variance ~ 0, possibly a hair below (the plain pass takes |var|)
    gp = DeviceGP(chunk=4096).factorise(X, y, ls)
    full = gp.score(Xs, dense=True)
    mu, sig, acq = (t.cpu().numpy() for t in (full.mu, full.sigma, full.acq))
    Xd = gp._dev(Xs)
    M = Xd.shape[0]
    chunk, wbytes = gp._ensure_post_workspace(M)
    lsp = gp.ls_h.ctypes.data_as(C.c_void_p)
    prev = None
    for J in (128, 512, 1024, gp.Np):
        o = [torch.empty(M, dtype=torch.float64, device=gp.device) for _ in range(3)]
        st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, lsp, gp._ptr(gp.U),
                                              gp._ptr(gp.alpha), 1.000101, _lib.ACQ_LCB, 4.0, 0.0, 0, chunk, J, gp._ptr(o[0]),
                                              gp._ptr(o[1]), gp._ptr(o[2]), gp._ptr(gp._result), gp._ptr(gp._work_post),
                                              wbytes, None, gp._stream())
        assert st == 0
        m, s, a = (t.cpu().numpy() for t in o)
        # the first pass takes the pair distances from the matrix cores (kstar_mfma.hip) and reports the mean from BELOW by
        # its error bound: never above the plain pass's mean, never further below than that bound - round 5: the entries
        # come from a one-term exponential (relative error <= 6e-8), so the bound is ~6e-8 sum|alpha| (7e-4 here)
        S0 = float(gp.alpha[: gp.N].abs().sum())
        assert np.all(m <= mu) and np.max(mu - m) <= 1.3e-7 * S0 + 1e-7
        assert np.all(s >= sig) and np.all(a >= acq)   # (the bound's variance is clamped at 0 and padded by 1e-8)
        if prev is not None:
            assert np.all(s <= prev + 1e-13)   # more observations, tighter bound
        prev = s
    assert np.max(np.abs(prev ** 2 - sig ** 2)) <= 2e-8   # the whole prefix is the plain pass, up to the pad
    # argument checks: prefix not a multiple of 128, LCB with a negative weight (the bound would point the wrong way)
    bad = [dict(J=100, p0=4.0), dict(J=256, p0=-1.0), dict(J=gp.Np + 128, p0=4.0)]
    for b in bad:
        st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, lsp, gp._ptr(gp.U),
                                              gp._ptr(gp.alpha), 1.000101, _lib.ACQ_LCB, b["p0"], 0.0, 0, chunk, b["J"], None,
                                              None, gp._ptr(o[2]), gp._ptr(gp._result), gp._ptr(gp._work_post), wbytes, None,
                                              gp._stream())
        assert st == -1   # GPBO_ERR_ARG


@pytest.mark.parametrize("d", [1, 2, 6, 7, 14, 15, 16])
def test_first_pass_kernel_for_every_operand_length(d):
    """The first pass builds K(X*,X) from ONE inner product of length d + 2 per pair on the fp64 matrix cores
    (csrc/kstar_mfma.hip: 1 .. 5 MFMAs of depth 4).  With the whole prefix the pass is the plain pass up to the expanded
    form's error: the mean from below by at most its bound, sigma above by at most the variance pad."""
    import torch

    X, y, Xs, ls = make_problem(700, 5000, d)
    Xs = Xs.copy()
    Xs[17, 0] = np.nan
    gp = DeviceGP(chunk=2048).factorise(X, y, ls)
    full = gp.score(Xs, dense=True)
    mu, sig = full.mu.cpu().numpy(), full.sigma.cpu().numpy()
    Xd = gp._dev(Xs)
    M = Xd.shape[0]
    chunk, wbytes = gp._ensure_post_workspace(M)
    o = [torch.empty(M, dtype=torch.float64, device=gp.device) for _ in range(3)]
    st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                          gp._ptr(gp.U), gp._ptr(gp.alpha), 1.000101, _lib.ACQ_LCB, 4.0, 0.0, 0, chunk, gp.Np,
                                          gp._ptr(o[0]), gp._ptr(o[1]), gp._ptr(o[2]), gp._ptr(gp._result), gp._ptr(gp._work_post),
                                          wbytes, None, gp._stream())
    assert st == 0
    m, s = o[0].cpu().numpy(), o[1].cpu().numpy()
    ok = np.ones(M, bool)
    ok[17] = False
    assert np.isnan(m[17]) and np.isnan(mu[17])
    S0 = float(gp.alpha[: gp.N].abs().sum())
    assert np.all(m[ok] <= mu[ok]) and np.max(mu[ok] - m[ok]) <= 1.3e-7 * S0 + 1e-7 * max(1.0, np.abs(y).max())
    assert np.all(s[ok] >= sig[ok]) and np.max(s[ok] ** 2 - sig[ok] ** 2) <= 2e-8


def test_ties_nan_and_chunk_invariance():
    X, y, Xs, ls = make_problem(600, 20000, 6)
    gp = DeviceGP(chunk=4096).factorise(X, y, ls)
    best = gp.score(Xs).best_idx
    # copies of the winning row: the LOWEST index among exact ties wins (point_selector.py:207)
    Xt = Xs.copy()
    for i in (19990, 7, 12345):
        Xt[i] = Xs[best]
    r, r64 = gp.score_bound(Xt), gp.score(Xt)
    assert r64.best_idx == min(7, best) and r.best_idx == r64.best_idx and r.best_val == pytest.approx(r64.best_val, rel=1e-13)
    # NaN candidates are counted and never selected
    Xn = Xs.copy()
    Xn[3, 0] = np.nan
    Xn[15000, 2] = np.nan
    r, r64 = gp.score_bound(Xn), gp.score(Xn)
    assert r.nan_count == r64.nan_count == 2 and r.best_idx == r64.best_idx
    # infinite candidate coordinates: the plain pass gives k = 0 (mean 0, prior sigma) - a legitimate, possibly winning, value
    Xi = Xs.copy()
    Xi[11, 1] = np.inf
    Xi[4000, 0] = -np.inf
    Xi[9000, 3] = 1e200
    r, r64 = gp.score_bound(Xi), gp.score(Xi)
    assert r.best_idx == r64.best_idx and r.nan_count == r64.nan_count
    r, r64 = gp.score_bound(Xi, explore=500.0), gp.score(Xi, explore=500.0)   # now the far-away candidates win (sigma = prior)
    assert r.best_idx == r64.best_idx and r.nan_count == r64.nan_count
    # chunking and the prefix length change the work, never the answer
    ref = gp.score(Xs)
    for chunk, prefix in ((1024, None), (8192, 128), (4096, 256)):
        g = DeviceGP(chunk=chunk).factorise(X, y, ls)
        _same(g.score_bound(Xs, prefix=prefix), ref)


def test_weak_bounds_end_in_the_plain_pass_or_many_survivors_never_in_a_wrong_point():
    """(a) a flat objective: the mean separates nothing; (b) observations sorted along one axis: the prefix knows one corner of
    the domain only; (c) 2,500 exact ties; (d) dense outputs, a negative weight, too few column blocks: not this route."""
    X, y, Xs, ls = make_problem(900, 30000, 5)
    gp = DeviceGP(chunk=8192)
    gp.factorise(X, np.full_like(y, 0.3) + 1e-9 * y, ls)
    _same(gp.score_bound(Xs), gp.score(Xs))
    order = np.argsort(X[:, 0])
    gp.factorise(X[order], y[order], ls)
    _same(gp.score_bound(Xs), gp.score(Xs))
    _same(gp.score_bound(Xs, acquisition="ei", f_best=float(y.min())), gp.score(Xs, acquisition="ei", f_best=float(y.min())))
    gp.factorise(X, y, ls)
    Xt = np.repeat(Xs[:1], 2500, axis=0)
    r = gp.score_bound(Xt)
    assert r.best_idx == 0
    r = gp.score_bound(Xs, dense=True)
    assert gp.last_screen["fallback"] and r.mu is not None and r.best_idx == gp.score(Xs).best_idx
    r = gp.score_bound(Xs, explore=-2.0)
    assert gp.last_screen["fallback"] and r.best_idx == gp.score(Xs, explore=-2.0).best_idx
    Xa, ya, Xsa, lsa = make_problem(100, 3000, 4)
    g2 = DeviceGP(chunk=1024).factorise(Xa, ya, lsa)
    r = g2.score_bound(Xsa)
    assert g2.last_screen["fallback"] and r.best_idx == g2.score(Xsa).best_idx


@pytest.mark.parametrize("explore", [10.0, 25.0, 1000.0])
def test_large_exploration_weights_go_through_the_second_level_not_back_to_the_plain_pass(explore):
    """The weaker the mean's grip, the more candidates survive the first-level bound: all of them may go on to the second
    level (1/16 of a plain pass per candidate), and the plain pass takes over only when more than M/8 reach the fp64 kernels.
    Whatever the route, the point is the plain pass's."""
    N, M = 2048, 1 << 17
    X, y, Xs, ls = make_problem(N, M, 8)
    gp = DeviceGP().factorise(X, y, ls)
    rb = gp.score_bound(Xs, explore=explore)
    scr = dict(gp.last_screen)
    _same(rb, gp.score(Xs, explore=explore))
    assert scr["mode"] == "bound" and scr["prefix2"] > 0
    if not scr["fallback"]:
        assert scr["rescored"] <= max(M // 8, 4 * 4096)
    if explore == 10.0:   # more first-level survivors than the old limit (M / 16) allowed, and still no plain pass
        assert scr["survivors"] > M // 16 and not scr["fallback"], scr
    # with the old limit forced, the same call ends in the plain pass - and in the same point
    gp.screen_cap = M // 16
    rc = gp.score_bound(Xs, explore=explore)
    _same(rc, rb)
    gp.screen_cap = None


@pytest.mark.parametrize("name", ["g5_d8_n512_m4096", "g5_d8_n2048_m4096", "g6_d16_n256_m2048"])
def test_bound_screen_selects_the_reference_point(golden, name):
    """Vectors produced by the reference itself (tests/golden/make_golden.py)."""
    g = golden(name)
    X, y, Xs, ls = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
    gp = DeviceGP(chunk=2048).factorise(X, y, g["kernel_params"])
    r = gp.score_bound(Xs)
    if g["top2_gap"] > 1e-7 * max(1.0, np.abs(y).max()):
        assert r.best_idx == _first_argmax(g["acq_func_eval"])
    assert abs(r.best_val - g["acq_func_eval"].max()) <= 1e-8 * max(1.0, np.abs(y).max())


def test_drop_in_class_without_dense_outputs_returns_the_same_multi_index():
    """PointSelector(dense_outputs=False): a caller that needs the next point only - the acquisition calls return the multi-index
    the full class returns, mean_func / cov_func / acq_func_eval stay None."""
    from bayesian_optimisation_amd import PointSelector

    rng = np.random.default_rng(3)
    N, d, g = 900, 3, 48
    X = rng.uniform(0, 1, (N, d))
    y = np.sin(6 * X[:, 0]) * np.cos(4 * X[:, 1]) + X[:, 2] ** 2 + 0.01 * rng.standard_normal(N)
    axes = [np.linspace(0, 1, g)] * d
    Xs = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, d)
    out = []
    for dense in (True, False):
        ps = PointSelector(dense_outputs=dense)
        ps.name, ps.iteration = "T", 0
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [g] * d, Xs
        ps.set_kernel_params(np.array([0.15, 0.2, 0.4]))
        ps.update_surrogate()
        out.append((ps.lower_confidence_bound(), ps.lower_confidence_bound(explore=1.0), ps.expected_improvement(), ps))
    full, only = out
    for a, b in zip(full[:3], only[:3]):
        assert np.array_equal(a, b) and a.dtype == np.int64 and a.shape == (d,)
    assert only[3].mean_func is None and only[3].cov_func is None and only[3].acq_func_eval is None
    assert full[3].mean_func.shape == (g, g, g)
    assert only[3]._gp.last_screen["mode"] == "bound"


@pytest.mark.parametrize("scale,ls_lo,ls_hi,yscale", [(1e3, 8.0, 60.0, 1e-3), (5e4, 300.0, 4e3, 1e-6), (1.0, 0.02, 0.2, 1e-4)])
def test_unnormalised_inputs_and_small_objectives(scale, ls_lo, ls_hi, yscale):
    """raw physical units (coordinates of 1e3 .. 5e4 with length scales of 10 .. 1e3, far from the centroid
    in length-scale units) and a small |y| - where an expanded-distance K(X*,X) entry error, amplified by |U|, could
    overstate |v[:J]|^2 by more than the fixed pad.  The stored prefix rows now come from the difference-form kernel of
    the plain pass: the bound must hold for every candidate and the selected point must be the plain pass's."""
    rng = np.random.default_rng(int(scale) + 7)
    d, N, M = 6, 1500, 60000
    X = rng.uniform(0, scale, (N, d))
    Xs = rng.uniform(-0.05 * scale, 1.05 * scale, (M, d))
    Xs[:200] = X[:200] + 1e-7 * scale * rng.standard_normal((200, d))  # candidates on top of observations: tiny variance
    ls = np.exp(rng.uniform(np.log(ls_lo), np.log(ls_hi), d))
    y = yscale * rng.standard_normal(N)
    gp = DeviceGP(chunk=1 << 15).factorise(X, y, ls)
    full = gp.score(Xs, dense=True)
    acq64 = full.acq.cpu().numpy()
    for kw in (dict(acquisition="lcb", explore=4.0), dict(acquisition="lcb", explore=0.25),
               dict(acquisition="ei", f_best=float(y.min()), xi=0.0)):
        r64 = gp.score(Xs, **kw)
        rb = gp.score_bound(Xs, **kw)
        assert rb.best_idx == r64.best_idx and rb.nan_count == r64.nan_count == 0
        assert abs(rb.best_val - r64.best_val) <= 1e-9 * max(1.0, abs(r64.best_val))
    # the first pass itself: an upper bound of the fp64 acquisition for EVERY candidate
    import ctypes as C

    from bayesian_optimisation_amd import _lib

    t = gp.torch
    Xsd = gp._dev(Xs)
    chunk, wbytes = gp._ensure_post_workspace(M)
    ub = t.empty(M, dtype=t.float64, device=gp.device)
    J = 128
    st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xsd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                          gp._ptr(gp.U), gp._ptr(gp.alpha), 1.000101, _lib.ACQ_LCB, 4.0, 0.0, 0, chunk, J, None,
                                          None, gp._ptr(ub), gp._ptr(gp._result), gp._ptr(gp._work_post), wbytes, None,
                                          gp._stream())
    assert st == 0
    t.cuda.synchronize()
    assert (ub.cpu().numpy() >= acq64).all()


def _fps_reference(X, ls, J):
    """Farthest-point sampling in length-scale units, NumPy: first = farthest from the centroid, ties to the lowest index."""
    Z = X / ls
    c = Z.mean(0)
    d0 = ((Z - c) ** 2).sum(1)
    order = [int(np.flatnonzero(d0 == d0.max())[0])]
    mind = np.full(len(X), np.inf)
    for _ in range(J - 1):
        p = order[-1]
        mind = np.minimum(mind, ((Z - Z[p]) ** 2).sum(1))
        mind[p] = -np.inf
        order.append(int(np.flatnonzero(mind == mind.max())[0]))
    return np.array(order)


@pytest.mark.parametrize("N,d", [(1500, 5), (9300, 3), (700, 16), (3000, 16), (5000, 8)])
def test_farthest_point_order_and_the_factorisation_of_the_permuted_problem(N, d):
    """gpbo_fps_order_f64 against NumPy: the J members, every other observation after them in index order, the gathered rows
    (register-resident kernel; one launch per member where a thread's points would not fit its registers: N = 9300,
    3000 x 16, 5000 x 8) - and factorise(order="fps") = the factorisation of exactly that permuted problem."""
    X, y, Xs, ls = make_problem(N, 512, d)
    gp = DeviceGP().factorise(X, y, ls, order="fps")
    J = gp.bound_prefix()
    assert gp.order == "fps" and J == 128 * max(1, gp.Np // 16 // 128)
    perm = gp.perm.cpu().numpy()
    ref = _fps_reference(X, ls, J)
    # distances are sums of d squares in a different association on the GPU (fma) - members agree unless two candidates
    # tie to the last bit; on this seeded problem they do not
    assert np.array_equal(perm[:J], ref)
    rest = np.setdiff1d(np.arange(N), ref)
    assert np.array_equal(perm[J:], rest) and np.array_equal(np.sort(perm), np.arange(N))
    for _ in range(3):   # the same order every time (the one-launch-per-member form hands partial results between workgroups)
        assert np.array_equal(DeviceGP().factorise(X, y, ls, order="fps").perm.cpu().numpy(), perm)
    assert np.array_equal(gp.X[:N].cpu().numpy(), X[perm]) and np.array_equal(gp.y[:N].cpu().numpy(), y[perm])
    Xa, ya = gp.observations_host()
    assert np.array_equal(Xa, X) and np.array_equal(ya, y)
    # the factors are those of the permuted problem ...
    K = O.kernel_rbf(X[perm], X[perm], ls) + 1e-6 * np.eye(N)   # kernel_rbf adds its own 1e-4 (same shapes)
    L = np.linalg.cholesky(K)
    U_ref = np.linalg.inv(L).T
    assert np.max(np.abs(gp.U[:N, :N].cpu().numpy() - U_ref)) <= 1e-9 * np.abs(U_ref).max()
    assert np.max(np.abs(gp.cov_meas_host() - (O.kernel_rbf(X, X, ls) + 1e-6 * np.eye(N)))) <= 1e-14   # caller's order
    # ... and the posterior is the arrival-order one within rounding (a GP does not know the order of its observations)
    ga = DeviceGP().factorise(X, y, ls)
    rf, ra = gp.score(Xs, dense=True), ga.score(Xs, dense=True)
    ys = max(1.0, float(np.abs(y).max()))
    assert np.max(np.abs(rf.mu.cpu().numpy() - ra.mu.cpu().numpy())) <= 1e-10 * ys
    assert np.max(np.abs(rf.sigma.cpu().numpy() - ra.sigma.cpu().numpy())) <= 1e-9
    assert rf.best_idx == ra.best_idx
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(rf.mu.cpu().numpy() - mu_o)) <= 1e-10 * ys and np.max(np.abs(rf.sigma.cpu().numpy() - sig_o)) <= 1e-9
    # argument checks of the C entry point
    import torch

    lib, t = gp.lib, torch
    Xd = gp._dev(X)
    pr = t.empty(N, dtype=t.int64, device=gp.device)
    wb = int(lib.gpbo_fps_order_workspace_bytes(N))
    w = t.empty(wb // 8 + 1, dtype=t.float64, device=gp.device)
    lsp = gp.ls_h.ctypes.data_as(C.c_void_p)
    assert lib.gpbo_fps_order_f64(gp._ptr(Xd), None, N, d, lsp, 0, gp._ptr(pr), None, None, gp._ptr(w), wb, gp._stream()) == -1
    assert lib.gpbo_fps_order_f64(gp._ptr(Xd), None, N, d, lsp, N + 1, gp._ptr(pr), None, None, gp._ptr(w), wb, gp._stream()) == -1
    assert lib.gpbo_fps_order_f64(gp._ptr(Xd), None, N, d, lsp, J, gp._ptr(pr), None, gp._ptr(w), gp._ptr(w), wb, gp._stream()) == -1  # yp without y
    assert lib.gpbo_fps_order_f64(gp._ptr(Xd), None, N, d, lsp, J, gp._ptr(pr), None, None, gp._ptr(w), 8, gp._stream()) == -3
    assert lib.gpbo_fps_order_f64(gp._ptr(Xd), None, N, d, lsp, J, gp._ptr(pr), None, None, gp._ptr(w), wb, gp._stream()) == 0
    t.cuda.synchronize()
    assert np.array_equal(pr.cpu().numpy(), perm)


@pytest.mark.parametrize("N,d,J", [(12000, 5, 96), (20000, 16, 64), (40000, 3, 48), (40000, 9, 48), (70000, 2, 24),
                                   (1024, 1, 1024), (1025, 2, 7)])
def test_every_form_of_the_selection_gives_the_numpy_sequence(N, d, J):
    """gpbo_fps_order_f64 alone (no factorisation at these sizes): co-operating workgroups of 512 threads with up to eight
    points per thread (12,000 x 5: 8 workgroups; 20,000 x 16 and 40,000 x 3: 10) exchanging one record per member; ONE
    LAUNCH PER MEMBER (fps_step_kernel, with its release / acquire ticket and its NaN rule) where 16 workgroups cannot hold
    the points - beyond 32,768 observations at d > 8 (40,000 x 9: four points per thread, 20 workgroups) and beyond 65,536 at
    d <= 8 (70,000 x 2); every observation a member (J = N); a ragged second point per thread (1,025) - the same sequence as
    NumPy, the same every time, a permutation of 0 .. N-1, rows gathered in that order."""
    import torch

    rng = np.random.default_rng(N + d)
    X = rng.uniform(0, 1, (N, d))
    X[N // 3] = X[N // 7]                       # a duplicated row: distance 0 to a member, chosen last
    y = rng.standard_normal(N)
    ls = np.exp(rng.uniform(np.log(0.2), np.log(2.0), d))
    gp = DeviceGP()
    lib, t = gp.lib, torch
    Xd, yd = gp._dev(X), gp._dev(y)
    wb = int(lib.gpbo_fps_order_workspace_bytes(N))
    w = t.empty(wb // 8 + 1, dtype=t.float64, device=gp.device)
    lsp = np.ascontiguousarray(ls).ctypes.data_as(C.c_void_p)
    ref = _fps_reference(X, ls, J)
    first = None
    for _ in range(3):
        pr = t.full((N,), -1, dtype=t.int64, device=gp.device)
        Xp, yp = t.empty_like(Xd), t.empty_like(yd)
        assert lib.gpbo_fps_order_f64(gp._ptr(Xd), gp._ptr(yd), N, d, lsp, J, gp._ptr(pr), gp._ptr(Xp), gp._ptr(yp), gp._ptr(w), wb,
                                      gp._stream()) == 0
        t.cuda.synchronize()
        perm = pr.cpu().numpy()
        assert np.array_equal(perm[:J], ref)
        assert np.array_equal(perm[J:], np.setdiff1d(np.arange(N), ref))
        assert np.array_equal(Xp.cpu().numpy(), X[perm]) and np.array_equal(yp.cpu().numpy(), y[perm])
        first = perm if first is None else first
        assert np.array_equal(perm, first)


@pytest.mark.parametrize("N,d", [(900, 3), (3000, 8), (9000, 4)])
def test_nan_observations_never_break_the_order(N, d):
    """A NaN coordinate makes the centroid - and with it every first-sweep distance - NaN; rows of NaN never have a real
    distance.  The order must still be a permutation (such rows count as -1: picked last) and the gathered copies the
    caller's rows; the factorisation of such data then fails the way the arrival-order one does (LinAlgError)."""
    import torch

    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d))
    y = rng.standard_normal(N)
    ls = np.full(d, 0.4)
    gp = DeviceGP()
    lib, t = gp.lib, torch
    lsp = np.ascontiguousarray(ls).ctypes.data_as(C.c_void_p)
    wb = int(lib.gpbo_fps_order_workspace_bytes(N))
    w = t.empty(wb // 8 + 1, dtype=t.float64, device=gp.device)
    for bad_rows in ([5], list(range(0, N, 2)), list(range(N))):
        Xn = X.copy()
        Xn[bad_rows, 0] = np.nan
        Xd, yd = gp._dev(Xn), gp._dev(y)
        pr = t.full((N,), -1, dtype=t.int64, device=gp.device)
        Xp, yp = t.empty_like(Xd), t.empty_like(yd)
        J = 128
        assert lib.gpbo_fps_order_f64(gp._ptr(Xd), gp._ptr(yd), N, d, lsp, J, gp._ptr(pr), gp._ptr(Xp), gp._ptr(yp), gp._ptr(w), wb,
                                      gp._stream()) == 0
        t.cuda.synchronize()
        perm = pr.cpu().numpy()
        assert np.array_equal(np.sort(perm), np.arange(N))
        assert np.array_equal(yp.cpu().numpy(), y[perm]) and np.array_equal(Xp.cpu().numpy(), Xn[perm], equal_nan=True)
        with pytest.raises(np.linalg.LinAlgError):
            DeviceGP().factorise(Xn, y, ls, order="fps")
        with pytest.raises(np.linalg.LinAlgError):
            DeviceGP().factorise(Xn, y, ls)


def test_failed_pivot_is_reported_as_the_callers_row_in_both_orders():
    """An observation with a NaN coordinate makes its own pivot the first one that fails, wherever the factorisation's order
    puts it: the LinAlgError names that observation as a row of the CALLER's arrays in both orders (as the host-pointer entry
    point does through `info`) - round 4 reported the position in factorisation order for order="fps"."""
    import re

    N, d = 3000, 4
    X, y, _, ls = make_problem(N, 8, d)
    bad = 2100
    X = X.copy()
    X[bad, 1] = np.nan
    for order in ("arrival", "fps"):
        with pytest.raises(np.linalg.LinAlgError) as e:
            DeviceGP().factorise(X, y, ls, order=order)
        row = int(re.search(r"pivot (\d+) of", str(e.value)).group(1))
        assert row - 1 == bad, (order, row)


@pytest.mark.parametrize("order", ["sobol", "sorted", "reversed", "clustered_first"])
def test_pruning_does_not_depend_on_the_order_of_the_observations(order):
    """the literal prefix of a history sorted along an axis (or whose first rows sit in one
    cluster) knows one corner of the domain and prunes nothing; factorised in farthest-point order it prunes like a Sobol
    history.  The selected point is the plain pass's in every order, and the same point in all of them."""
    N, M, d = 2048, 1 << 17, 8
    X, y, Xs, ls = make_problem(N, M, d)
    if order == "sorted":
        o = np.argsort(X[:, 0])
    elif order == "reversed":
        o = np.arange(N)[::-1].copy()
    elif order == "clustered_first":
        c = ((X - 0.25) ** 2).sum(1)
        o = np.argsort(c)          # the first rows are the observations nearest one point of the domain
    else:
        o = np.arange(N)
    gp = DeviceGP(chunk=1 << 15).factorise(X[o], y[o], ls, order="fps")
    ga = DeviceGP(chunk=1 << 15).factorise(X[o], y[o], ls)
    for kw in (dict(acquisition="lcb", explore=4.0), dict(acquisition="ei", f_best=float(y.min()), xi=0.0)):
        r64 = gp.score(Xs, **kw)
        rb = gp.score_bound(Xs, **kw)
        st = dict(gp.last_screen)
        _same(rb, r64, max(1.0, float(np.abs(y).max())))
        assert st["order"] == "fps" and not st["fallback"] and st["rescored"] < M // 4, (order, st)
        assert ga.score(Xs, **kw).best_idx == rb.best_idx   # and the arrival-order factorisation's plain pass agrees
    if order in ("sorted", "clustered_first"):
        ra = ga.score_bound(Xs)
        arrival = dict(ga.last_screen)
        assert arrival["order"] == "arrival" and ra.best_idx == gp.score_bound(Xs).best_idx
        assert gp.last_screen["rescored"] <= arrival.get("rescored", M) or arrival["fallback"]


# ---- the bound of the route that SHIPS, per candidate -------------------------------------------------
# Since round 4 the route has ONE factorisation (the farthest-point-ordered observations): bounds and exact values come from
# the same U through the same kernels, |v[:J]|^2 is a partial sum of the squares the plain pass adds up.  These tests hold
# acq_ub >= acq for EVERY candidate on that form - both levels, unnormalised inputs, candidates on observations, an
# ill-conditioned history.

def _prefix_pass(gp, Xs, J, kind=None, p0=4.0, p1=0.0):
    import torch

    Xd = gp._dev(Xs)
    M = Xd.shape[0]
    chunk, wbytes = gp._ensure_post_workspace(M)
    o = [torch.empty(M, dtype=torch.float64, device=gp.device) for _ in range(3)]
    st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                          gp._ptr(gp.U), gp._ptr(gp.alpha), 1.000101, _lib.ACQ_LCB if kind is None else kind,
                                          p0, p1, 0, chunk, J, gp._ptr(o[0]), gp._ptr(o[1]), gp._ptr(o[2]), gp._ptr(gp._result),
                                          gp._ptr(gp._work_post), wbytes, None, gp._stream())
    assert st == 0
    torch.cuda.synchronize()
    return tuple(t.cpu().numpy() for t in o)


def _ill_conditioned(N, M, d, seed):
    """Hundreds of points on a line, a tenth of them duplicated rows, the rest scattered: cond(K) ~ 1e7 at the reference's
    jitter - the case the fuzz sweeps report as worst for the mean."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0, 1, N)
    X = 0.5 + np.outer(t - 0.5, np.linspace(0.2, 0.5, d))
    X[N // 2:] = rng.uniform(0, 1, (N - N // 2, d))
    dup = rng.integers(0, N // 2, N // 10)
    X[rng.integers(0, N, N // 10)] = X[dup]
    X = X[rng.permutation(N)]
    ls = np.full(d, 0.6)
    y = np.sin(5 * X[:, 0]) + 0.3 * X.sum(1) + 1e-3 * rng.standard_normal(N)
    Xs = rng.uniform(0, 1, (M, d))
    Xs[:400] = X[rng.integers(0, N, 400)]                                    # candidates ON observations
    Xs[400:800] = 0.5 + np.outer(rng.uniform(-0.5, 0.5, 400), np.linspace(0.2, 0.5, d))   # ... and on the line between them
    return X, y, Xs, ls


@pytest.mark.parametrize("problem", ["sobol", "raw_1e3", "raw_5e4", "tiny_ls", "ill_conditioned", "sorted_history", "tight_cluster"])
def test_shipping_bound_holds_for_every_candidate_at_both_levels(problem):
    """factorise(order="fps") + gpbo_posterior_prefix_f64 at J = 128 / 256 / 1024 (the first level of N = 2048 / 4096 and
    the second level of N = 4096) and J = N/4 (this problem's second level): sigma_ub >= sigma and acq_ub >= acq of the plain
    pass ON THE SAME FACTORISATION for every candidate, LCB and EI; the bounds tighten as J grows; and the plain pass's own
    variance stays far from the |var| reflection (>= half the jitter) even at cond(K) ~ 1e7."""
    N, M, d = 1500, 30000, 6
    rng = np.random.default_rng(11)
    if problem == "sobol":
        X, y, Xs, ls = make_problem(N, M, d)
        Xs = Xs.copy()
        Xs[:60] = X[::25]
    elif problem in ("raw_1e3", "raw_5e4", "tiny_ls"):
        scale, ls_lo, ls_hi, yscale = {"raw_1e3": (1e3, 8.0, 60.0, 1e-3), "raw_5e4": (5e4, 300.0, 4e3, 1e-6),
                                       "tiny_ls": (1.0, 0.02, 0.2, 1e-4)}[problem]
        X = rng.uniform(0, scale, (N, d))
        Xs = rng.uniform(-0.05 * scale, 1.05 * scale, (M, d))
        Xs[:200] = X[:200] + 1e-7 * scale * rng.standard_normal((200, d))
        ls = np.exp(rng.uniform(np.log(ls_lo), np.log(ls_hi), d))
        y = yscale * rng.standard_normal(N)
    elif problem == "ill_conditioned":
        X, y, Xs, ls = _ill_conditioned(N, M, d, 5)
        K = O.kernel_rbf(X, X, ls) + 1e-6 * np.eye(N)
        assert np.linalg.cond(K) > 3e6
    elif problem == "tight_cluster":
        # every observation and candidate within a hundredth of a length scale of one point: k ~ 1 everywhere, the mean is a
        # sum of N terms of size |alpha_i| that cancel to O(1), and the expanded distance's slack term (|a|^2 S0 + S1)
        # vanishes - what keeps the first pass's mean BELOW the plain pass's is the summation term of its slack (round 4)
        X = 0.5 + 0.01 * rng.uniform(-1, 1, (N, d))
        Xs = 0.5 + 0.01 * rng.uniform(-1, 1, (M, d))
        Xs[:300] = X[:300]
        ls = np.full(d, 1.0)
        y = np.sin(40 * X[:, 0]) + X[:, 1:].sum(1)
    else:
        X, y, Xs, ls = make_problem(N, M, d)
        o = np.argsort(X[:, 0])
        X, y = X[o], y[o]
    gp = DeviceGP(chunk=8192).factorise(X, y, ls, order="fps")
    full = gp.score(Xs, dense=True)
    mu, sig, acq = (t.cpu().numpy() for t in (full.mu, full.sigma, full.acq))
    var = gp.score(Xs[:800], dense=True).sigma.cpu().numpy() ** 2
    assert var.min() >= 0.5 * 1.01e-4   # true variance >= jitter: the plain pass never gets near a negative value
    fb = float(y.min())
    ei = gp.score(Xs, acquisition="ei", f_best=fb, dense=True).acq.cpu().numpy()
    prev = None
    for J in (128, 256, 384, 1024):      # 384 = N/4 rounded to the granule: this problem's own second level
        m, s, a = _prefix_pass(gp, Xs, J)
        assert np.all(m <= mu)           # the mean is reported from below (kstar_mfma.hip), never above the plain pass's
        assert np.all(s >= sig) and np.all(a >= acq), (problem, J, float((sig - s).max()))
        _, _, ae = _prefix_pass(gp, Xs, J, kind=_lib.ACQ_EI, p0=fb, p1=0.0)
        # (EI = imp Phi(z) + sigma phi(z) cancels for z << 0: values of 1e-100 carry rounding noise of their own size - the
        #  bound form rounds outward by that noise, gpbo_acquisition_ub: no tolerance here)
        assert np.all(ae >= ei), (problem, J, float((ei - ae).max()), int(np.argmax(ei - ae)))
        if prev is not None:
            assert np.all(s <= prev + 1e-13)
        prev = s
    # the whole route: same point as the plain pass of the same factorisation AND of the arrival-order one
    ga = DeviceGP(chunk=8192).factorise(X, y, ls)
    for kw in (dict(acquisition="lcb", explore=4.0), dict(acquisition="lcb", explore=0.5), dict(acquisition="ei", f_best=fb)):
        rb, r64 = gp.score_bound(Xs, **kw), gp.score(Xs, **kw)
        assert rb.best_idx == r64.best_idx and rb.nan_count == r64.nan_count == 0
        assert abs(rb.best_val - r64.best_val) <= 1e-9 * max(1.0, abs(r64.best_val))
        ra = ga.score(Xs, **kw)
        if problem != "ill_conditioned":      # (there the two factorisations' values differ by cond * eps: ties may flip)
            assert ra.best_idx == rb.best_idx


def test_second_level_bound_of_the_headline_size_for_every_candidate():
    """N = 4096: J = 256 (first level) and 1024 (second level) exactly as bench.py --dtype f64b runs them."""
    X, y, Xs, ls = make_problem(4096, 20000, 8)
    gp = DeviceGP(chunk=4096).factorise(X, y, ls, order="fps")
    assert gp.bound_prefix() == 256
    full = gp.score(Xs, dense=True)
    sig, acq = full.sigma.cpu().numpy(), full.acq.cpu().numpy()
    s1 = None
    for J in (256, 1024):
        m, s, a = _prefix_pass(gp, Xs, J)
        assert np.all(s >= sig) and np.all(a >= acq)
        s1 = s if s1 is None else s1
    assert np.all(s <= s1 + 1e-13) and np.mean(s - sig) < np.mean(s1 - sig)
    rb = gp.score_bound(Xs)
    assert gp.last_screen["prefix"] == 256 and gp.last_screen["prefix2"] == 1024 and rb.best_idx == full.best_idx


def test_state_loaded_into_a_used_surrogate_takes_its_own_order_with_it():
    """score_bound, then load_state of a DIFFERENT surrogate with the same padded size, then
    score_bound again - the bound must belong to the loaded observations (round 3 kept the old observation subset and could
    prune the true arg-max; there is no such cached state any more: the order is part of the factorisation and of its
    state)."""
    N, M, d = 1200, 40000, 5
    X1, y1, Xs, ls1 = make_problem(N, M, d)
    rng = np.random.default_rng(9)
    X2 = rng.uniform(0, 1, (N, d))
    X2 = X2[np.argsort(X2[:, 1])]
    y2 = np.cos(7 * X2[:, 0]) - X2[:, 2] + 0.01 * rng.standard_normal(N)
    ls2 = np.array([0.3, 0.5, 0.2, 0.9, 0.4])
    other = DeviceGP(chunk=8192).factorise(X2, y2, ls2, order="fps")
    st = other.state_dict()
    assert "perm" in st and np.array_equal(np.sort(st["perm"]), np.arange(N))
    gp = DeviceGP(chunk=8192).factorise(X1, y1, ls1, order="fps")
    r1 = gp.score_bound(Xs)
    assert r1.best_idx == gp.score(Xs).best_idx
    gp.load_state_dict(st)
    assert gp.order == "fps" and np.array_equal(gp.perm.cpu().numpy(), st["perm"])
    Xa, ya = gp.observations_host()
    assert np.array_equal(Xa, X2) and np.array_equal(ya, y2)
    for kw in (dict(acquisition="lcb", explore=4.0), dict(acquisition="ei", f_best=float(y2.min()))):
        rb, r64, ro = gp.score_bound(Xs, **kw), gp.score(Xs, **kw), other.score(Xs, **kw)
        assert not gp.last_screen["fallback"]
        assert rb.best_idx == r64.best_idx == ro.best_idx and rb.best_val == pytest.approx(ro.best_val, rel=1e-12)
    # an appended observation goes last in both orders; the state of an arrival-order surrogate carries no permutation
    gp.append(np.full(d, 0.5), 0.1)
    assert gp.perm.shape[0] == N + 1 and int(gp.perm[-1]) == N
    Xa, ya = gp.observations_host()
    assert np.array_equal(Xa[:N], X2) and np.array_equal(Xa[N], np.full(d, 0.5)) and ya[N] == 0.1
    assert gp.score_bound(Xs).best_idx == gp.score(Xs).best_idx
    assert "perm" not in DeviceGP().factorise(X1, y1, ls1).state_dict()


def test_routes_the_bound_must_not_take_with_a_permuted_factorisation():
    """The N == M quirk is keyed on the arrival index (point_selector.py:173): a permuted factorisation refuses it; a jitter
    too small for the |var| rule sends score_bound to the plain pass."""
    X, y, Xs, ls = make_problem(600, 600, 4)
    gp = DeviceGP(chunk=1024).factorise(X, y, ls, order="fps")
    with pytest.raises(ValueError):
        gp.score(Xs, diag_add=1e-4)
    g0 = DeviceGP(chunk=1024).factorise(X, y, ls, 1e-8, 0.0)
    X2, y2, Xs2, ls2 = make_problem(600, 20000, 4)
    g0.factorise(X2, y2, ls2, 1e-8, 0.0, order="fps")
    r = g0.score_bound(Xs2, prior_var=1.0 + 1e-8)
    assert g0.last_screen["fallback"] and "jitter" in g0.last_screen["reason"]
    assert r.best_idx == g0.score(Xs2, prior_var=1.0 + 1e-8).best_idx


def test_a_workgroup_that_never_answers_ends_in_the_identity_order_not_in_a_hang():
    """The co-operating workgroups of the selection wait for each other's records with BOUNDED polls.  In a DIAGNOSTICS build
    of the library (ab_libs/diag_fps.so: subset.hip with -DGPBO_DIAGNOSTICS, made by __graft_entry__.build(); the shipped
    library does not read the switch) GPBO_FPS_MUTE makes one of them stay silent, as if it had never been scheduled: the
    others give up after ~1 s, every workgroup leaves, the order that comes back is the identity - the arrival order, with
    which the route is still exact - and the fall-back is REPORTED (gpbo_fps_order_status -> DeviceGP.order_fell_back,
    last_screen["order"] == "arrival (fps fell back)").  (A child process: the switch is read once per process.)"""
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(repo, "ab_libs", "diag_fps.so")
    if not os.path.exists(diag):
        subprocess.run(["bash", os.path.join(repo, "tools", "build_variant.sh"), "diag_fps", "subset", "-DGPBO_DIAGNOSTICS"],
                       check=True, capture_output=True, timeout=600)
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from bayesian_optimisation_amd import DeviceGP\n"
        "from bayesian_optimisation_amd.synthetic import make_problem\n"
        "X, y, Xs, ls = make_problem(9000, 40000, 4)\n"
        "t = time.time(); gp = DeviceGP(chunk=8192).factorise(X, y, ls, order='fps'); torch.cuda.synchronize(); dt = time.time() - t\n"
        "perm = gp.perm.cpu().numpy()\n"
        "rb, r64 = gp.score_bound(Xs), gp.score(Xs)\n"
        "print('RESULT', bool(np.array_equal(perm, np.arange(9000))), rb.best_idx == r64.best_idx, round(dt, 2), "
        "gp.order_fell_back(), gp.last_screen['order'].replace(' ', '_'))\n" % repo)
    base = {k: v for k, v in os.environ.items() if k not in ("GPBO_FPS_MUTE", "GPBO_LIB")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(base, GPBO_FPS_MUTE="3", GPBO_LIB=diag))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1] == "True" and line[2] == "True", out.stdout
    assert float(line[3]) < 60.0
    assert line[4] == "True" and line[5] == "arrival_(fps_fell_back)", out.stdout
    # the SHIPPED library does not read the switch: the same call with it set gives the farthest-point order
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(base, GPBO_FPS_MUTE="3"))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1] == "False" and line[2] == "True" and float(line[3]) < 20.0, out.stdout
    assert line[4] == "False" and line[5] == "fps", out.stdout
