"""End-to-end parity of the HIP path on the MI355X: against the golden vectors produced by the
reference, against the CPU oracle on seeded inputs, and - at BASELINE.json's full sizes - through
size-independent properties (chunk / shard invariance, reported arg-max == first maximum of the dense
acquisition, sub-sampled oracle check)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bayesian_optimisation_amd import DeviceGP, PointSelector  # noqa: E402
from bayesian_optimisation_amd import distributed as D  # noqa: E402
from bayesian_optimisation_amd.synthetic import make_problem  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

# fp64 tolerances: vs the reference |dmu| <= 1e-9 max(1,|y|inf), |dsigma| <= 1e-8,
# |dacq| <= 1e-8 max(1,|y|inf); vs the Cholesky-route oracle an order tighter.
TOL_MU, TOL_SIG, TOL_ACQ = 1e-9, 1e-8, 1e-8


def _check_against(mu, sig, acq, idx, g, tight=1.0):
    ys = max(1.0, float(np.max(np.abs(g["y"]))))
    assert np.max(np.abs(mu - g["mean_func"].ravel())) <= TOL_MU * ys * tight
    assert np.max(np.abs(sig - g["cov_func"].ravel())) <= TOL_SIG * tight
    assert np.max(np.abs(acq - g["acq_func_eval"].ravel())) <= TOL_ACQ * ys * tight
    if g["top2_gap"] > 1e-7 * ys or g["n_max_ties"] > 1:
        assert idx == int(np.ravel_multi_index(tuple(g["index"]), g["mean_func"].shape))


def _first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


@pytest.mark.parametrize("name", ["g5_d8_n64_m1024", "g5_d8_n512_m4096", "g5_d8_n2048_m4096", "g6_d16_n256_m2048",
                                  "g9_d24_n96_m512"])   # the last: 24 features, the any-d kernels (point_selector.py:22)
def test_fused_path_vs_reference_golden(golden, name):
    g = golden(name)
    X, y, Xs, ls = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
    gp = DeviceGP().factorise(X, y, ls)
    r = gp.score(Xs, acquisition="lcb", explore=4.0, dense=True)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    assert r.nan_count == 0
    _check_against(mu, sig, acq, r.best_idx, g)
    assert r.best_idx == _first_argmax(acq) and r.best_val == acq.max()
    # and an order tighter against the Cholesky-route oracle
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(mu - mu_o)) <= 1e-10 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(sig - sig_o)) <= 1e-9
    assert np.array_equal(acq, 4.0 * sig - mu)  # the LCB line itself is bit-exact given mu, sigma


@pytest.mark.parametrize("N,M,d,chunk", [(1, 50, 1, 512), (7, 2500, 2, 1024), (129, 1300, 5, 512), (300, 5000, 8, 2048)])
def test_fused_path_vs_oracle_ragged_sizes(N, M, d, chunk):
    rng = np.random.default_rng(N * 7 + M)
    X = rng.uniform(0, 1, (N, d))
    Xs = rng.uniform(0, 1, (M, d))
    y = 10.0 * rng.standard_normal(N)
    ls = np.geomspace(0.15, 0.9, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score(Xs, dense=True, idx_offset=1000)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(mu - mu_o)) <= 1e-10 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(sig - sig_o)) <= 1e-9
    assert r.best_idx == 1000 + _first_argmax(acq) and r.best_val == acq.max()
    acq_o = O.lcb(mu_o, sig_o, 4)
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx - 1000 == _first_argmax(acq_o)


def test_expected_improvement_vs_oracle():
    X, y, Xs, ls = make_problem(256, 4096, 8)
    gp = DeviceGP().factorise(X, y, ls)
    r = gp.score(Xs, acquisition="ei", f_best=float(y.min()), xi=0.01, dense=True)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    ei_same_post = O.expected_improvement(mu, sig, float(y.min()), 0.01)
    np.testing.assert_allclose(acq, ei_same_post, rtol=1e-12, atol=1e-15)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    ei_o = O.expected_improvement(mu_o, sig_o, float(y.min()), 0.01)
    assert np.max(np.abs(acq - ei_o)) <= 1e-8
    assert r.best_idx == _first_argmax(acq)
    top2 = np.sort(ei_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx == _first_argmax(ei_o)


def test_ties_resolve_to_lowest_global_index(golden):
    g = golden("g4_tie_tiny_ls")
    gp = DeviceGP(chunk=512).factorise(g["X"], g["y"], g["kernel_params"])
    r = gp.score(g["Xs"], dense=True)
    acq = r.acq.cpu().numpy()
    assert np.array_equal(acq, g["acq_func_eval"].ravel())  # k* underflows to exactly 0: 2,500 identical values
    assert r.best_idx == 0
    # the same tie split over two "ranks": the lower shard must win
    lo0, hi0 = D.shard_bounds(2500, 2, 0)
    lo1, hi1 = D.shard_bounds(2500, 2, 1)
    r0 = gp.score(g["Xs"][lo0:hi0], idx_offset=lo0)
    r1 = gp.score(g["Xs"][lo1:hi1], idx_offset=lo1)
    assert D.reduce_records([(r1.best_val, r1.best_idx, 0), (r0.best_val, r0.best_idx, 0)])[1] == 0


def test_nan_is_counted_not_hidden():
    X, y, Xs, ls = make_problem(16, 128, 2)
    y = y.copy()
    y[3] = np.nan
    gp = DeviceGP().factorise(X, y, ls)
    r = gp.score(Xs)
    assert r.nan_count == 128


def test_not_positive_definite_raises():
    X = np.zeros((4, 2))
    X[:, 0] = [0.0, 0.0, 1.0, 1.0]  # duplicated rows, and a negative jitter to break definiteness
    with pytest.raises(np.linalg.LinAlgError):
        DeviceGP().factorise(X, np.ones(4), np.array([1.0, 1.0]), jitter1=-1e-3, jitter2=0.0)


# ----------------------------------------------------------------------------------------------
# the drop-in class against the reference's golden outputs
# ----------------------------------------------------------------------------------------------
def _run_dropin(g, preset=False):
    ps = PointSelector()
    ps.name = "T"
    ps.iteration = 0
    ps.measured_pts = g["X"]
    ps.measured_vals = g["y"]
    ps.feature_domain = [int(v) for v in g["feature_domain"]]
    ps.predicted_pts = g["Xs"]
    if preset:
        ps.set_kernel_params(g["kernel_params"] if "kernel_params" in g else g["ls"])
    else:
        ps.length_scales = g["length_scales"]
    ps.update_surrogate()
    idx = ps.lower_confidence_bound(float(g["explore"])) if "explore" in g else ps.lower_confidence_bound()
    return ps, idx


# (G10, round 4: randomised DAG-shaped cases run through the reference - placeholder objective 10000 in the last row,
#  duplicated grid points, exploration weights 0.5 / 1 / 2 / 4, 1,514- and 10-way ties)
@pytest.mark.parametrize("name", ["g1_m32", "g1_m50", "g4_ard_n2", "g2_n1_tr", "g2_n5_a", "g2_n20_tr", "g2_n12_a",
                                  "g3_n1_2d", "g10_2d_0", "g10_2d_1", "g10_2d_2", "g10_2d_3", "g10_2d_4", "g10_1d_0", "g10_1d_1", "g10_1d_2"])
def test_dropin_full_path_with_ard(golden, name):
    g = golden(name)
    ps, idx = _run_dropin(g)
    assert np.array_equal(np.asarray(ps.kernel_params), g["kernel_params"])
    assert ps.kernel_params.shape == g["kernel_params"].shape
    assert idx.dtype == np.int64 and idx.shape == g["index"].shape
    assert ps.mean_func.shape == g["mean_func"].shape
    _check_against(ps.mean_func.ravel(), ps.cov_func.ravel(), ps.acq_func_eval.ravel(),
                   int(np.ravel_multi_index(tuple(idx), g["mean_func"].shape)), g)
    assert isinstance(ps.measured_pts, list) and isinstance(ps.measured_vals, list)  # point_selector.py:101-102
    assert ps.cov_meas.shape == (len(g["X"]), len(g["X"]))
    assert ps.cov_pred is not None and ps.cov_pred.shape == (len(g["Xs"]), len(g["Xs"]))
    assert ps.cov_meas_pred.shape == (len(g["Xs"]), len(g["X"]))
    if "nlogml" in g:
        np.testing.assert_allclose(ps.nlogml, g["nlogml"], rtol=2e-6)


@pytest.mark.parametrize("name", ["g11_2d_n64", "g11_2d_n100"])
def test_dropin_full_path_with_ard_beyond_one_panel_of_the_fused_kernel(golden, name):
    """G11 (round 5): the reference run with 64 and 100 observations - the fused likelihood kernel's largest one-panel case
    and a two-panel case.  At N = 100 half of the reference's float32 grid is -inf (np.linalg.det underflows,
    point_selector.py:117-119) and its ARD choice is the first such cell: the same choice, the same posterior, the same
    index; the grid itself equal wherever the reference's determinant is a normal number, -inf where it is zero, and left
    out where it is a denormal (log det in [-745, -708]: the reference's own value is rounding noise there)."""
    g = golden(name)
    ps, idx = _run_dropin(g)
    assert np.array_equal(np.asarray(ps.kernel_params), g["kernel_params"]) and ps.kernel_params.shape == g["kernel_params"].shape
    _check_against(ps.mean_func.ravel(), ps.cov_func.ravel(), ps.acq_func_eval.ravel(),
                   int(np.ravel_multi_index(tuple(idx), g["mean_func"].shape)), g)
    ref = g["nlogml"]
    assert ps.nlogml.dtype == np.float32 and ps.nlogml.shape == ref.shape
    lsg = g["length_scales"]
    cells = np.stack(np.meshgrid(lsg[0], lsg[1], indexing="ij"), -1).reshape(-1, 2)
    logdet = np.array([np.linalg.slogdet(O.kernel_rbf(g["X"], g["X"], c))[1] for c in cells]).reshape(ref.shape)
    normal = logdet > -707.0
    zero = logdet < -746.0
    assert normal.sum() + zero.sum() >= ref.size - 300 and normal.sum() >= 900   # (N = 100: 960 normal, 1,256 zero, 284 between)
    np.testing.assert_allclose(ps.nlogml[normal], ref[normal], rtol=2e-6)
    assert np.all(np.isneginf(ref[zero])) and np.all(np.isneginf(ps.nlogml[zero]))
    assert np.array_equal(np.argwhere(ps.nlogml == ps.nlogml.min())[0], np.argwhere(ref == ref.min())[0])


@pytest.mark.parametrize("name", ["g4_tie_tiny_ls", "g4_dup_rows", "g7_n_eq_m", "g9_d24_n96_m512"])
def test_dropin_preset_length_scales(golden, name):
    g = golden(name)
    if "X" not in g:   # G5/G6/G9 store the generator's arguments, not the arrays
        g["X"], _, g["Xs"], _ = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
        g["feature_domain"] = np.array([int(g["M"])])
    ps, idx = _run_dropin(g, preset=True)
    _check_against(ps.mean_func.ravel(), ps.cov_func.ravel(), ps.acq_func_eval.ravel(),
                   int(np.ravel_multi_index(tuple(idx), g["mean_func"].shape)), g)
    if name == "g7_n_eq_m":  # shape-coincidence quirk: +1e-4 on the diagonal of K(X*, X)
        np.testing.assert_allclose(np.diag(ps.cov_meas_pred), g["cov_meas_pred_diag"], rtol=0, atol=1e-15)


def test_dropin_nan_raises_index_error(golden):
    g = golden("g8_nan")
    with pytest.raises(IndexError):
        _run_dropin(g, preset=True)


def test_dropin_second_acquisition_matches_numpy_line():
    g_X, g_y, g_Xs, ls = make_problem(40, 900, 2)
    ps = PointSelector()
    ps.measured_pts, ps.measured_vals = g_X, g_y
    ps.feature_domain, ps.predicted_pts = [30, 30], g_Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    i4 = ps.lower_confidence_bound()
    a4 = ps.acq_func_eval.copy()
    i1 = ps.lower_confidence_bound(explore=1)
    assert np.array_equal(a4, 4 * ps.cov_func - ps.mean_func)
    assert np.array_equal(ps.acq_func_eval, 1 * ps.cov_func - ps.mean_func)
    assert np.array_equal(i4, np.argwhere(a4 == a4.max())[0])
    assert np.array_equal(i1, np.argwhere(ps.acq_func_eval == ps.acq_func_eval.max())[0])
    ie = ps.expected_improvement()
    ei = O.expected_improvement(ps.mean_func, ps.cov_func, float(np.min(g_y)))
    np.testing.assert_allclose(ps.acq_func_eval, ei, rtol=1e-12, atol=1e-15)
    assert np.array_equal(ie, np.argwhere(ps.acq_func_eval == ps.acq_func_eval.max())[0])


# ----------------------------------------------------------------------------------------------
# BASELINE.json config 2 at full size: d=8, N=512, M=2^20 - size-independent properties
# ----------------------------------------------------------------------------------------------
def test_config2_full_size_properties():
    N, M, d = 512, 1 << 20, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=1 << 17).factorise(X, y, ls)
    r = gp.score(Xs, dense=True)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    assert r.nan_count == 0 and np.isfinite(acq).all()
    # (1) reported arg-max is the first maximum of the dense acquisition
    assert r.best_idx == _first_argmax(acq) and r.best_val == acq.max()
    # (2) chunk-size invariance, bit for bit
    gp2 = DeviceGP(chunk=1 << 15).factorise(X, y, ls)
    r2 = gp2.score(Xs, dense=True)
    assert np.array_equal(r2.acq.cpu().numpy(), acq) and r2.best_idx == r.best_idx
    # (3) shard invariance: 8 contiguous shards + the lexicographic reduce == single call
    recs = []
    for rank in range(8):
        lo, hi = D.shard_bounds(M, 8, rank)
        rr = gp.score(Xs[lo:hi], idx_offset=lo)
        recs.append((rr.best_val, rr.best_idx, rr.nan_count))
    assert D.reduce_records(recs)[:2] == (r.best_val, r.best_idx)
    # (4) oracle on a seeded sub-sample that includes the winner and its runner-ups
    rng = np.random.default_rng(0)
    top = np.argsort(acq)[-64:]
    sub = np.unique(np.concatenate([rng.choice(M, 8192, replace=False), top]))
    mu_o, sig_o = O.posterior_chol(X, y, Xs[sub], ls)
    assert np.max(np.abs(mu[sub] - mu_o)) <= 1e-10 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(sig[sub] - sig_o)) <= 1e-9
    acq_o = O.lcb(mu_o, sig_o, 4)
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert sub[_first_argmax(acq_o)] == r.best_idx


# ----------------------------------------------------------------------------------------------
# fp32 scoring path (BASELINE config 4 shape: d=16; fp64 factorisation, fp32 M-proportional work)
# ----------------------------------------------------------------------------------------------
# fp32 mode = fp32 SCREEN of the variance product + fp64 decision (csrc/rescore.hip).  Tolerances, written here (SURVEY G6
# "tolerance widened"): the mean is the fp64 path's bit for bit; sigma carries the fp32 rounding of K*, U and a length-N
# fp32 accumulation, |dsigma| <= 5e-3; the SELECTED POINT is the fp64 path's exactly (index and value), hence the
# oracle's first arg-max whenever the oracle's top-2 gap exceeds the fp64 noise (1e-7, as everywhere in this file).
@pytest.mark.parametrize("N,M,d,chunk", [(256, 2048, 16, 1024), (300, 5000, 8, 2048), (40, 900, 2, 1024),
                                         (1024, 4096, 16, 4096), (500, 20000, 3, 4096)])
def test_fp32_path_vs_oracle(N, M, d, chunk):
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score_f32(Xs, dense=True, idx_offset=7)
    assert r.mu.dtype == gp.torch.float64
    mu, sig, acq = (v.cpu().numpy() for v in (r.mu, r.sigma, r.acq))
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    acq_o = O.lcb(mu_o, sig_o, 4)
    scale = max(1.0, float(np.abs(y).max()))
    r64 = gp.score(Xs, dense=True, idx_offset=7)
    assert np.array_equal(mu, r64.mu.cpu().numpy())          # fp64 mean, same kernel arithmetic
    assert np.max(np.abs(sig - sig_o)) <= 5e-3
    assert np.max(np.abs(acq - acq_o)) <= 2e-2 * scale
    assert r.nan_count == 0
    # the decision is the fp64 kernels' (their column-split launch: a different summation order, ~1e-16 relative)
    assert r.best_idx == r64.best_idx and abs(r.best_val - r64.best_val) <= 1e-12 * scale
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx == 7 + _first_argmax(acq_o)
    st = gp.last_screen
    assert not st["fallback"] and 4 * st["err_max"] <= st["tau"]
    assert st["survivors"] < M // 2 or M <= 2048   # (below ~1,000 candidates the strided check sample is everyone)
    # EI through the same screen
    f_best = float(y.min())
    e32 = gp.score_f32(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    e64 = gp.score(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    assert e32.best_idx == e64.best_idx and abs(e32.best_val - e64.best_val) <= 1e-12 * scale
    assert np.max(np.abs(r64.sigma.cpu().numpy() - sig_o)) <= 1e-9


def test_fp32_screen_falls_back_to_the_fp64_pass_when_too_many_candidates_survive(golden):
    """2,500 exact ties (k* == 0 everywhere): nothing can be screened out; with a cap below that the call must hand the
    decision to the plain fp64 pass - index 0, as the reference (np.argwhere(...)[0])."""
    g = golden("g4_tie_tiny_ls")
    gp = DeviceGP(chunk=1024).factorise(g["X"], g["y"], g["kernel_params"])
    gp.screen_cap = 600
    r = gp.score_f32(g["Xs"], dense=True)
    assert gp.last_screen["fallback"] and gp.last_screen["survivors"] == len(g["Xs"])
    assert r.best_idx == 0 and r.nan_count == 0
    gp.screen_cap = None
    r = gp.score_f32(g["Xs"], dense=True)
    assert not gp.last_screen["fallback"] and r.best_idx == 0


def test_fp32_screen_raises_its_tolerance_when_the_first_guess_is_too_small():
    X, y, Xs, ls = make_problem(1024, 8192, 16)
    gp = DeviceGP(chunk=4096).factorise(X, y, ls)
    gp.SCREEN_TAU0 = 1e-12   # far below the fp32 error of the variance: the check must catch it and widen
    r = gp.score_f32(Xs)
    r64 = gp.score(Xs)
    st = gp.last_screen
    assert st["rounds"] > 1 and st["tau"] > 1e-12 and 4 * st["err_max"] <= st["tau"] and not st["fallback"]
    assert r.best_idx == r64.best_idx and abs(r.best_val - r64.best_val) <= 1e-12 * max(1.0, abs(r64.best_val))


def test_fp32_screen_counts_nan_candidates():
    X, y, Xs, ls = make_problem(64, 3000, 4)
    Xs = Xs.copy()
    Xs[1234, 2] = np.nan
    r = DeviceGP(chunk=1024).factorise(X, y, ls).score_f32(Xs)
    assert r.nan_count == 1 and r.best_idx != 1234


def test_fp32_chunk_invariance_and_ties(golden):
    X, y, Xs, ls = make_problem(200, 6000, 8)
    a = DeviceGP(chunk=1024).factorise(X, y, ls).score_f32(Xs, dense=True)
    b = DeviceGP(chunk=4096).factorise(X, y, ls).score_f32(Xs, dense=True)
    assert np.array_equal(a.acq.cpu().numpy(), b.acq.cpu().numpy()) and a.best_idx == b.best_idx
    g = golden("g4_tie_tiny_ls")
    r = DeviceGP(chunk=1024).factorise(g["X"], g["y"], g["kernel_params"]).score_f32(g["Xs"], dense=True)
    assert r.best_idx == 0 and len(np.unique(r.acq.cpu().numpy())) == 1


# ----------------------------------------------------------------------------------------------
# q = 8 Monte-Carlo qEI (BASELINE config 5 shape: d=8; parity pinned by the oracle's restatement only)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,M,d,chunk,S", [(64, 1024, 8, 512, 512), (300, 2048, 8, 1024, 512), (33, 808, 3, 512, 100)])
def test_qei_vs_oracle(N, M, d, chunk, S):
    X, y, Xs, ls = make_problem(N, M, d)
    Z = O.qei_base_samples(S, 8, 7)
    f_best = float(y.min())
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score_qei(Xs, Z, f_best, xi=0.0, dense=True, batch_offset=5)
    got = r.acq.cpu().numpy()
    ref = O.qei_mc(X, y, Xs, ls, Z, f_best)
    assert r.nan_count == 0 and got.shape == (M // 8,)
    assert np.max(np.abs(got - ref)) <= 1e-9 * max(1.0, np.abs(y).max())   # fp64: same samples, same algebra
    assert r.best_idx == 5 + _first_argmax(got) and r.best_val == got.max()
    top2 = np.sort(ref)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx - 5 == _first_argmax(ref)


def test_qei_nan_and_single_sample_limit():
    X, y, Xs, ls = make_problem(50, 512, 4)
    gp = DeviceGP().factorise(X, y, ls)
    # one sample z = 0: qEI_b = max(0, max_j (f_best - mu_j)) exactly
    r = gp.score_qei(Xs, np.zeros((1, 8)), 0.3, dense=True)
    mu = gp.score(Xs, dense=True).mu.cpu().numpy().reshape(-1, 8)
    np.testing.assert_allclose(r.acq.cpu().numpy(), np.maximum(0.0, (0.3 - mu).max(1)), rtol=0, atol=1e-12)
    y2 = y.copy()
    y2[0] = np.nan
    r2 = DeviceGP().factorise(X, y2, ls).score_qei(Xs, O.qei_base_samples(64), 0.0)
    assert r2.nan_count == 512 // 8


def test_config3_shape_n4096_subsampled():
    """BASELINE config 3 per-GPU shape (d=8, N=4096): oracle on a sub-sample that contains the top candidates."""
    N, M, d = 4096, 1 << 16, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=1 << 15).factorise(X, y, ls)
    r = gp.score(Xs, dense=True)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    assert r.nan_count == 0 and r.best_idx == _first_argmax(acq)
    sub = np.unique(np.concatenate([np.random.default_rng(3).choice(M, 2048, replace=False), np.argsort(acq)[-32:]]))
    mu_o, sig_o = O.posterior_chol(X, y, Xs[sub], ls)
    assert np.max(np.abs(mu[sub] - mu_o)) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(sig[sub] - sig_o)) <= 1e-8
    acq_o = O.lcb(mu_o, sig_o, 4)
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert sub[_first_argmax(acq_o)] == r.best_idx


def test_config4_shape_n8192_fp32_subsampled():
    """BASELINE config 4 per-GPU shape (d=16, N=8192, fp64 factorisation + fp32 scoring): oracle on a sub-sample;
    chunk invariance of the arg-max; the fp64 factorisation itself against the oracle's alpha."""
    N, M, d = 8192, 1 << 14, 16
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=1 << 13).factorise(X, y, ls)
    _, _, alpha_o = O.factorise(X, y, ls)
    a = gp.alpha[:N].cpu().numpy()
    assert np.max(np.abs(a - alpha_o)) <= 1e-7 * np.abs(alpha_o).max()
    r = gp.score_f32(Xs, dense=True)
    mu, sig, acq = r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()
    r64 = gp.score(Xs)
    assert r.nan_count == 0 and r.best_idx == r64.best_idx                                  # the fp64 decision
    assert abs(r.best_val - r64.best_val) <= 1e-12 * max(1.0, abs(r64.best_val))
    assert not gp.last_screen["fallback"] and gp.last_screen["survivors"] < M // 4
    sub = np.unique(np.concatenate([np.random.default_rng(4).choice(M, 256, replace=False), np.argsort(acq)[-16:]]))
    mu_o, sig_o = O.posterior_chol(X, y, Xs[sub], ls)
    acq_o = O.lcb(mu_o, sig_o, 4)
    assert np.max(np.abs(mu[sub] - mu_o)) <= 1e-9 * max(1.0, np.abs(y).max()) + 1e-12 * float(np.abs(alpha_o).sum())
    assert np.max(np.abs(sig[sub] ** 2 - sig_o ** 2)) <= 5e-3
    # against the fp32 CPU restatement of the same arithmetic: both carry fp32 error of the same size
    _, sig_o32 = O.posterior_chol(X, y, Xs[sub], ls, variance_dtype=np.float32)
    assert np.max(np.abs(sig[sub] ** 2 - sig_o32 ** 2)) <= 5e-3
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert sub[_first_argmax(acq_o)] == r.best_idx     # the oracle's first arg-max (the fp32 top-16 are in `sub`)
    r2 = DeviceGP(chunk=1 << 12).factorise(X, y, ls).score_f32(Xs)
    assert (r2.best_idx, r2.best_val) == (r.best_idx, r.best_val)


def test_config5_shape_qei_n2048_subsampled():
    """BASELINE config 5 shape (q=8, S=512, d=8, N=2048): oracle qEI on a sub-sample of the batches."""
    N, M, d = 2048, 1 << 13, 8
    X, y, Xs, ls = make_problem(N, M, d)
    Z = O.qei_base_samples(512, 8, 7)
    f_best = float(y.min())
    gp = DeviceGP(chunk=1 << 12).factorise(X, y, ls)
    r = gp.score_qei(Xs, Z, f_best, dense=True)
    got = r.acq.cpu().numpy()
    assert r.nan_count == 0 and r.best_idx == _first_argmax(got) and r.best_val == got.max()
    batches = np.unique(np.concatenate([np.random.default_rng(5).choice(M // 8, 48, replace=False), np.argsort(got)[-8:]]))
    rows = (batches[:, None] * 8 + np.arange(8)).ravel()
    ref = O.qei_mc(X, y, Xs[rows], ls, Z, f_best)
    assert np.max(np.abs(got[batches] - ref)) <= 1e-8 * max(1.0, np.abs(y).max())
    top2 = np.sort(ref)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert batches[_first_argmax(ref)] == r.best_idx


# ----------------------------------------------------------------------------------------------
# edge shapes: padding boundaries of N (128), of the candidate tiles (256) and chunks (512), every feature count
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [2, 127, 128, 129, 256, 257])
@pytest.mark.parametrize("M", [1, 255, 256, 257, 511, 513, 1025])
def test_padding_boundaries(N, M):
    rng = np.random.default_rng(N * 1000 + M)
    d = 3
    X, Xs = rng.uniform(0, 1, (N, d)), rng.uniform(0, 1, (M, d))
    y = rng.standard_normal(N)
    ls = np.array([0.25, 0.5, 1.0])
    gp = DeviceGP(chunk=512).factorise(X, y, ls)
    # N == M: the reference's kernel_rbf adds its 1e-4 jitter to K(X, X*) too (shape-equality rule); the oracle
    # reproduces that and the caller of the C ABI asks for it through diag_add (PointSelector does)
    quirk = 1e-4 if N == M else 0.0
    r = gp.score(Xs, dense=True, diag_add=quirk)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(r.mu.cpu().numpy() - mu_o)) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r.sigma.cpu().numpy() - sig_o)) <= 1e-8
    acq = r.acq.cpu().numpy()
    assert r.nan_count == 0 and r.best_idx == _first_argmax(acq) and r.best_val == acq.max()
    r32 = gp.score_f32(Xs, dense=True, diag_add=quirk)
    assert np.max(np.abs(r32.sigma.cpu().numpy() - sig_o)) <= 5e-3
    assert r32.best_idx == _first_argmax(r32.acq.cpu().numpy())


@pytest.mark.parametrize("d", list(range(1, 17)))
def test_every_feature_count(d):
    X, y, Xs, ls = make_problem(70, 600, d)
    gp = DeviceGP(chunk=512).factorise(X, y, ls)
    r = gp.score(Xs, dense=True)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(r.mu.cpu().numpy() - mu_o)) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r.sigma.cpu().numpy() - sig_o)) <= 1e-8
    np.testing.assert_allclose(gp.cov_meas_host(), O.kernel_rbf(X, X, ls) + 1e-6 * np.eye(70), rtol=0, atol=5e-15)
    r32 = gp.score_f32(Xs, dense=True)
    assert np.max(np.abs(r32.mu.cpu().numpy() - mu_o)) <= 5e-3 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("N,M,d", [(60, 3000, 17), (300, 5000, 40), (33, 700, 130)])
def test_any_feature_count_on_the_fp64_route(N, M, d):
    """point_selector.py:22: the reference's class is agnostic to the dimensionality of the feature space.  Beyond the 16
    unrolled feature counts the fp64 route runs on the slow any-d kernels: same tolerances against the oracle, same
    selected point, chunk invariance; the other routes say so instead of mis-computing."""
    rng = np.random.default_rng(d)
    X = rng.uniform(0, 1, (N, d))
    Xs = rng.uniform(0, 1, (M, d))
    y = rng.standard_normal(N)
    ls = np.exp(rng.uniform(np.log(0.8), np.log(3.0), d)) * np.sqrt(d / 8.0)
    gp = DeviceGP(chunk=1024).factorise(X, y, ls)
    r = gp.score(Xs, dense=True, idx_offset=3)
    mu, sig, acq = (t.cpu().numpy() for t in (r.mu, r.sigma, r.acq))
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    acq_o = O.lcb(mu_o, sig_o, 4)
    assert np.max(np.abs(mu - mu_o)) <= 1e-10 * max(1.0, np.abs(y).max()) and np.max(np.abs(sig - sig_o)) <= 1e-9
    assert r.best_idx == 3 + _first_argmax(acq) and r.nan_count == 0
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx == 3 + _first_argmax(acq_o)
    np.testing.assert_allclose(gp.cov_meas_host(), O.kernel_rbf(X, X, ls) + 1e-6 * np.eye(N), rtol=0, atol=5e-15)
    r2 = DeviceGP(chunk=512).factorise(X, y, ls).score(Xs, dense=True, idx_offset=3)
    assert np.array_equal(r2.acq.cpu().numpy(), acq) and r2.best_idx == r.best_idx
    e = gp.score(Xs, acquisition="ei", f_best=float(y.min()), xi=0.0, dense=True)
    ei_o = O.expected_improvement(mu_o, sig_o, float(y.min()), 0.0)
    assert np.max(np.abs(e.acq.cpu().numpy() - ei_o)) <= 1e-8 * max(1.0, np.abs(y).max())
    # the drop-in class on the same data (the reference's call path is exactly factorise + LCB)
    ps = PointSelector()
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [M], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    assert ps.lower_confidence_bound()[0] == _first_argmax(acq) and np.array_equal(ps.mean_func, mu)
    # routes that need the unrolled kernels refuse; the bound route hands over to the plain pass
    for call in (lambda: gp.score_f32(Xs), lambda: gp.score_i8(Xs), lambda: gp.score_i8c(Xs), lambda: gp.append(X[0], 0.0)):
        with pytest.raises(ValueError):
            call()
    rb = gp.score_bound(Xs, idx_offset=3)
    assert rb.best_idx == r.best_idx and gp.last_screen["fallback"]


def test_unsupported_feature_count_raises():
    X, y, Xs, ls = make_problem(10, 20, 2)
    with pytest.raises(ValueError):
        DeviceGP().factorise(np.zeros((10, 1100)), y, np.ones(1100))


# ----------------------------------------------------------------------------------------------
# N > 1 end to end on one GPU: two gloo ranks share cuda:0, each scores its contiguous shard through the
# real kernels; the drop-in class must return exactly what a single process returns.
# ----------------------------------------------------------------------------------------------
def _sharded_worker(rank, world, port, q):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, Xs, ls = make_problem(48, 2400, 2)
        ps = PointSelector(device="cuda:0")
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [50, 48], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        idx = ps.lower_confidence_bound()
        idx2 = ps.lower_confidence_bound(explore=1.5)
        acq2 = ps.acq_func_eval.copy()
        qpts = ps.q_expected_improvement(n_samples=64, seed=7)
        q.put((rank, idx.tolist(), idx2.tolist(), ps.mean_func.copy(), ps.cov_func.copy(), acq2, qpts.tolist(),
               ps.acq_func_eval.copy()))
    finally:
        dist.destroy_process_group()


def test_dropin_sharded_over_two_ranks_matches_single_process():
    import socket

    import torch.multiprocessing as mp

    X, y, Xs, ls = make_problem(48, 2400, 2)
    ps = PointSelector()
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [50, 48], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    ref_idx = ps.lower_confidence_bound().tolist()
    ref_mu, ref_sd = ps.mean_func.copy(), ps.cov_func.copy()
    ref_idx2 = ps.lower_confidence_bound(explore=1.5).tolist()
    ref_acq2 = ps.acq_func_eval.copy()
    ref_qpts = ps.q_expected_improvement(n_samples=64, seed=7).tolist()
    ref_qei = ps.acq_func_eval.copy()

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, idx, idx2, mu, sd, acq2, qpts, qei in res:
        assert idx == ref_idx and idx2 == ref_idx2
        assert np.array_equal(mu, ref_mu) and np.array_equal(sd, ref_sd) and np.array_equal(acq2, ref_acq2)
        assert qpts == ref_qpts and np.array_equal(qei, ref_qei)


def _sharded_big_worker(rank, world, port, q):
    import hashlib
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, Xs, ls = make_problem(256, 1 << 22, 4)
        ps = PointSelector(device="cuda:0")
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [1 << 22], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        idx = ps.lower_confidence_bound()
        h = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (ps.mean_func, ps.cov_func, ps.acq_func_eval)]
        q.put((rank, idx.tolist(), h, ps.mean_func.shape))
    finally:
        dist.destroy_process_group()


def test_dropin_sharded_at_m_2e22_gathers_the_dense_outputs_as_tensors():
    """M = 2^22 candidates over two ranks (one GPU, gloo): the three dense attributes the
    reference's caller reads (select_parameters.py:167-169) come back through ONE tensor collective per call
    (distributed.gather_concat_tensors), identical to the single-process arrays bit for bit."""
    import hashlib
    import socket

    import torch.multiprocessing as mp

    X, y, Xs, ls = make_problem(256, 1 << 22, 4)
    ps = PointSelector()
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [1 << 22], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    ref_idx = ps.lower_confidence_bound().tolist()
    ref_h = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (ps.mean_func, ps.cov_func, ps.acq_func_eval)]
    del ps
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_big_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, idx, h, shape in res:
        assert idx == ref_idx and h == ref_h and tuple(shape) == (1 << 22,)


def _sharded_select_only_worker(rank, world, port, q):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, Xs, ls = make_problem(1500, 90000, 5)
        X = X[np.argsort(X[:, 0])]           # a sorted history: the arrival prefix would prune little
        ps = PointSelector(dense_outputs=False)
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [300, 300], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        i4 = ps.lower_confidence_bound().tolist()
        scr = dict(ps._gp.last_screen)
        ie = ps.expected_improvement().tolist()
        perm = ps._gp.perm.cpu().numpy()
        q.put((rank, i4, ie, scr["mode"], scr["order"], bool(scr["fallback"]), int(scr["candidates"]), perm[:8].tolist(),
               ps.mean_func is None))
    finally:
        dist.destroy_process_group()


def test_next_point_only_sharded_over_two_ranks():
    """PointSelector(dense_outputs=False) with the candidates sharded over two ranks (gloo, both on the one GPU): every rank
    puts the observations in the same farthest-point order (the selection is deterministic), bounds its own 45,000
    candidates, and the one exchange step gives both the single process's multi-index - for LCB and EI."""
    import socket

    import torch.multiprocessing as mp

    X, y, Xs, ls = make_problem(1500, 90000, 5)
    X = X[np.argsort(X[:, 0])]
    ref = PointSelector()
    ref.measured_pts, ref.measured_vals = X, y
    ref.feature_domain, ref.predicted_pts = [300, 300], Xs
    ref.set_kernel_params(ls)
    ref.update_surrogate()
    want4 = ref.lower_confidence_bound().tolist()
    wante = ref.expected_improvement().tolist()
    one = PointSelector(dense_outputs=False)
    one.measured_pts, one.measured_vals = X, y
    one.feature_domain, one.predicted_pts = [300, 300], Xs
    one.set_kernel_params(ls)
    one.update_surrogate()
    assert one.lower_confidence_bound().tolist() == want4 and one._gp.order == "fps"
    perm_ref = one._gp.perm.cpu().numpy()[:8].tolist()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_select_only_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, i4, ie, mode, order, fallback, cands, perm8, no_dense in res:
        assert i4 == want4 and ie == wante
        assert mode == "bound" and order == "fps" and not fallback and cands == 45000 and no_dense
        assert perm8 == perm_ref


def test_dropin_q_expected_improvement():
    X, y, Xs, ls = make_problem(30, 1600, 2)
    ps = PointSelector()
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [40, 40], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    pts = ps.q_expected_improvement(n_samples=256, seed=7)
    ref = O.qei_mc(X, y, Xs, ls, O.qei_base_samples(256, 8, 7), float(np.min(y)))
    assert pts.shape == (8, 2) and pts.dtype == np.int64
    np.testing.assert_allclose(ps.acq_func_eval, ref, rtol=0, atol=1e-9)
    b = int(np.flatnonzero(ps.acq_func_eval == ps.acq_func_eval.max())[0])
    assert np.array_equal(np.ravel_multi_index(tuple(pts.T), (40, 40)), b * 8 + np.arange(8))


def test_dropin_fp32_precision_mode(golden):
    g = golden("g6_d16_n256_m2048")
    X, y, Xs, ls = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
    ps = PointSelector(precision="fp32")
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [int(g["M"])], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()
    assert ps.mean_func.dtype == np.float64
    assert np.max(np.abs(ps.mean_func - g["mean_func"])) <= 5e-3 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(ps.cov_func - g["cov_func"])) <= 5e-3
    # the selected point is the reference's own (fp64 decision behind the fp32 screen)
    assert idx[0] == _first_argmax(g["acq_func_eval"])


@pytest.mark.parametrize("precision", ["fp32", "i8", "i8c"])
def test_screened_precisions_decide_every_acquisition_in_fp64(precision):
    """EI and LCB with another `explore` under a screened precision: the index is the fp64 selector's (screen + fp64
    re-score with THAT acquisition), not the arg-max of an acquisition built on the screen's sigma; mean_func is the
    fp64 kernels'; cov_func stays inside the documented bound, which last_screen repeats."""
    from bayesian_optimisation_amd.point_selector import SCREEN_SIGMA_TOL

    X, y, Xs, ls = make_problem(700, 6000, 6)
    out = {}
    for prec in ("fp64", precision):
        ps = PointSelector(precision=prec)
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [6000], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        out[prec] = (ps.lower_confidence_bound(), ps.expected_improvement(), ps.expected_improvement(xi=0.05),
                     ps.lower_confidence_bound(explore=1), ps.lower_confidence_bound(explore=10), ps.mean_func.copy(),
                     ps.cov_func.copy(), ps)
    a, b = out["fp64"], out[precision]
    for i in range(5):
        assert np.array_equal(a[i], b[i]), i
    assert np.array_equal(a[5], b[5])
    assert np.max(np.abs(a[6] - b[6])) <= SCREEN_SIGMA_TOL[precision]
    ls_ = b[7].last_screen
    assert ls_["sigma_abs_tol"] == SCREEN_SIGMA_TOL[precision] and not ls_["fallback"] and 4 * ls_["err_max"] <= ls_["tau"]
    assert a[7].last_screen is None


def test_nan_candidate_coordinate_is_counted():
    X, y, Xs, ls = make_problem(20, 600, 3)
    Xs = Xs.copy()
    Xs[17, 1] = np.nan
    r = DeviceGP(chunk=512).factorise(X, y, ls).score(Xs, dense=True)
    assert r.nan_count == 1 and np.isnan(r.acq.cpu().numpy()[17]) and r.best_idx != 17


def test_exchange_step_over_rccl_in_a_one_rank_group():
    """The N>1 exchange (all_gather_into_tensor of device int64 records on the nccl = RCCL backend) exercised on
    the one GPU of this box: a one-rank group with the collective forced.  Bit patterns survive the transit."""
    import os

    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    import socket

    with socket.socket() as sk:  # a free port, not a fixed one (shared hosts)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for rec in [(3.25, 17, 0), (-0.0, 2 ** 40 + 3, 5), (float("-inf"), 0, 0), (1e-310, 9, 1)]:
            v, i, n = D.allreduce_argmax(*rec, force_collective=True)
            assert (i, n) == (rec[1], rec[2]) and np.float64(v).tobytes() == np.float64(rec[0]).tobytes()
        v, i, n = D.allreduce_argmax(float("nan"), 4, 2, force_collective=True)
        assert n == 2 and v == float("-inf")  # a NaN record never wins; it is reported through the count
        # the same exchange straight from DeviceGP's device record (what bench.py's step does)
        X, y, Xs, ls = make_problem(30, 700, 3)
        gp = DeviceGP(chunk=512).factorise(X, y, ls)
        q = gp.score(Xs, idx_offset=11)
        assert D.allreduce_status(gp.status, force_collective=True) == (q.best_val, q.best_idx, 0, 0)
        assert D.allreduce_status(gp.status) == (q.best_val, q.best_idx, 0, 0)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks timing
        dist.barrier()
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()


def test_two_stream_overlap_mode_gives_identical_results():
    """GPBO_OVERLAP=1 (K(X*,X) of chunk c+1 on a helper stream beside the variance kernel of chunk c; opt-in, read once
    per process) must not change a bit."""
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {repo!r})\n"
        "from bayesian_optimisation_amd import DeviceGP\n"
        "from bayesian_optimisation_amd.synthetic import make_problem\n"
        "X, y, Xs, ls = make_problem(200, 5000, 8)\n"
        "r = DeviceGP(chunk=1024).factorise(X, y, ls).score(Xs, dense=True, idx_offset=3)\n"
        "print(r.best_idx, repr(r.best_val), r.nan_count, repr(float(r.sigma.sum().item())), repr(float(r.mu.sum().item())))\n"
    )
    outs = []
    for overlap in ("0", "1"):
        env = dict(os.environ, GPBO_OVERLAP=overlap)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


# ----------------------------------------------------------------------------------------------
# ARD search beyond the reference's two layouts (d > 2)
# ----------------------------------------------------------------------------------------------
def _selector_for(X, y, Xs):
    ps = PointSelector()
    ps.name, ps.iteration = "T", 0
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [len(Xs)], Xs
    return ps


@pytest.mark.parametrize("d,N", [(3, 40), (8, 90), (5, 300)])
def test_ard_coordinate_search_for_more_than_two_features(d, N):
    """length_scales = one axis per feature: coordinate-wise search from the axis mid-points, first minimum per axis,
    two sweeps - the reference's likelihood (point_selector.py:111-120) evaluated on the GPU; same winner as the oracle."""
    X, y, Xs, _ = make_problem(N, 2048, d)
    axes = [np.linspace(0.05 + 0.01 * k, 0.6 + 0.05 * k, 9 + k) for k in range(d)]
    ps = _selector_for(X, y, Xs)
    ps.length_scales = axes
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()
    ls_o, grids_o = O.coordinate_search(X, y, axes, sweeps=2)
    assert np.array_equal(ps.kernel_params, ls_o)
    for g, go in zip(ps.nlogml, grids_o):
        fin = np.isfinite(go)
        np.testing.assert_allclose(g[fin], go[fin], rtol=1e-5, atol=1e-2)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls_o)
    assert idx[0] == _first_argmax(O.lcb(mu_o, sig_o, 4))


def test_ard_explicit_cell_list_d8():
    X, y, Xs, _ = make_problem(60, 1024, 8)
    cells = np.exp(np.random.default_rng(3).uniform(np.log(0.05), np.log(0.8), size=(300, 8)))
    ps = _selector_for(X, y, Xs)
    ps.set_length_scale_cells(cells)
    ps.update_surrogate()
    ref = O.nlml_cells(X, y, cells)
    assert np.array_equal(ps.kernel_params, cells[int(np.argwhere(ref == np.amin(ref))[0][0])])
    fin = np.isfinite(ref)
    np.testing.assert_allclose(ps.nlogml[fin], ref[fin], rtol=1e-5, atol=1e-2)
    # one observation: no search, the middle cell (the reference's mid-point rule, point_selector.py:63-73)
    ps1 = _selector_for(X[:1], y[:1], Xs)
    ps1.set_length_scale_cells(cells)
    ps1.update_surrogate()
    assert np.array_equal(ps1.kernel_params, cells[150])


def test_large_call_column_groups_agree_with_the_ungrouped_kernel_and_the_oracle():
    """Calls of >= 32,768 candidates with >= 16 column blocks (N >= 2048) split a candidate tile's column blocks over 8
    workgroups (sigma_acq.hip); smaller calls do not.  Same candidates through both: dense values equal to the rounding of a
    different summation order, same selected point; oracle on a sub-sample; chunk-size invariance bit for bit."""
    N, M, d = 2048, 40960, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=1 << 14).factorise(X, y, ls)
    big = gp.score(Xs, dense=True)                                   # grouped
    h = M // 2
    a, b = gp.score(Xs[:h], dense=True), gp.score(Xs[h:], dense=True, idx_offset=h)   # 20,480 each: ungrouped
    sig2 = np.concatenate([a.sigma.cpu().numpy(), b.sigma.cpu().numpy()])
    mu2 = np.concatenate([a.mu.cpu().numpy(), b.mu.cpu().numpy()])
    assert np.array_equal(big.mu.cpu().numpy(), mu2)
    assert np.max(np.abs(big.sigma.cpu().numpy() - sig2)) <= 1e-12
    best2 = max([(a.best_val, -a.best_idx), (b.best_val, -b.best_idx)])
    assert abs(big.best_val - best2[0]) <= 1e-12 and (big.best_idx == -best2[1] or abs(a.best_val - b.best_val) <= 1e-12)
    sub = np.unique(np.concatenate([np.random.default_rng(5).choice(M, 200, replace=False),
                                    np.argsort(big.acq.cpu().numpy())[-8:]]))
    mu_o, sig_o = O.posterior_chol(X, y, Xs[sub], ls)
    assert np.max(np.abs(big.mu.cpu().numpy()[sub] - mu_o)) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(big.sigma.cpu().numpy()[sub] - sig_o)) <= 1e-8
    other = DeviceGP(chunk=1 << 13).factorise(X, y, ls).score(Xs, dense=True)
    assert np.array_equal(other.sigma.cpu().numpy(), big.sigma.cpu().numpy()) and other.best_idx == big.best_idx


def test_ard_coordinate_search_d8_n512_logdet_mode_picks_finite_cells():
    """d = 8, N = 512 (BASELINE config 2's surrogate): in the reference's likelihood most cells are -inf here (its determinant
    underflows) and the search returns the first such cell; PointSelector(likelihood="logdet") searches the same axes on
    finite fp64 values - the oracle's coordinate search with its Cholesky likelihood gives the same length scales, and the
    selected point is the oracle's for them."""
    N, d = 512, 8
    X, y, Xs, _ = make_problem(N, 4096, d)
    axes = [np.geomspace(0.1, 3.0, 12 + (k % 3)) for k in range(d)]
    ps = PointSelector(likelihood="logdet")
    ps.name, ps.iteration = "t", 0
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [len(Xs)], Xs
    ps.length_scales = axes
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()
    ls_o, grids_o = O.coordinate_search(X, y, axes, sweeps=2, nlml=O.nlml_cells_logdet)
    assert np.array_equal(ps.kernel_params, ls_o)
    for g, go in zip(ps.nlogml, grids_o):
        assert g.dtype == np.float64 and np.isfinite(g).all()
        np.testing.assert_allclose(g, go, rtol=1e-10)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls_o)
    assert idx[0] == _first_argmax(O.lcb(mu_o, sig_o, 4))
    # the reference's likelihood on the same axes: -inf cells, so its "search" cannot tell the axes' cells apart
    ref = PointSelector()
    ref.name, ref.iteration = "t", 0
    ref.measured_pts, ref.measured_vals = X, y
    ref.feature_domain, ref.predicted_pts = [len(Xs)], Xs
    ref.length_scales = axes
    ref.update_surrogate()
    assert any(np.isneginf(g).any() for g in ref.nlogml)


def test_dense_and_next_point_only_selectors_agree_beyond_the_rounding_of_the_order():
    """dense_outputs=False factorises the observations in farthest-point order (N > 128 here, so the order really differs from
    the arrival order), dense_outputs=True in arrival order: the same GP, roundings of a different order.  Both selectors
    return the same multi-index whenever the top-2 gap of the acquisition exceeds that rounding (1e-8), on a grid WITH exact
    ties (duplicated candidate rows: lowest index wins in both) and near-maxima; and the index is the oracle's."""
    N, d = 700, 3
    X, y, Xs, ls = make_problem(N, 40000, d)
    Xs = Xs.copy()
    Xs[20000:20050] = Xs[100:150]            # exact duplicates: exact ties wherever one of them is the maximum
    out = {}
    for dense in (True, False):
        ps = PointSelector(dense_outputs=dense)
        ps.name, ps.iteration = "t", 0
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [len(Xs)], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        out[dense] = {e: ps.lower_confidence_bound(e) for e in (0.5, 1.0, 4.0, 10.0)}
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    for e in (0.5, 1.0, 4.0, 10.0):
        acq = O.lcb(mu_o, sig_o, e)
        top = np.sort(acq)[::-1]
        first = _first_argmax(acq)
        gap = top[0] - top[top < top[0]][0] if (top < top[0]).any() else np.inf
        if gap > 1e-8:
            assert out[True][e][0] == first == out[False][e][0], (e, gap)
        else:   # a near-tie: each mode may pick either of the near-maxima, nothing else
            near = set(np.flatnonzero(acq >= top[0] - 1e-8).tolist())
            assert int(out[True][e][0]) in near and int(out[False][e][0]) in near
