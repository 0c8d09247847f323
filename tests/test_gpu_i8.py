"""int8-sliced variance screen (csrc/ozaki.hip): the N^2-per-candidate product from 20 exact int8 slice products on the
integer matrix cores, fp64 means, fp64 decision behind it (csrc/rescore.hip).

Tolerances (written here): |dmu| = 0 against the fp64 kernels (same arithmetic); |dsigma| <= 2e-9 against the oracle
(tools/ozaki_error.py: 1.3e-10 at N = 4096 from the two 2^-47 roundings and the dropped digit pairs); the selected point
is the fp64 kernels' (index equal, value within the rounding of their column-split launch)."""
import numpy as np
import pytest

from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


@pytest.mark.parametrize("N,M,d,chunk", [(100, 1000, 3, 512), (256, 2048, 8, 1024), (129, 3000, 1, 512),
                                         (700, 5000, 8, 2048), (2048, 4096, 8, 4096), (1000, 6000, 16, 1024)])
def test_i8_screen_vs_oracle_and_fp64_kernels(N, M, d, chunk):
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score_i8(Xs, dense=True, idx_offset=5)
    r64 = gp.score(Xs, dense=True, idx_offset=5)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    acq_o = O.lcb(mu_o, sig_o, 4)
    scale = max(1.0, float(np.abs(y).max()))
    assert np.array_equal(r.mu.cpu().numpy(), r64.mu.cpu().numpy())
    assert np.max(np.abs(r.sigma.cpu().numpy() - sig_o)) <= 2e-9
    assert np.max(np.abs(r.sigma.cpu().numpy() - r64.sigma.cpu().numpy())) <= 2e-9
    assert np.max(np.abs(r.acq.cpu().numpy() - acq_o)) <= 1e-8 * scale
    assert r.nan_count == 0
    assert r.best_idx == r64.best_idx and abs(r.best_val - r64.best_val) <= 1e-12 * scale
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx == 5 + _first_argmax(acq_o)
    st = gp.last_screen
    assert st["mode"] == "i8" and not st["fallback"] and st["rounds"] == 1 and st["tau"] == 1e-9 and st["err_max"] < 2.5e-10
    f_best = float(y.min())
    e8 = gp.score_i8(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    e64 = gp.score(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    assert e8.best_idx == e64.best_idx and abs(e8.best_val - e64.best_val) <= 1e-12 * scale


def test_i8_chunk_invariance_bit_for_bit_and_ties(golden):
    """Everything after the two fixed-point roundings is exact integer arithmetic: the dense values cannot depend on how
    the candidates are cut into chunks.  2,500 exact ties -> index 0 (point_selector.py:207)."""
    X, y, Xs, ls = make_problem(300, 6000, 8)
    a = DeviceGP(chunk=1024).factorise(X, y, ls).score_i8(Xs, dense=True)
    b = DeviceGP(chunk=4096).factorise(X, y, ls).score_i8(Xs, dense=True)
    assert np.array_equal(a.sigma.cpu().numpy(), b.sigma.cpu().numpy()) and a.best_idx == b.best_idx
    g = golden("g4_tie_tiny_ls")
    r = DeviceGP(chunk=1024).factorise(g["X"], g["y"], g["kernel_params"]).score_i8(g["Xs"], dense=True)
    assert r.best_idx == 0 and len(np.unique(r.acq.cpu().numpy())) == 1


@pytest.mark.parametrize("name", ["g5_d8_n512_m4096", "g5_d8_n2048_m4096", "g6_d16_n256_m2048"])
def test_i8_screen_vs_reference_golden(golden, name):
    """Against vectors produced by the reference itself (tests/golden/make_golden.py): SURVEY.md's fp64 tolerances."""
    g = golden(name)
    X, y, Xs, ls = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
    gp = DeviceGP(chunk=2048).factorise(X, y, g["kernel_params"])
    r = gp.score_i8(Xs, dense=True)
    assert np.max(np.abs(r.mu.cpu().numpy() - g["mean_func"])) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r.sigma.cpu().numpy() - g["cov_func"])) <= 1e-8
    assert np.max(np.abs(r.acq.cpu().numpy() - g["acq_func_eval"])) <= 1e-8 * max(1.0, np.abs(y).max())
    if g["top2_gap"] > 1e-7 * max(1.0, np.abs(y).max()):
        assert r.best_idx == _first_argmax(g["acq_func_eval"])


def test_i8_nan_candidate_and_size_limit():
    X, y, Xs, ls = make_problem(64, 3000, 4)
    Xs = Xs.copy()
    Xs[77, 0] = np.nan
    r = DeviceGP(chunk=1024).factorise(X, y, ls).score_i8(Xs)
    assert r.nan_count == 1 and r.best_idx != 77


@pytest.mark.parametrize("name", ["g1_m32", "g1_m50", "g2_n20_tr", "g4_dup_rows", "g7_n_eq_m"])
def test_dropin_class_with_the_int8_screen_reproduces_the_reference(golden, name):
    """The reference's own runs (ARD search included where the fixture has one) through PointSelector(precision="i8"):
    same length scales, same selected point, dense outputs within the fp64 tolerances of SURVEY.md 8(a)."""
    from bayesian_optimisation_amd import PointSelector

    g = golden(name)
    ps = PointSelector(precision="i8")
    ps.name, ps.iteration = "T", 0
    ps.measured_pts, ps.measured_vals = g["X"], g["y"]
    ps.feature_domain, ps.predicted_pts = [int(v) for v in g["feature_domain"]], g["Xs"]
    if "length_scales" in g:
        ps.length_scales = g["length_scales"]
    else:
        ps.set_kernel_params(g["kernel_params"] if "kernel_params" in g else g["ls"])
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()
    ys = max(1.0, float(np.abs(g["y"]).max()))
    assert np.max(np.abs(ps.mean_func - g["mean_func"])) <= 1e-9 * ys
    assert np.max(np.abs(ps.cov_func - g["cov_func"])) <= 1e-8
    assert np.max(np.abs(ps.acq_func_eval - g["acq_func_eval"])) <= 1e-8 * ys
    if g["top2_gap"] > 1e-7 * ys or g["n_max_ties"] > 1:
        assert np.array_equal(idx, g["index"])


def test_screens_with_a_negative_explore_weight():
    """lower_confidence_bound(explore < 0) makes the acquisition DEcrease with sigma: the screen's interval must still
    bracket the fp64 value (bounds are ordered, not assumed)."""
    X, y, Xs, ls = make_problem(300, 40000, 4)
    gp = DeviceGP(chunk=8192).factorise(X, y, ls)
    r64 = gp.score(Xs, acquisition="lcb", explore=-2.5)
    for route in ("score_i8", "score_f32"):
        r = getattr(gp, route)(Xs, acquisition="lcb", explore=-2.5)
        assert r.best_idx == r64.best_idx and abs(r.best_val - r64.best_val) <= 1e-12 * max(1.0, abs(r64.best_val))
        assert not gp.last_screen["fallback"]
