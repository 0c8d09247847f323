"""Coarse int8 screen (csrc/ozaki.hip, gpbo_posterior_acq_i8c): the three leading balanced base-256 digits of K* and of the
column-scaled U, the six slice products with a + b <= 2, exact int32 accumulation, 256 x 128 tiles - the cheapest pass that
still leaves only a handful of candidates for the fp64 kernels (csrc/rescore.hip), which decide.

Tolerances (written here): the mean is the fp64 kernels' bit for bit; |var64 - var| <= 1e-3 (measured 1.5e-5 at N = 256,
2.1e-4 at N = 4096: tools/ozaki_error.py "sk=3 su=3 keep=3"), checked on every call against the re-scored rows with a
factor 4 to spare (tau is raised and the selection repeated otherwise); the selected point is the fp64 kernels' (index equal,
value within the rounding of their column-split launch) and the oracle's first arg-max."""
import numpy as np
import pytest

from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


@pytest.mark.parametrize("N,M,d,chunk", [(100, 1000, 3, 512), (256, 2048, 8, 1024), (129, 3001, 1, 512),
                                         (700, 5000, 8, 2048), (2048, 4096, 8, 4096), (1000, 6000, 16, 1024),
                                         (2100, 33000, 8, 1 << 14)])
def test_coarse_screen_vs_oracle_and_fp64_kernels(N, M, d, chunk):
    """Np / 128 < 16: one workgroup per 256-row tile; N >= 1921: eight column groups per tile on one XCD."""
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    r = gp.score_i8c(Xs, dense=True, idx_offset=5)
    st = dict(gp.last_screen)
    r64 = gp.score(Xs, dense=True, idx_offset=5)
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    acq_o = O.lcb(mu_o, sig_o, 4)
    scale = max(1.0, float(np.abs(y).max()))
    assert np.array_equal(r.mu.cpu().numpy(), r64.mu.cpu().numpy())
    assert np.max(np.abs(r.sigma.cpu().numpy() ** 2 - sig_o ** 2)) <= 1e-3
    assert r.nan_count == 0
    assert r.best_idx == r64.best_idx and abs(r.best_val - r64.best_val) <= 1e-12 * scale
    top2 = np.sort(acq_o)[-2:]
    if top2[1] - top2[0] > 1e-7:
        assert r.best_idx == 5 + _first_argmax(acq_o)
    assert st["mode"] == "i8c" and not st["fallback"] and 4.0 * st["err_max"] <= st["tau"]
    if M >= 30000:   # (the survivor count includes the ~1,024 sampled rows that check tau)
        assert st["survivors"] < 2000
    f_best = float(y.min())
    e8 = gp.score_i8c(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    e64 = gp.score(Xs, acquisition="ei", f_best=f_best, xi=0.0)
    assert e8.best_idx == e64.best_idx and abs(e8.best_val - e64.best_val) <= 1e-12 * scale


def test_coarse_screen_chunk_invariance_bit_for_bit_and_ties(golden):
    """Integer arithmetic after the one rounding of each operand: the dense values cannot depend on how the candidates
    are cut into chunks or tiles.  2,500 exact ties -> index 0 (point_selector.py:207)."""
    X, y, Xs, ls = make_problem(300, 6000, 8)
    a = DeviceGP(chunk=1024).factorise(X, y, ls).score_i8c(Xs, dense=True)
    b = DeviceGP(chunk=4096).factorise(X, y, ls).score_i8c(Xs, dense=True)
    assert np.array_equal(a.sigma.cpu().numpy(), b.sigma.cpu().numpy()) and a.best_idx == b.best_idx
    g = golden("g4_tie_tiny_ls")
    r = DeviceGP(chunk=1024).factorise(g["X"], g["y"], g["kernel_params"]).score_i8c(g["Xs"], dense=True)
    assert r.best_idx == 0


def test_coarse_screen_is_the_leading_digits_of_the_full_one():
    """Same U fragments (slices 0-2 of the six), K* digits cut from the same fixed-point value: the full pass agrees with
    the fp64 kernels to 1e-9, the coarse one differs from both by the dropped digits only (1e-7 .. 1e-3)."""
    X, y, Xs, ls = make_problem(1500, 8192, 8)
    gp = DeviceGP(chunk=4096).factorise(X, y, ls)
    c = gp.score_i8c(Xs, dense=True).sigma.cpu().numpy() ** 2
    f = gp.score_i8(Xs, dense=True).sigma.cpu().numpy() ** 2
    d64 = gp.score(Xs, dense=True).sigma.cpu().numpy() ** 2
    assert np.max(np.abs(f - d64)) < 1e-9 and 1e-7 < np.max(np.abs(c - d64)) < 1e-3


@pytest.mark.parametrize("name", ["g5_d8_n512_m4096", "g5_d8_n2048_m4096", "g6_d16_n256_m2048"])
def test_coarse_screen_selects_the_reference_point(golden, name):
    """Vectors produced by the reference itself (tests/golden/make_golden.py): mean within SURVEY.md's fp64 tolerance, the
    variance within the screen's, the selected point the reference's."""
    g = golden(name)
    X, y, Xs, ls = make_problem(int(g["N"]), int(g["M"]), int(g["d"]))
    gp = DeviceGP(chunk=2048).factorise(X, y, g["kernel_params"])
    r = gp.score_i8c(Xs, dense=True)
    assert np.max(np.abs(r.mu.cpu().numpy() - g["mean_func"])) <= 1e-9 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r.sigma.cpu().numpy() ** 2 - g["cov_func"] ** 2)) <= 1e-3
    if g["top2_gap"] > 1e-7 * max(1.0, np.abs(y).max()):
        assert r.best_idx == _first_argmax(g["acq_func_eval"])


def test_coarse_screen_nan_candidate_negative_explore_and_config4_shape():
    X, y, Xs, ls = make_problem(64, 3000, 4)
    Xs = Xs.copy()
    Xs[77, 0] = np.nan
    r = DeviceGP(chunk=1024).factorise(X, y, ls).score_i8c(Xs)
    assert r.nan_count == 1 and r.best_idx != 77
    X, y, Xs, ls = make_problem(300, 40000, 4)
    gp = DeviceGP(chunk=8192).factorise(X, y, ls)
    r64 = gp.score(Xs, acquisition="lcb", explore=-2.5)
    r = gp.score_i8c(Xs, acquisition="lcb", explore=-2.5)
    assert r.best_idx == r64.best_idx and not gp.last_screen["fallback"]
    # BASELINE configs[3]'s shape (d=16, N=8192), 2^16 candidates: the error grows with N; whatever tau the call settles on,
    # the decision is the fp64 kernels'
    X, y, Xs, ls = make_problem(8192, 1 << 16, 16)
    gp = DeviceGP().factorise(X, y, ls)
    r64 = gp.score(Xs)
    r = gp.score_i8c(Xs)
    st = gp.last_screen
    assert r.best_idx == r64.best_idx and not st["fallback"] and st["rounds"] <= 2 and 4.0 * st["err_max"] <= st["tau"]
