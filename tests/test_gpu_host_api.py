"""The host-pointer entry points (gpbo_select_next_host_f64, gpbo_nlml_grid_host_f64) and the NumPy + ctypes
class built on them: same numbers as the tensor-resident path, same answers as the reference's golden vectors,
and no PyTorch in the process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (this process mixes both routes: PyTorch first, see test_mixing_...)

from bayesian_optimisation_amd import DeviceGP, PointSelector  # noqa: E402
from bayesian_optimisation_amd import host_binding as H  # noqa: E402
from bayesian_optimisation_amd.synthetic import make_problem  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_MU, TOL_SIG, TOL_ACQ = 1e-9, 1e-8, 1e-8  # as tests/test_gpu_parity.py


def _run_host(g, preset=False):
    ps = H.PointSelectorHost()
    ps.name, ps.iteration = "T", 0
    ps.measured_pts, ps.measured_vals = g["X"], g["y"]
    ps.feature_domain = [int(v) for v in g["feature_domain"]]
    ps.predicted_pts = g["Xs"]
    if preset:
        ps.set_kernel_params(g["kernel_params"] if "kernel_params" in g else g["ls"])
    else:
        ps.length_scales = g["length_scales"]
    ps.update_surrogate()
    return ps, (ps.lower_confidence_bound(float(g["explore"])) if "explore" in g else ps.lower_confidence_bound())


@pytest.mark.parametrize("N,M,d", [(1, 50, 1), (37, 2500, 2), (300, 5000, 8)])
def test_host_call_is_bitwise_the_device_pointer_path(N, M, d):
    X, y, Xs, ls = make_problem(N, M, d)
    r = H.select_next(X, y, ls, Xs, want_cov_meas=True, chunk=1024)
    gp = DeviceGP(chunk=1024).factorise(X, y, ls)
    q = gp.score(Xs, dense=True)
    assert r["info"] == 0 and r["nan_count"] == 0
    assert np.array_equal(r["mu"], q.mu.cpu().numpy()) and np.array_equal(r["sigma"], q.sigma.cpu().numpy())
    assert np.array_equal(r["acq"], q.acq.cpu().numpy())
    assert (r["best_idx"], r["best_val"]) == (q.best_idx, q.best_val)
    assert np.array_equal(r["cov_meas"], gp.cov_meas_host())
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(r["mu"] - mu_o)) <= 1e-10 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r["sigma"] - sig_o)) <= 1e-9
    # expected improvement through the same entry point
    e = H.select_next(X, y, ls, Xs, acquisition="ei", f_best=float(y.min()), xi=0.01, chunk=1024)
    qe = gp.score(Xs, acquisition="ei", f_best=float(y.min()), xi=0.01, dense=True)
    assert np.array_equal(e["acq"], qe.acq.cpu().numpy()) and e["best_idx"] == qe.best_idx


def test_host_call_without_dense_outputs_takes_the_exact_prefix_bound_and_returns_the_same_point():
    """A plain-C caller that asks for the next point only (mu / sigma / acq pointers NULL): from 32,768 candidates and
    N >= 897 the entry point runs the branch and bound of DESIGN 4d - same index, same NaN count, value equal up to the
    rounding of the column-split re-scoring launch."""
    X, y, Xs, ls = make_problem(1300, 50000, 6)
    Xs = Xs.copy()
    Xs[41, 3] = np.nan
    for kw in (dict(acquisition="lcb", explore=4.0), dict(acquisition="ei", f_best=float(y.min()), xi=0.0)):
        full = H.select_next(X, y, ls, Xs, dense=True, **kw)
        only = H.select_next(X, y, ls, Xs, dense=False, **kw)
        assert only["mu"] is None and only["info"] == 0
        assert only["best_idx"] == full["best_idx"] and only["nan_count"] == full["nan_count"] == 1
        assert abs(only["best_val"] - full["best_val"]) <= 1e-12 * max(1.0, abs(full["best_val"]))
    # a negative weight cannot use the bound: the plain pass answers, bit for bit
    full = H.select_next(X, y, ls, Xs, dense=True, explore=-1.5)
    only = H.select_next(X, y, ls, Xs, dense=False, explore=-1.5)
    assert (only["best_idx"], only["best_val"]) == (full["best_idx"], full["best_val"])


def test_failed_factorisation_is_reported_by_the_callers_row_in_either_order():
    """An observation with a NaN coordinate: the covariance matrix has a NaN pivot in that row, nothing is scored
    (best_idx = -1) and info names the row - the CALLER's row, also when the call had put the observations in
    farthest-point order for the exact bound (no dense outputs)."""
    X, y, Xs, ls = make_problem(1300, 40000, 5)
    X = X.copy()
    X[777, 2] = np.nan
    for dense in (True, False):
        r = H.select_next(X, y, ls, Xs, dense=dense)
        assert r["info"] == 778 and r["best_idx"] == -1, (dense, r["info"])


@pytest.mark.parametrize("name", ["g1_m32", "g1_m50", "g4_ard_n2", "g2_n1_tr", "g2_n5_a", "g2_n20_tr", "g3_n1_2d",
                                  "g10_2d_0", "g10_2d_1", "g10_2d_3", "g10_2d_4", "g10_1d_0", "g10_1d_2"])
def test_host_class_full_path_with_ard_vs_reference_golden(golden, name):
    g = golden(name)
    ps, idx = _run_host(g)
    assert np.array_equal(np.asarray(ps.kernel_params), g["kernel_params"])
    assert ps.kernel_params.shape == g["kernel_params"].shape
    assert idx.dtype == np.int64 and idx.shape == g["index"].shape
    ys = max(1.0, float(np.max(np.abs(g["y"]))))
    assert np.max(np.abs(ps.mean_func - g["mean_func"])) <= TOL_MU * ys
    assert np.max(np.abs(ps.cov_func - g["cov_func"])) <= TOL_SIG
    assert np.max(np.abs(ps.acq_func_eval - g["acq_func_eval"])) <= TOL_ACQ * ys
    if g["top2_gap"] > 1e-7 * ys or g["n_max_ties"] > 1:
        assert np.array_equal(idx, g["index"])
    assert isinstance(ps.measured_pts, list) and isinstance(ps.measured_vals, list)
    assert ps.cov_meas.shape == (len(g["X"]), len(g["X"]))
    if "nlogml" in g:
        np.testing.assert_allclose(ps.nlogml, g["nlogml"], rtol=2e-6)


def test_host_class_errors_are_the_references(golden):
    with pytest.raises(IndexError):          # NaN in the acquisition: point_selector.py:207
        _run_host(golden("g8_nan"), preset=True)
    g = golden("g7_n_eq_m")                  # shape-coincidence quirk goes through diag_add
    ps, idx = _run_host(g, preset=True)
    assert np.max(np.abs(ps.cov_func - g["cov_func"])) <= TOL_SIG and np.array_equal(idx, g["index"])
    ps = H.PointSelectorHost()
    ps.measured_pts = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 1.0]])
    ps.measured_vals = np.ones(3)
    ps.feature_domain, ps.predicted_pts = [4], np.zeros((4, 2))
    ps.set_kernel_params([1.0, 1.0])
    ps.update_surrogate()                     # duplicates are fine with the reference's jitter
    r = H.select_next(ps._inputs[0], ps._inputs[1], [1.0, 1.0], np.zeros((4, 2)))
    assert r["info"] == 0


def test_second_acquisition_and_ei_on_the_host_class():
    X, y, Xs, ls = make_problem(40, 900, 2)
    ps = H.PointSelectorHost()
    ps.measured_pts, ps.measured_vals = X, y
    ps.feature_domain, ps.predicted_pts = [30, 30], Xs
    ps.set_kernel_params(ls)
    ps.update_surrogate()
    i4 = ps.lower_confidence_bound()
    a4 = ps.acq_func_eval.copy()
    i1 = ps.lower_confidence_bound(explore=1)
    assert np.array_equal(ps.acq_func_eval, 1.0 * ps.cov_func - ps.mean_func)
    assert np.array_equal(a4, 4.0 * ps.cov_func - ps.mean_func)
    assert np.array_equal(i4, np.unravel_index(int(np.flatnonzero(a4.ravel() == a4.max())[0]), (30, 30)))
    assert i1.shape == (2,)
    ie = ps.expected_improvement(xi=0.01)
    ei_o = O.expected_improvement(ps.mean_func.ravel(), ps.cov_func.ravel(), float(np.min(y)), 0.01)
    np.testing.assert_allclose(ps.acq_func_eval.ravel(), ei_o, rtol=1e-12, atol=1e-15)
    assert np.array_equal(ie, np.unravel_index(int(np.flatnonzero(ei_o == ei_o.max())[0]), (30, 30)))


def test_likelihood_grid_beyond_the_lds_kernel():
    X, y, _, ls = make_problem(200, 16, 2)
    cells = np.stack(np.meshgrid([0.2, 0.4, 0.8], [0.3, 0.9], indexing="ij"), -1).reshape(-1, 2)
    a = H.nlml_grid(X, y, cells)
    b = DeviceGP().nlml_grid(X, y, cells)
    assert a.dtype == np.float32 and np.array_equal(a, b, equal_nan=True)
    ref = O.nlml_grid(X, y, [np.array([0.2, 0.4, 0.8]), np.array([0.3, 0.9])]).ravel()
    fin = np.isfinite(ref)
    np.testing.assert_allclose(a[fin], ref[fin], rtol=2e-6)
    assert np.array_equal(np.isfinite(a), fin)


def test_host_binding_runs_without_pytorch_in_the_process():
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "from bayesian_optimisation_amd.host_binding import PointSelectorHost\n"
        "g = dict(np.load(sys.argv[1]))\n"
        "ps = PointSelectorHost()\n"
        "ps.name, ps.iteration = 'T', 0\n"
        "ps.measured_pts, ps.measured_vals = g['X'], g['y']\n"
        "ps.feature_domain = [int(v) for v in g['feature_domain']]\n"
        "ps.predicted_pts, ps.length_scales = g['Xs'], g['length_scales']\n"
        "ps.update_surrogate()\n"
        "idx = ps.lower_confidence_bound()\n"
        "assert 'torch' not in sys.modules, 'PyTorch was imported'\n"
        "assert np.array_equal(idx, g['index']), (idx, g['index'])\n"
        "print('ok', idx.tolist())\n"
    )
    fixture = os.path.join(REPO, "tests", "golden", "g1_m32.npz")
    out = subprocess.run([sys.executable, "-c", code, fixture], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith("ok")


def test_mixing_host_binding_and_pytorch_in_one_process():
    """libgpbo.so binds to the HIP runtime PyTorch ships (two runtimes in one process: the second one sees no GPU).
    PyTorch first, then the host-pointer route, then the tensor-resident route: fine, same numbers.  The host-pointer
    route first and the tensor-resident route afterwards is refused with a clear error: bringing PyTorch's GPU
    context up after the library has initialised HIP was seen to dead-lock now and then."""
    common = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "rng = np.random.default_rng(0)\n"
        "X, y, Xs = rng.uniform(0, 1, (10, 2)), rng.standard_normal(10), rng.uniform(0, 1, (100, 2))\n"
    )
    good = common + (
        "import torch\n"
        "from bayesian_optimisation_amd import host_binding as H, DeviceGP\n"
        "r = H.select_next(X, y, [0.3, 0.3], Xs)\n"
        "assert torch.cuda.is_available(), 'PyTorch lost the GPU'\n"
        "q = DeviceGP(chunk=512).factorise(X, y, [0.3, 0.3]).score(Xs, dense=True)\n"
        "assert q.best_idx == r['best_idx'] and np.array_equal(q.mu.cpu().numpy(), r['mu'])\n"
        "print('ok')\n"
    )
    out = subprocess.run([sys.executable, "-c", good], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("ok")
    refused = common + (
        "from bayesian_optimisation_amd import host_binding as H, DeviceGP\n"
        "from bayesian_optimisation_amd._lib import GpboError\n"
        "r = H.select_next(X, y, [0.3, 0.3], Xs)\n"
        "assert 'torch' not in sys.modules\n"
        "try:\n"
        "    DeviceGP()\n"
        "except GpboError as e:\n"
        "    assert 'import torch' in str(e)\n"
        "    print('refused')\n"
    )
    out = subprocess.run([sys.executable, "-c", refused], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("refused")


def test_host_binding_q_expected_improvement_matches_the_tensor_resident_class():
    X, y, Xs, ls = make_problem(30, 1600, 2)
    out = []
    for cls in (H.PointSelectorHost, PointSelector):
        ps = cls()
        ps.measured_pts, ps.measured_vals = X, y
        ps.feature_domain, ps.predicted_pts = [40, 40], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        out.append((ps.q_expected_improvement(n_samples=256, seed=7), ps.acq_func_eval.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    ref = O.qei_mc(X, y, Xs, ls, O.qei_base_samples(256, 8, 7), float(np.min(y)))
    np.testing.assert_allclose(out[0][1], ref, rtol=0, atol=1e-9)
