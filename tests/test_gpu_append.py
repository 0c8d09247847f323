"""Appending observations to a factorised surrogate (gpbo_append_f64) against a fresh factorisation of the
extended data and against the CPU oracle: same posterior, same selected point."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bayesian_optimisation_amd import DeviceGP  # noqa: E402
from bayesian_optimisation_amd.synthetic import make_problem  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def _first_argmax(a):
    return int(np.flatnonzero(a == a.max())[0])


def _dense(gp, Xs):
    r = gp.score(Xs, dense=True)
    return r, r.mu.cpu().numpy(), r.sigma.cpu().numpy(), r.acq.cpu().numpy()


@pytest.mark.parametrize("N0,extra,d", [(1, 3, 1), (30, 5, 2), (126, 4, 3), (500, 20, 8), (1020, 8, 8)])
def test_append_matches_full_factorisation_and_oracle(N0, extra, d):
    """N0 -> N0+extra one row at a time, crossing the 128-row padding where the sizes say so."""
    X, y, Xs, ls = make_problem(N0 + extra, 3000, d)
    gp = DeviceGP(chunk=1024).factorise(X[:N0], y[:N0], ls)
    U0 = gp.U[:N0, :N0].cpu().numpy()
    for i in range(N0, N0 + extra):
        gp.append(X[i], y[i])
    assert gp.N == N0 + extra
    ref = DeviceGP(chunk=1024).factorise(X, y, ls)
    N = N0 + extra
    # the appended rows of K are the rows a fresh build produces, bit for bit
    assert np.array_equal(gp.K[:N, :N].cpu().numpy(), ref.K[:N, :N].cpu().numpy())
    Ua, Ur = gp.U[:N, :N].cpu().numpy(), ref.U[:N, :N].cpu().numpy()
    assert np.array_equal(Ua[:N0, :N0], U0) and not Ua[N0:, :N0].any()  # old columns are untouched
    scale = np.abs(Ur).max()
    assert np.max(np.abs(Ua - Ur)) <= 1e-9 * scale
    r, mu, sig, acq = _dense(gp, Xs)
    r2, mu2, sig2, acq2 = _dense(ref, Xs)
    ys = max(1.0, float(np.abs(y).max()))
    assert np.max(np.abs(mu - mu2)) <= 1e-10 * ys and np.max(np.abs(sig - sig2)) <= 1e-9
    mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
    assert np.max(np.abs(mu - mu_o)) <= 1e-10 * ys
    assert np.max(np.abs(sig - sig_o)) <= 1e-9
    acq_o = O.lcb(mu_o, sig_o, 4)
    top2 = np.sort(acq_o)[-2:]
    assert r.best_idx == _first_argmax(acq)
    if top2[1] - top2[0] > 1e-7 * ys:
        assert r.best_idx == _first_argmax(acq_o) == r2.best_idx


def test_append_does_not_write_to_the_callers_tensors():
    import torch

    X, y, Xs, ls = make_problem(41, 512, 4)
    Xd = torch.from_numpy(X[:40]).cuda()
    yd = torch.from_numpy(y[:40]).cuda()
    gp = DeviceGP(chunk=512).factorise(Xd, yd, ls)
    gp.append(X[40], y[40])
    assert np.array_equal(Xd.cpu().numpy(), X[:40]) and np.array_equal(yd.cpu().numpy(), y[:40])
    assert gp.X.data_ptr() != Xd.data_ptr()


def test_append_of_a_duplicate_point_is_still_positive_definite():
    """The jitter keeps an exact duplicate factorisable (lambda^2 ~ 2e-4), as it does in the reference."""
    X, y, Xs, ls = make_problem(64, 1024, 3)
    gp = DeviceGP(chunk=1024).factorise(X, y, ls)
    gp.append(X[10], y[10] + 0.5)
    Xe, ye = np.vstack([X, X[10:11]]), np.append(y, y[10] + 0.5)
    mu_o, sig_o = O.posterior_chol(Xe, ye, Xs, ls)
    r, mu, sig, acq = _dense(gp, Xs)
    assert np.max(np.abs(mu - mu_o)) <= 1e-8 * max(1.0, np.abs(ye).max())
    assert np.max(np.abs(sig - sig_o)) <= 1e-7


def test_append_reports_a_failed_pivot():
    """Without jitter a duplicate makes K singular: info = N+1, the error names factorise() as the way out."""
    X, y, Xs, ls = make_problem(32, 512, 2)
    gp = DeviceGP(chunk=512).factorise(X, y, ls, jitter1=1e-4, jitter2=0.0)
    before = gp.score(Xs, dense=True)
    mu0, sig0 = before.mu.cpu().numpy().copy(), before.sigma.cpu().numpy().copy()
    gp.jitter1 = -1e-3  # the appended diagonal falls below l.l
    with pytest.raises(np.linalg.LinAlgError):
        gp.append(X[3], y[3])
    assert gp.N == 32
    # a caller that catches the error and keeps scoring gets the surrogate of the 32 old observations, bit for bit
    after = gp.score(Xs, dense=True)
    assert after.best_idx == before.best_idx and after.best_val == before.best_val
    assert np.array_equal(after.mu.cpu().numpy(), mu0) and np.array_equal(after.sigma.cpu().numpy(), sig0)


def test_state_round_trip_then_append(tmp_path):
    """Persistence across jobs: save in one surrogate, load in another, append, same answer as refactorising."""
    X, y, Xs, ls = make_problem(201, 2048, 6)
    a = DeviceGP(chunk=1024).factorise(X[:200], y[:200], ls)
    path = str(tmp_path / "surrogate_state.npz")
    a.save_state(path)
    b = DeviceGP(chunk=1024).load_state(path)
    ra, rb = a.score(Xs, dense=True), b.score(Xs, dense=True)
    assert ra.best_idx == rb.best_idx and ra.best_val == rb.best_val
    assert np.array_equal(ra.sigma.cpu().numpy(), rb.sigma.cpu().numpy())
    b.append(X[200], y[200])
    ref = DeviceGP(chunk=1024).factorise(X, y, ls)
    r1, r2 = b.score(Xs, dense=True), ref.score(Xs, dense=True)
    assert np.max(np.abs(r1.mu.cpu().numpy() - r2.mu.cpu().numpy())) <= 1e-10 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(r1.sigma.cpu().numpy() - r2.sigma.cpu().numpy())) <= 1e-9
    assert r1.best_idx == r2.best_idx


def test_append_then_fp32_scoring_uses_the_new_factors():
    X, y, Xs, ls = make_problem(300, 2048, 8)
    gp = DeviceGP().factorise(X[:299], y[:299], ls)
    gp.prepare_f32()
    gp.append(X[299], y[299])
    r = gp.score_f32(Xs, dense=True)
    ref = DeviceGP().factorise(X, y, ls)
    r2 = ref.score_f32(Xs, dense=True)
    assert np.allclose(r.sigma.cpu().numpy(), r2.sigma.cpu().numpy(), atol=2e-3)
    assert np.allclose(r.mu.cpu().numpy(), r2.mu.cpu().numpy(), atol=2e-3 * max(1.0, np.abs(y).max()))


def test_point_selector_incremental_and_state_file(tmp_path):
    """The drop-in class with incremental=True: first call factorises, the next (one more row, same length
    scales) appends, a change of length scales falls back; a second process picks the factors up from a state
    file.  Every call selects what the stateless class selects."""
    from bayesian_optimisation_amd import PointSelector

    rng = np.random.default_rng(5)
    g1, g2 = np.linspace(0, 1, 40), np.linspace(0, 2, 30)
    Xs = np.stack(np.meshgrid(g1, g2, indexing="ij"), -1).reshape(-1, 2)
    X = rng.uniform(0, 1, (24, 2)) * [1.0, 2.0]
    y = np.sin(5 * X[:, 0]) + np.cos(3 * X[:, 1]) + 0.01 * rng.standard_normal(24)

    def run(ps, n, ls):
        ps.name, ps.iteration = "t", n
        ps.measured_pts, ps.measured_vals = X[:n].copy(), y[:n].copy()
        ps.feature_domain, ps.predicted_pts = [40, 30], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        return ps.lower_confidence_bound()

    path = str(tmp_path / "state.npz")
    inc = PointSelector(incremental=True, state_path=path)
    ls = np.array([0.3, 0.5])
    for n, want in [(20, "factorise"), (21, "append"), (23, "append")]:
        idx = run(inc, n, ls)
        assert inc.last_update == want
        plain = PointSelector()
        idx_p = run(plain, n, ls)
        assert np.array_equal(idx, idx_p)
        assert np.max(np.abs(inc.mean_func - plain.mean_func)) <= 1e-10 * max(1.0, np.abs(y).max())
        assert np.max(np.abs(inc.cov_func - plain.cov_func)) <= 1e-9
    # another job: new object, same state file
    other = PointSelector(state_path=path)
    idx = run(other, 24, ls)
    assert other.last_update == "append"
    plain = PointSelector()
    assert np.array_equal(idx, run(plain, 24, ls))
    # new length scales: the held factors are useless
    run(other, 24, np.array([0.31, 0.5]))
    assert other.last_update == "factorise"
    # an edited earlier observation: likewise
    y[3] += 1.0
    run(other, 24, np.array([0.31, 0.5]))
    assert other.last_update == "factorise"


def test_point_selector_refreshes_after_many_appended_columns(monkeypatch):
    """Appended columns go through the explicit inverse factor; after MAX_APPENDED_COLUMNS of them the class takes
    the full factorisation again."""
    from bayesian_optimisation_amd import PointSelector
    from bayesian_optimisation_amd import point_selector as PS

    monkeypatch.setattr(PS, "MAX_APPENDED_COLUMNS", 3)
    X, y, Xs, ls = make_problem(16, 400, 2)
    ps = PointSelector(incremental=True)
    seen = []
    for n in range(10, 17):
        ps.measured_pts, ps.measured_vals = X[:n].copy(), y[:n].copy()
        ps.feature_domain, ps.predicted_pts = [20, 20], Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        seen.append(ps.last_update)
    assert seen == ["factorise", "append", "append", "append", "factorise", "append", "append"]


def test_next_point_only_selector_keeps_its_ordered_factorisation_across_jobs(tmp_path):
    """PointSelector(dense_outputs=False, state_path=...): the first job factorises the observations in farthest-point order
    and saves that state (order included); the next jobs - new objects, a few more rows in the CALLER's order - recognise the
    old rows through the permutation, append, and select what a stateless full-evaluation object selects; a dense job that
    hits the N == M quirk with such a state refactorises in arrival order instead of appending."""
    from bayesian_optimisation_amd import PointSelector

    rng = np.random.default_rng(15)
    d, n0 = 3, 1100
    X = rng.uniform(0, 1, (n0 + 3, d))
    X[:n0] = X[:n0][np.argsort(X[:n0, 0])]                       # a sorted history
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + X[:, 2] + 0.01 * rng.standard_normal(n0 + 3)
    g = 34
    axes = [np.linspace(0, 1, g)] * d
    Xs = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, d)   # 39,304 candidates
    ls = np.array([0.2, 0.3, 0.5])
    path = str(tmp_path / "ordered_state.npz")

    def run(ps, n):
        ps.name, ps.iteration = "t", n
        ps.measured_pts, ps.measured_vals = X[:n].copy(), y[:n].copy()
        ps.feature_domain, ps.predicted_pts = [g] * d, Xs
        ps.set_kernel_params(ls)
        ps.update_surrogate()
        return ps.lower_confidence_bound(), ps.expected_improvement()

    for n, want in [(n0, "factorise"), (n0 + 1, "append"), (n0 + 3, "append")]:
        job = PointSelector(dense_outputs=False, state_path=path)   # a fresh object per job, as in the DAG
        i4, ie = run(job, n)
        assert job.last_update == want and job._gp.order == "fps" and job._gp.last_screen["mode"] == "bound"
        plain = PointSelector()
        p4, pe = run(plain, n)
        assert np.array_equal(i4, p4) and np.array_equal(ie, pe)
        Xa, ya = job._gp.observations_host()
        assert np.array_equal(Xa, X[:n]) and np.array_equal(ya, y[:n])
    st = dict(np.load(path))
    assert "perm" in st and len(st["perm"]) == n0 + 3 and st["perm"][-1] == n0 + 2
    # a dense job whose candidates have the observations' shape (the N == M quirk): the ordered state cannot serve it
    quirk = PointSelector(state_path=path)
    Xq = rng.uniform(0, 1, (n0 + 4, d))
    quirk.name, quirk.iteration = "t", 0
    quirk.measured_pts = np.concatenate([X[:n0 + 3], rng.uniform(0, 1, (1, d))])
    quirk.measured_vals = np.concatenate([y[:n0 + 3], [0.3]])
    quirk.feature_domain, quirk.predicted_pts = [n0 + 4], Xq
    quirk.set_kernel_params(ls)
    quirk.update_surrogate()
    assert quirk.last_update == "factorise" and quirk._gp.order == "arrival"
