"""The SELECT_PARAMETERS driver against the reference's OWN run of its script.

tests/golden/driver_*.npz were produced by executing /root/reference/select_parameters.py itself (under runpy, site
paths remapped, a recording stand-in for plot_utils: tests/golden/make_driver_golden.py).  Every step of every sequence is
replayed here through bayesian_optimisation_amd.select_parameters on the recorded input state and must leave the same
opto_log.JSON (parsed: same keys, same values), the same measured_points/*.npy (array_equal, float64) and make the same
plot calls (function, plot name, iteration, array shapes, rows shown; arrays within the fp64 tolerances).
CPU: the GP step comes from the oracle-backed selector (file formats and branch logic need no GPU).
GPU (-m gpu): the same replay through the real PointSelector, fp64 and int8-screened."""
import json
import os
import sys
import types

import numpy as np
import pytest

from bayesian_optimisation_amd import select_parameters as SP
from oracle import gp_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEQUENCES = ["pair01", "pair23", "amp56", "amp78", "rise"]


class OracleSelector:
    """Attribute-protocol stand-in backed by the CPU oracle (tests only)."""

    def update_surrogate(self):
        self._out = O.select_next(np.array(self.measured_pts), np.array(self.measured_vals), self.predicted_pts,
                                  self.feature_domain, length_scales=self.length_scales)
        self.kernel_params = self._out["kernel_params"]
        self.mean_func, self.cov_func = self._out["mean_func"], self._out["cov_func"]

    def lower_confidence_bound(self, explore=4):
        self.acq_func_eval = self._out["acq_func_eval"]
        return self._out["index"]


class _FixedDraw:
    """Stands in for the Generator of the first-ever step: returns the indices the reference's np.random.randint drew."""

    def __init__(self, idx):
        self.idx = np.asarray(idx)

    def integers(self, lo, hi, size):
        assert (lo, hi, size) == (0, 50, 2)
        return self.idx


@pytest.fixture
def recording_plot_utils():
    calls = []
    m = types.ModuleType("plot_utils")

    def rec(fn):
        def f(*a):
            calls.append((fn, a))
        return f

    for fn in ("surrogate_uncert_acquistion", "surrogate_uncert_acquistion_1d", "plot_ARD_LL", "plot_ARD_LL_1d"):
        setattr(m, fn, rec(fn))
    old = sys.modules.get("plot_utils")
    sys.modules["plot_utils"] = m
    yield calls
    if old is None:
        sys.modules.pop("plot_utils", None)
    else:
        sys.modules["plot_utils"] = old


def _restore(base, z, k):
    os.makedirs(os.path.join(base, "measured_points"), exist_ok=True)
    for f in os.listdir(os.path.join(base, "measured_points")):
        os.unlink(os.path.join(base, "measured_points", f))
    with open(os.path.join(base, "opto_log.JSON"), "w") as f:
        f.write(str(z[f"step{k}:before:json"]))
    for key in z.files:
        if key.startswith(f"step{k}:before:npy:"):
            np.save(os.path.join(base, "measured_points", key.split("npy:", 1)[1]), z[key])


def replay(seq, base, factory, calls, mu_tol, sig_tol):
    z = np.load(os.path.join(GOLDEN, f"driver_{seq}.npz"))
    for k in range(int(z["n_steps"])):
        _restore(base, z, k)
        del calls[:]
        rng = _FixedDraw(z["first_random_index"]) if "first_random_index" in z.files else None
        SP.select_parameters(base, selector_factory=factory, rng=rng)
        # state files
        got = json.load(open(os.path.join(base, "opto_log.JSON")))
        want = json.loads(str(z[f"step{k}:after:json"]))
        assert got == want, f"{seq} step {k}: opto_log.JSON differs"
        raw = open(os.path.join(base, "opto_log.JSON")).read()
        assert raw == json.dumps(want, indent=4)                       # json.dump(indent=4) layout (select_parameters.py:207)
        names = [key.split("npy:", 1)[1] for key in z.files if key.startswith(f"step{k}:after:npy:")]
        assert sorted(os.listdir(os.path.join(base, "measured_points"))) == sorted(names)
        for n in names:
            a = np.load(os.path.join(base, "measured_points", n))
            b = z[f"step{k}:after:npy:{n}"]
            assert a.dtype == b.dtype == np.float64 and np.array_equal(a, b), f"{seq} step {k}: {n} differs"
        # plot calls made by the driver itself (the ARD plots belong to the selector class)
        want_plots = [(i, p) for i, p in enumerate(json.loads(str(z[f"step{k}:plots"]))) if p["fn"].startswith("surrogate_")]
        got_plots = [c for c in calls if c[0].startswith("surrogate_")]
        assert [c[0] for c in got_plots] == [p["fn"] for _, p in want_plots]
        for (fn, a), (i, p) in zip(got_plots, want_plots):
            name, iteration, rows = a[-3], a[-2], a[-1]
            assert (name, iteration, len(rows)) == (p["name"], p["iteration"], p["n_measured"])
            assert [list(np.shape(v)) for v in a[:-3]] == p["shapes"]
            ys = max(1.0, float(np.abs(np.asarray(rows)[:, -1]).max()))
            assert np.max(np.abs(a[0] - z[f"step{k}:plot{i}:mu"])) <= mu_tol * ys
            assert np.max(np.abs(a[1] - z[f"step{k}:plot{i}:cov"])) <= sig_tol
            assert np.max(np.abs(a[2] - z[f"step{k}:plot{i}:acq"])) <= 4 * sig_tol + mu_tol * ys


@pytest.mark.parametrize("seq", SEQUENCES)
def test_driver_reproduces_the_reference_script_files_and_plot_calls(seq, tmp_path, recording_plot_utils):
    replay(seq, str(tmp_path), OracleSelector, recording_plot_utils, mu_tol=1e-9, sig_tol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "i8"])
@pytest.mark.parametrize("seq", SEQUENCES)
def test_driver_on_the_gpu_reproduces_the_reference_script(seq, precision, tmp_path, recording_plot_utils):
    from bayesian_optimisation_amd import PointSelector

    replay(seq, str(tmp_path), lambda: PointSelector(precision=precision), recording_plot_utils, mu_tol=1e-9, sig_tol=1e-8)
    # the class made the reference's ARD plot calls too (point_selector.py:146,163) whenever it searched length scales
    z = np.load(os.path.join(GOLDEN, f"driver_{seq}.npz"))
    last = int(z["n_steps"]) - 1
    want = [p["fn"] for p in json.loads(str(z[f"step{last}:plots"])) if p["fn"].startswith("plot_ARD")]
    assert [c[0] for c in recording_plot_utils if c[0].startswith("plot_ARD")] == want
