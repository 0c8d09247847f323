"""Host-side logic of the drop-in classes that needs no GPU: the length-scale branch of update_surrogate()
(point_selector.py:60-73), tune_kernel's first-minimum rule on the float32 grid (:141, :159) and the shape of
`kernel_params`, driven with the likelihood grids stored in the reference's golden vectors instead of a device."""
import numpy as np
import pytest

from bayesian_optimisation_amd import PointSelector
from bayesian_optimisation_amd import host_binding as H


class _GridFromFixture:
    """Stands in for the surrogate object inside tune_kernel: returns the reference's own float32 grid."""

    def __init__(self, grid):
        self.grid = np.asarray(grid, dtype=np.float32)
        self.calls = 0

    def nlml_grid(self, X, y, cells, jitter=1e-4):
        self.calls += 1
        assert len(cells) == self.grid.size
        return self.grid.ravel().copy()


@pytest.mark.parametrize("name", ["g1_m32", "g1_m50", "g4_ard_n2", "g2_n5_a", "g2_n20_tr", "g2_n12_a"])
def test_tune_kernel_picks_the_references_cell(golden, name):
    g = golden(name)
    ps = PointSelector()
    ps._gp = _GridFromFixture(g["nlogml"])
    ps.measured_pts, ps.measured_vals = g["X"], g["y"]
    ps.length_scales = g["length_scales"]
    ls = ps._select_kernel_params(np.asarray(g["X"], dtype=np.float64))
    assert ps._gp.calls == 1
    assert np.array_equal(np.asarray(ps.kernel_params), g["kernel_params"])
    assert ps.kernel_params.shape == g["kernel_params"].shape          # (2,) for 2-D, (1, 1) for the 1-D search (:161)
    assert ls.ndim == 1 and np.array_equal(ls, np.ravel(g["kernel_params"]))
    assert ps.nlogml.shape == g["nlogml"].shape


def test_first_minimum_wins_on_float32_ties():
    ps = PointSelector()
    grid = np.full((3, 4), 5.0, dtype=np.float32)
    grid[1, 2] = grid[2, 0] = 1.0                                        # tie: row-major first is (1, 2)
    ps._gp = _GridFromFixture(grid)
    ps.measured_pts, ps.measured_vals = np.zeros((3, 2)), np.zeros(3)
    ps.length_scales = [np.array([1.0, 2.0, 3.0]), np.array([10.0, 20.0, 30.0, 40.0])]
    ps.tune_kernel()
    assert np.array_equal(ps.kernel_params, [2.0, 30.0])
    nan_grid = grid.copy()
    nan_grid[0, 0] = np.nan                                              # the reference: amin is NaN -> IndexError
    ps._gp = _GridFromFixture(nan_grid)
    with pytest.raises(IndexError):
        ps.tune_kernel()


@pytest.mark.parametrize("name", ["g2_n1_tr", "g3_n1_2d"])
def test_single_observation_takes_the_middle_of_each_axis(golden, name):
    g = golden(name)
    ps = PointSelector()
    ps._gp = _GridFromFixture(np.zeros(1))                               # must not be consulted
    ps.length_scales = g["length_scales"]
    ls = ps._select_kernel_params(np.asarray(g["X"], dtype=np.float64))
    assert ps._gp.calls == 0
    assert np.array_equal(np.asarray(ps.kernel_params), g["kernel_params"])
    assert np.array_equal(ls, np.ravel(g["kernel_params"]))


def test_preset_kernel_params_skip_the_search():
    ps = PointSelector()
    ps._gp = _GridFromFixture(np.zeros(1))
    ps.set_kernel_params([0.3, 0.4, 0.5])
    ls = ps._select_kernel_params(np.zeros((10, 3)))
    assert ps._gp.calls == 0 and np.array_equal(ls, [0.3, 0.4, 0.5])


def test_host_binding_checks_shapes_before_touching_the_device():
    X, y, Xs = np.zeros((5, 2)), np.zeros(5), np.zeros((7, 2))
    with pytest.raises(ValueError):
        H.select_next(X, y, [1.0, 1.0, 1.0], Xs)                         # one length scale too many
    with pytest.raises(ValueError):
        H.select_next(X, y[:4], [1.0, 1.0], Xs)
    with pytest.raises(ValueError):
        H.select_next(X, y, [1.0, 1.0], np.zeros((7, 3)))
    with pytest.raises(ValueError):
        H.select_next(X, y, [1.0, 1.0], Xs, acquisition="ucb")
    with pytest.raises(ValueError):
        H.select_next(X, y, [1.0, 1.0], Xs, acquisition="ei")            # EI needs the incumbent


def test_precision_and_q_ei_argument_checks():
    with pytest.raises(ValueError):
        PointSelector(precision="fp16")
    with pytest.raises(RuntimeError):
        H.PointSelectorHost().q_expected_improvement()                   # before update_surrogate()
    ps = PointSelector()
    with pytest.raises(RuntimeError):
        ps.lower_confidence_bound()                                      # before update_surrogate()


def test_replayed_pmc_figures_belong_to_todays_kernel_sources():
    """bench.py replays roofline.traffic / kstar_roofline.valu from the committed rocprofv3 --pmc pass of the headline shape
    (counters cannot be read inside an un-profiled run).  The entry stores the hash of the kernel sources it was collected on
    (profiles/source_hash.py); this test fails when those sources have changed since - re-run profiles/collect.sh and
    profiles/summarise.py - and bench.py reports `traffic_stale: true` in that case."""
    import importlib.util
    import json
    import os

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("source_hash", os.path.join(repo, "profiles", "source_hash.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    shapes = json.load(open(os.path.join(repo, "profiles", "pmc_sigma_acq.json")))
    head = shapes["N=4096,d=8,dtype=f64,candidates_per_launch=131072"]
    assert "kernel_source_hash" in head, "the headline entry must say which sources it was collected on"
    for key, e in shapes.items():
        if "kernel_source_hash" in e:
            dtype = key.split("dtype=")[1].split(",")[0]
            assert e["kernel_source_hash"] == sh.kernel_source_hash(dtype), \
                f"{key}: kernel sources changed since {e['source']} was collected: re-run profiles/collect.sh + summarise.py"
    # and bench.py's helper says the same thing
    import sys

    sys.path.insert(0, repo)
    import bench

    e, stale = bench._pmc_entry(4096, 8, "f64", 131072)
    assert e is head or e == head
    assert stale is False
    # round 5: every replayed entry was re-collected on the final sources and carries its hash - none is "unknown" any more
    for key in shapes:
        assert "kernel_source_hash" in shapes[key], key
    e, stale = bench._pmc_entry(4096, 8, "i8", 131072)
    assert e is not None and stale is False
    # ... and the likelihood grid's entries (profiles/pmc_ard.json, profiles/collect_ard.sh) likewise
    ard = json.load(open(os.path.join(repo, "profiles", "pmc_ard.json")))
    assert {"N=32,d=2,cells=2500", "N=64,d=2,cells=2500", "N=176,d=2,cells=2500", "N=512,d=8,cells=2500", "N=1024,d=8,cells=2500"} <= set(ard)
    assert ard["N=32,d=2,cells=2500"]["hash_key"] == ard["N=64,d=2,cells=2500"]["hash_key"] == "ard_wave"   # (the wave-per-cell kernel)
    for key, e in ard.items():
        assert e["kernel_source_hash"] == sh.kernel_source_hash(e.get("hash_key", "ard")), \
            f"{key}: ard.hip changed since {e['source']} was collected: re-run profiles/collect_ard.sh + summarise_ard.py"
    e, stale = bench._pmc_ard_entry(512, 8)
    assert e is not None and stale is False
    assert bench._pmc_entry(123, 8, "f64", 131072) == (None, None)
