#!/usr/bin/env python3
"""Golden fixtures for the SELECT_PARAMETERS step, produced by RUNNING the reference's own driver script.

/root/reference/select_parameters.py executes top to bottom at import and reads / writes absolute site paths
(/home/hunt-stokes/bayesian_optimisation/...).  Here it is run unmodified under `runpy`, in the build container
only, with
  * `open`, `np.load`, `np.save` remapping that site prefix into a scratch directory (cwd = the same directory, so the
    script's relative `measured_points/...`, `macros/...` paths land there too),
  * a stand-in `plot_utils` module that draws nothing and RECORDS the calls the script and the reference's
    PointSelector make (function name, plot name, iteration, array shapes),
  * NumPy's global random state seeded before the very first (random) step.
Nothing of the reference is stored: the fixtures are the state files going in (opto_log.JSON, measured_points/*.npy)
and coming out, the arrays handed to the plot calls (mean, sigma, acquisition), and the recorded plot calls.

Sequences (select_parameters.py:120-207 and :209-337):
  pair01   first-ever 2-D step (random grid point, :217-250), then three GP steps (N = 1 mid-point length scales,
           N = 2 and 3 with the 50 x 50 ARD grid), each followed by what time_residuals.py:204-217 does to the state
  pair23   (T3, T4) at iteration 0 of a restarted block (row taken from block_best_params, :259-260), then one more
  amp56    1-D amplitudes [5, 6]: iteration 0 from block_best_params (:135-139), iteration 1 from the .npy (:142)
  amp78    1-D amplitudes [7, 8] (max weight 0.1), two steps
  rise     1-D rise time [4], two steps
Usage: python tests/golden/make_driver_golden.py      ->  tests/golden/driver_<sequence>.npz
"""
import builtins
import contextlib
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SITE = "/home/hunt-stokes/bayesian_optimisation"

import numpy as np  # noqa: E402

CALLS = []


def _shape(a):
    try:
        return list(np.shape(a))
    except Exception:  # noqa: BLE001
        return None


def _stub_plot_utils():
    m = types.ModuleType("plot_utils")

    def surrogate_uncert_acquistion(mu, cov, acq, meshX, meshY, name, iteration, measured_points):
        CALLS.append(dict(fn="surrogate_uncert_acquistion", name=str(name), iteration=int(iteration),
                          shapes=[_shape(mu), _shape(cov), _shape(acq), _shape(meshX), _shape(meshY)],
                          n_measured=len(measured_points), mu=np.array(mu), cov=np.array(cov), acq=np.array(acq)))

    def surrogate_uncert_acquistion_1d(mu, cov, acq, pts, name, iteration, measured_points):
        CALLS.append(dict(fn="surrogate_uncert_acquistion_1d", name=str(name), iteration=int(iteration),
                          shapes=[_shape(mu), _shape(cov), _shape(acq), _shape(pts)], n_measured=len(measured_points),
                          mu=np.array(mu), cov=np.array(cov), acq=np.array(acq)))

    def plot_ARD_LL(nlogml, kernel_params, length_scales, name, iteration):
        CALLS.append(dict(fn="plot_ARD_LL", iteration=int(iteration), shapes=[_shape(nlogml)],
                          kernel_params=np.array(kernel_params, dtype=np.float64).reshape(-1)))

    def plot_ARD_LL_1d(nlogml, kernel_params, length_scales, name, iteration):
        CALLS.append(dict(fn="plot_ARD_LL_1d", iteration=int(iteration), shapes=[_shape(nlogml)],
                          kernel_params=np.array(kernel_params, dtype=np.float64).reshape(-1)))

    def plot_tRes_agreement(*a, **k):
        CALLS.append(dict(fn="plot_tRes_agreement"))

    for f in (surrogate_uncert_acquistion, surrogate_uncert_acquistion_1d, plot_ARD_LL, plot_ARD_LL_1d, plot_tRes_agreement):
        setattr(m, f.__name__, f)
    m.__all__ = [f.__name__ for f in (surrogate_uncert_acquistion, surrogate_uncert_acquistion_1d, plot_ARD_LL,
                                      plot_ARD_LL_1d, plot_tRes_agreement)]
    return m


def run_reference_step(scratch):
    """One execution of the reference's select_parameters.py against the state in `scratch`."""
    def remap(p):
        p = os.fspath(p) if not isinstance(p, (int, io.IOBase)) else p
        if isinstance(p, str) and p.startswith(SITE):
            return scratch + p[len(SITE):]
        return p

    real_open, real_load, real_save = builtins.open, np.load, np.save
    builtins.open = lambda f, *a, **k: real_open(remap(f), *a, **k)
    np.load = lambda f, *a, **k: real_load(remap(f), *a, **k)
    np.save = lambda f, arr, *a, **k: real_save(remap(f), arr, *a, **k)
    cwd = os.getcwd()
    os.chdir(scratch)
    sys.modules["plot_utils"] = _stub_plot_utils()
    sys.modules.pop("point_selector", None)
    sys.path.insert(0, REF)
    del CALLS[:]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            runpy.run_path(os.path.join(REF, "select_parameters.py"), run_name="__reference_select_parameters__")
    finally:
        builtins.open, np.load, np.save = real_open, real_load, real_save
        os.chdir(cwd)
        sys.path.remove(REF)
        sys.modules.pop("plot_utils", None)
        sys.modules.pop("point_selector", None)
    return [dict(c) for c in CALLS]


def fake_time_residuals(scratch, objective):
    """What time_residuals.py:204-217 leaves behind: the objective in the last row of the newest .npy and the sampling
    iteration advanced (so the next step loads the .npy instead of block_best_params)."""
    mp = os.path.join(scratch, "measured_points")
    newest = max((os.path.join(mp, f) for f in os.listdir(mp)), key=os.path.getmtime)
    pts = np.load(newest)
    pts[-1, -1] = objective
    np.save(newest, pts)
    with open(os.path.join(scratch, "opto_log.JSON")) as f:
        info = json.load(f)
    info["parameters"]["obj"] = objective
    info["iteration_info"]["current_block"]["param_sampling"]["param_sample_iter"] += 1
    with open(os.path.join(scratch, "opto_log.JSON"), "w") as f:
        json.dump(info, f, indent=4)


def snapshot(scratch):
    out = {"json": open(os.path.join(scratch, "opto_log.JSON")).read()}
    mp = os.path.join(scratch, "measured_points")
    for f in sorted(os.listdir(mp)):
        out["npy:" + f] = np.load(os.path.join(mp, f))
    return out


def fresh_scratch(edit=None):
    scratch = tempfile.mkdtemp(prefix="gpbo_driver_golden_")
    for d in ("measured_points", "macros", "plots", "submit_files"):
        os.makedirs(os.path.join(scratch, d))
    shutil.copy(os.path.join(REF, "bi214_template.mac"), scratch)   # read by create_macro; its output is not stored
    with open(os.path.join(HERE, "opto_log_clean.json")) as f:
        info = json.load(f)
    if edit:
        edit(info)
    with open(os.path.join(scratch, "opto_log.JSON"), "w") as f:
        json.dump(info, f, indent=4)
    return scratch


def record_sequence(name, edit, objectives, seed=None):
    """Run len(objectives) steps; store state before / after every step and the plot calls of the step."""
    scratch = fresh_scratch(edit)
    store = {"n_steps": np.array(len(objectives)), "numpy_version": np.array(np.__version__)}
    if seed is not None:
        np.random.seed(seed)
        state = np.random.get_state()
        probe = np.random.randint(50, size=2)      # what select_parameters.py:219 will draw
        np.random.set_state(state)
        store["first_random_index"] = probe
    for k, obj in enumerate(objectives):
        before = snapshot(scratch)
        calls = run_reference_step(scratch)
        after = snapshot(scratch)
        for tag, snap in (("before", before), ("after", after)):
            for key, val in snap.items():
                store[f"step{k}:{tag}:{key}"] = np.array(val) if key == "json" else val
        plots = []
        for c in calls:
            entry = {kk: vv for kk, vv in c.items() if not isinstance(vv, np.ndarray)}
            for kk, vv in c.items():
                if isinstance(vv, np.ndarray):
                    store[f"step{k}:plot{len(plots)}:{kk}"] = vv
            plots.append(entry)
        store[f"step{k}:plots"] = np.array(json.dumps(plots))
        fake_time_residuals(scratch, obj)
        store[f"step{k}:objective"] = np.array(float(obj))
    np.savez_compressed(os.path.join(HERE, f"driver_{name}.npz"), **store)
    shutil.rmtree(scratch)
    print(f"driver_{name}.npz: {len(objectives)} steps")


def main():
    def set_params(cp, block="FIRST_PAIR", block_iter=0, algo_iter=0, best=None):
        def edit(info):
            cb = info["iteration_info"]["current_block"]
            cb["param_sampling"]["current_parameters"] = cp
            cb["block_name"] = block
            cb["iteration"] = block_iter
            info["iteration_info"]["full_algo_iter"] = algo_iter
            if best:
                cb["block_best_params"].update(best)
        return edit

    record_sequence("pair01", set_params([0, 1]), [5.2e7, 3.1e7, 4.4e7, 2.9e7], seed=20240607)
    record_sequence("pair23", set_params([2, 3], block="SECOND_PAIR", block_iter=1, best=dict(T3=104.08163265306122,
                    T4=316.3265306122449, obj=3.3e7)), [2.7e7, 3.0e7])
    record_sequence("amp56", set_params([5, 6], best=dict(A1=0.7, obj=2.5e7)), [2.2e7, 2.4e7, 2.1e7])
    record_sequence("amp78", set_params([7, 8], block="SECOND_PAIR", best=dict(A3=0.05, obj=2.0e7)), [1.9e7, 1.95e7])
    record_sequence("rise", set_params([4], block="RISE_TIME", best=dict(TR=1.22, obj=1.8e7)), [1.7e7, 1.75e7, 1.72e7])


if __name__ == "__main__":
    main()
