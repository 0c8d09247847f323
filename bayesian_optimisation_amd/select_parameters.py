"""`SELECT_PARAMETERS` step of the reference's DAG, driven by the MI355X `PointSelector`.

Behavioural mirror of /root/reference/select_parameters.py (a script that runs top to bottom at import
with hard-coded site paths) as a callable: same inputs (`opto_log.JSON`, `measured_points/*.npy`), same
outputs (next sample appended to the `.npy` with a placeholder objective, `opto_log.JSON` rewritten with
`json.dump(indent=4)`, RAT macro + simulate.submit when the template is present, lines appended to
`algo_log.txt`, the reference's `plot_utils` panels when that module is importable), same branch structure:

  * 1-D tuning (amplitudes `[5, 6]` / `[7, 8]`, rise time `[4]`)           select_parameters.py:120-207
  * 2-D tuning of a decay-constant pair, incl. the very first random draw   select_parameters.py:209-337

The on-disk formats are unchanged: `.npy` rows `[f, obj]` / `[f0, f1, obj]` float64,
placeholder objectives 10000 (1000 for the first random point), JSON schema untouched.
Only the GP step differs: it runs on the GPU through `PointSelector` (no CPU fallback).

Usage inside the DAG (executables/select_parameters.sh):   python3 -m bayesian_optimisation_amd.select_parameters
"""
from __future__ import annotations

import json
import os
import string
from typing import Callable, Optional

import numpy as np

PARAMETER_NAMES = np.array(["T1", "T2", "T3", "T4", "TR", "A1", "A2", "A3", "A4"])  # select_parameters.py:59
GRANULARITY = 50                                                                     # :62


def domains():
    """Parameter domains (select_parameters.py:63-67)."""
    g = GRANULARITY
    return dict(T1=np.linspace(1, 14, g), T2=np.linspace(10, 90, g), T3=np.linspace(60, 150, g),
                T4=np.linspace(200, 500, g), TR=np.linspace(0.1, 2.0, g))


def length_scale_grids():
    """ARD search grids per parameter index (select_parameters.py:70-75)."""
    l1, l2 = np.linspace(0.5, 10, 50), np.linspace(2, 100, 50)
    l3, l4 = np.linspace(10, 30, 50), np.linspace(50, 100, 50)
    lth = np.linspace(0.1, 2, 20)
    return [l1, l2, l3, l4, l1, lth, lth, lth, lth]


def candidate_grid(axis0: np.ndarray, axis1: np.ndarray) -> np.ndarray:
    """Row-major grid X*[i*g + j] = (axis0[i], axis1[j])  (select_parameters.py:273-279)."""
    a, b = np.meshgrid(axis0, axis1, indexing="ij")
    return np.stack([a.ravel(), b.ravel()], axis=1)


def _plot_utils():
    """The reference's plotting module (plot_utils.py; the script star-imports it, select_parameters.py:2) when it is
    importable from the job's working directory - plots are a side output and never a reason to fail the step."""
    try:
        import plot_utils  # type: ignore

        return plot_utils
    except Exception:  # noqa: BLE001
        return None


def _plot(fn_name, *args):
    pu = _plot_utils()
    fn = getattr(pu, fn_name, None) if pu is not None else None
    if fn is None:
        return False
    try:
        fn(*args)
    except Exception:  # noqa: BLE001
        return False
    return True


class _Log:
    def __init__(self, path, echo):
        self.path, self.echo = path, echo

    def __call__(self, msg):
        with open(self.path, "a") as f:
            f.write(msg)
        if self.echo:
            print(msg)


def _write_macro(base_dir, params_update, name):
    """create_macro (select_parameters.py:14-42): fill the RAT macro template and point simulate.submit at it.
    Skipped when the template is not there (it belongs to the experiment, not to this package)."""
    tpl = os.path.join(base_dir, "bi214_template.mac")
    if not os.path.exists(tpl):
        return False
    with open(tpl) as f:
        raw = string.Template(f.read())
    d, a, tr = params_update[0:4], params_update[5:], params_update[4]
    text = raw.substitute(MATERIAL="labppo_2p2_scintillator", T1=d[0], T2=d[1], T3=d[2], T4=d[3], TR=tr,
                          A1=a[0], A2=a[1], A3=a[2], A4=a[3])
    os.makedirs(os.path.join(base_dir, "macros"), exist_ok=True)
    with open(os.path.join(base_dir, "macros", f"{name}.mac"), "w") as f:
        f.write(text)
    sub_dir = os.path.join(base_dir, "submit_files")
    os.makedirs(sub_dir, exist_ok=True)
    with open(os.path.join(sub_dir, "simulate.submit"), "w") as f:
        f.write(f"""
        executable = {base_dir}/executables/submit_simulations.sh
        arguments  = $(Process) {name}.mac
        log        = {base_dir}/condor_logs/sim_logs/$(Process).log
        output     = {base_dir}/condor_logs/sim_outputs/$(Process).out
        error      = {base_dir}/condor_logs/sim_errors/$(Process).err
        request_memory = 1024MB
        queue 10
        """)
    return True


def _params_vector(parameters):
    return np.array([parameters[k] for k in ("T1", "T2", "T3", "T4")] + [parameters["TR"]]
                    + [parameters[k] for k in ("A1", "A2", "A3", "A4")])


def _convergence(info, within):
    ps = info["iteration_info"]["current_block"]["param_sampling"]
    if within:
        ps["conv_points"] += 1      # select_parameters.py:190-193 / :321-325
    else:
        ps["conv_points"] = 0       # :195-199 / :327-331


def select_parameters(base_dir: str = ".", selector_factory: Optional[Callable] = None,
                      rng: Optional[np.random.Generator] = None, echo: bool = False,
                      keep_surrogate: bool = False, likelihood: str = "reference") -> dict:
    """Run one SELECT_PARAMETERS step in `base_dir`.  Returns a summary dict (what was chosen).
    keep_surrogate: carry the factorisation from job to job in `<measured_points file>.surrogate.npz` (the factory is
    then called with `state_path=...`; new rows are appended in O(N^2) while the chosen length scales stay the same).
    likelihood: "reference" (the reference's ARD likelihood, determinant underflow included) or "logdet" (opt-in, finite at
    any number of observations: INTEGRATION.md section 4); anything but the default is passed on to the selector's constructor."""
    if likelihood not in ("reference", "logdet"):
        raise ValueError("likelihood must be 'reference' or 'logdet'")
    if selector_factory is None:
        from .point_selector import PointSelector as selector_factory  # the GPU class (raises without a GPU)
    if likelihood != "reference":
        import functools

        selector_factory = functools.partial(selector_factory, likelihood=likelihood)
    log = _Log(os.path.join(base_dir, "algo_log.txt"), echo)
    json_path = os.path.join(base_dir, "opto_log.JSON")
    mp_dir = os.path.join(base_dir, "measured_points")
    os.makedirs(mp_dir, exist_ok=True)
    log("\n\n### THIS IS SELECT PARAMETERS SCRIPT ###\n")
    with open(json_path) as f:
        info = json.load(f)

    sample_info = info["iteration_info"]["current_block"]["param_sampling"]
    iteration = sample_info["param_sample_iter"]
    curr_params = sample_info["current_parameters"]
    dom, lsg = domains(), length_scale_grids()
    algo_iter = info["iteration_info"]["full_algo_iter"]
    block_iter = info["iteration_info"]["current_block"]["iteration"]
    block_name = info["iteration_info"]["current_block"]["block_name"]
    block_best = info["iteration_info"]["current_block"]["block_best_params"]
    log(f"\n SELECT PARAMETERS for ITERATION {iteration} in BLOCK {block_name} iter {block_iter} "
        f"and ALGO LOOP {algo_iter}.")
    summary = dict(iteration=iteration, curr_params=list(curr_params))

    if curr_params in ([5, 6], [7, 8], [4]):
        # ---------------- 1-D: amplitudes or rise time (select_parameters.py:120-207) ----------------
        feature_name = str(PARAMETER_NAMES[curr_params[0]])
        max_weight = None
        if curr_params[0] == 4:
            feature_domain = dom["TR"]
        else:
            max_weight = 0.9 if curr_params == [5, 6] else 0.1                       # :78-83
            feature_domain = np.linspace(0.01, max_weight, GRANULARITY)
        npy = os.path.join(mp_dir, f"{feature_name}_ALGO_{algo_iter}_BLOCK_{block_iter}.npy")
        if iteration == 0:
            measured_points = np.array([[block_best[feature_name], block_best["obj"]]])   # :135-139
        else:
            measured_points = np.load(npy)                                            # :142
        opt = selector_factory(state_path=npy[:-4] + ".surrogate.npz") if keep_surrogate else selector_factory()
        opt.name, opt.iteration = feature_name, iteration
        opt.measured_pts = measured_points[:, 0].reshape((len(measured_points), 1))
        opt.measured_vals = measured_points[:, 1]
        opt.feature_domain = [GRANULARITY]
        opt.predicted_pts = feature_domain.reshape((GRANULARITY, 1))
        opt.length_scales = lsg[curr_params[0]]
        opt.update_surrogate()
        next_sample = opt.lower_confidence_bound()
        updated = feature_domain[next_sample[0]]
        log(f"\nSelected {feature_name} = {updated} as next sample position.")
        rows = measured_points.tolist()
        rows.append([updated, 10000])                                                 # :163 placeholder objective
        np.save(npy, rows)
        # :167-170 the iteration's panel, from host copies of the posterior
        plot_name = f"{feature_name}_ALGO_{algo_iter}_BLOCK_{block_iter}_{iteration}"
        _plot("surrogate_uncert_acquistion_1d", opt.mean_func, opt.cov_func, opt.acq_func_eval, opt.predicted_pts,
              plot_name, iteration, rows)

        parameters = info["parameters"]
        params_update = _params_vector(parameters)
        if feature_name == "TR":
            params_update[curr_params] = updated
        else:
            params_update[curr_params] = [updated, max_weight - updated]
        _write_macro(base_dir, params_update, feature_name)
        last = parameters[feature_name]
        perc = abs(last - updated) / last
        log(f"\n% Change from last measurement is: {last} --> {updated} ({perc}) %.")
        _convergence(info, perc <= 0.05)
        info["parameters"][feature_name] = updated
        if feature_name != "TR":
            info["parameters"][str(PARAMETER_NAMES[curr_params[1]])] = max_weight - updated
        summary.update(feature=feature_name, selected=float(updated), index=[int(next_sample[0])], selector=opt)
    else:
        # ---------------- 2-D: a pair of decay constants (select_parameters.py:209-337) ----------------
        cp = np.array(curr_params)
        names = [str(v) for v in PARAMETER_NAMES[cp]]
        npy = os.path.join(mp_dir, f"{names[0]}_{names[1]}_ALGO_{algo_iter}_BLOCK_{block_iter}.npy")
        parameters = info["parameters"]
        first_ever = algo_iter == 0 and block_iter == 0 and iteration == 0 and list(curr_params) == [0, 1]
        if first_ever:
            # truly the first tuning of (T1, T2): random grid point, no GP   (:217-250)
            rng = rng or np.random.default_rng()
            idx = rng.integers(0, GRANULARITY, size=2)
            updated = np.array([dom["T1"][idx[0]], dom["T2"][idx[1]]])
            log(f"\nFirst iteration for algorithm. Randomly selected {names[0]} = {updated[0]} and "
                f"{names[1]} = {updated[1]}.")
            params_update = _params_vector(parameters)
            params_update[cp] = updated
            _write_macro(base_dir, params_update, f"{names[0]}_{names[1]}")
            for key, val in zip(("T1", "T2"), updated):
                info["parameters"][key] = val
                info["iteration_info"]["initial_parameters"][key] = val
                info["iteration_info"]["current_block"]["prev_params"][key] = val
            np.save(npy, [updated.tolist() + [1000]])                                 # :249 first placeholder
            summary.update(feature=names, selected=updated.tolist(), index=[int(v) for v in idx], selector=None)
        else:
            if iteration == 0:
                measured_points = np.array([[block_best[names[0]], block_best[names[1]], block_best["obj"]]])
            else:
                measured_points = np.load(npy)                                        # :265
            axes = [dom["T1"], dom["T2"]] if curr_params[0] == 0 else [dom["T3"], dom["T4"]]
            opt = selector_factory(state_path=npy[:-4] + ".surrogate.npz") if keep_surrogate else selector_factory()
            opt.name, opt.iteration = PARAMETER_NAMES[cp], iteration
            opt.measured_pts = measured_points[:, 0:2].reshape((len(measured_points), 2))
            opt.measured_vals = measured_points[:, 2]
            opt.feature_domain = [GRANULARITY, GRANULARITY]
            opt.predicted_pts = candidate_grid(axes[0], axes[1])
            opt.length_scales = np.array([lsg[curr_params[0]], lsg[curr_params[1]]])
            opt.update_surrogate()
            next_sample = opt.lower_confidence_bound()
            updated = np.array([axes[0][next_sample[0]], axes[1][next_sample[1]]])
            log(f"\nNext sample at {names[0]} = {updated[0]} | {names[1]} = {updated[1]}.")
            rows = measured_points.tolist()
            rows.append([updated[0], updated[1], 10000])                              # :299
            np.save(npy, rows)
            # :303-307 the iteration's panel, from host copies of the posterior
            plot_name = f"{names[0]}_{names[1]}_ALGO_{algo_iter}_BLOCK_{block_iter}_{iteration}"
            mesh_x, mesh_y = np.meshgrid(axes[0], axes[1])
            _plot("surrogate_uncert_acquistion", opt.mean_func, opt.cov_func, opt.acq_func_eval, mesh_x, mesh_y,
                  plot_name, iteration, rows)
            params_update = _params_vector(parameters)
            params_update[cp] = updated
            _write_macro(base_dir, params_update, f"{names[0]}_{names[1]}")
            last = np.array([parameters[names[0]], parameters[names[1]]])
            perc = abs(last - updated) / last
            log(f"\n% Change from last measurement is: {last} --> {updated} ({perc}) %.")
            _convergence(info, bool(np.all(perc <= 0.05)))
            info["parameters"][names[0]] = updated[0]
            info["parameters"][names[1]] = updated[1]
            summary.update(feature=names, selected=updated.tolist(), index=[int(v) for v in next_sample],
                           selector=opt)

    with open(json_path, "w") as f:
        json.dump(info, f, indent=4)                                                  # :207 / :337
    return summary


def main():
    select_parameters(os.environ.get("GPBO_BASE_DIR", os.getcwd()), echo=True,
                      keep_surrogate=os.environ.get("GPBO_KEEP_SURROGATE", "0") == "1",
                      likelihood=os.environ.get("GPBO_LIKELIHOOD", "reference"))


if __name__ == "__main__":
    main()
