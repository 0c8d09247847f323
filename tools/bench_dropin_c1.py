"""Latency of the drop-in class at BASELINE config 1's shape (d=2, N=32, 50x50 ARD grid, M=32x32 and 50x50),
the sizes the reference's DAG actually runs (golden fixture inputs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import PointSelector, PointSelectorHost

for name in ("g1_m32", "g1_m50"):
    g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".npz")))
    def run(preset, cls=PointSelector):
        ps = cls()
        ps.name, ps.iteration = "T", 0
        ps.measured_pts, ps.measured_vals = g["X"], g["y"]
        ps.feature_domain = [int(v) for v in g["feature_domain"]]
        ps.predicted_pts = g["Xs"]
        if preset:
            ps.set_kernel_params(g["kernel_params"])
        else:
            ps.length_scales = g["length_scales"]
        ps.update_surrogate()
        return ps.lower_confidence_bound()
    for preset in (False, True):
        run(preset); run(preset)
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            t = time.perf_counter(); idx = run(preset); ts.append(time.perf_counter() - t)
        print(f"{name} M={len(g['Xs'])} {'preset ls' if preset else 'with 50x50 ARD grid'}: median {np.median(ts)*1e3:.2f} ms, "
              f"index {idx.tolist()} (reference {g['index'].tolist()})")
    for preset in (False, True):
        run(preset, PointSelectorHost); ts = []
        for _ in range(7):
            t = time.perf_counter(); idx = run(preset, PointSelectorHost); ts.append(time.perf_counter() - t)
        print(f"{name} host-pointer class, {'preset ls' if preset else 'with 50x50 ARD grid'}: median {np.median(ts)*1e3:.2f} ms, index {idx.tolist()}")
