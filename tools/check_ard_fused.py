"""The one-launch likelihood grid against the oracle, and its time: python tools/check_ard_fused.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O

gp = DeviceGP()
sizes = [int(a) for a in sys.argv[1:]] or [65, 100, 176, 300, 512, 1030]
rng = np.random.default_rng(5)
for d in (2, 8):
    for N in sizes:
        X, y, _, _ = make_problem(N, 8, d)
        G = 40
        cells = rng.uniform(0.15, 2.5, size=(G, d))
        want = O.nlml_cells_logdet(X, y, cells)
        got = gp.nlml_grid(X, y, cells, likelihood="logdet")
        ref_want = O.nlml_cells_stable(X, y, cells).astype(np.float32)
        ref_got = gp.nlml_grid(X, y, cells)
        rel = np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want)))
        ok_ref = np.array_equal(np.isfinite(ref_got), np.isfinite(ref_want)) and np.allclose(
            ref_got[np.isfinite(ref_want)], ref_want[np.isfinite(ref_want)], rtol=2e-6, atol=0)
        print(f"d={d} N={N}: logdet max rel err {rel:.2e}; reference mode matches oracle: {ok_ref} "
              f"(finite {int(np.isfinite(ref_want).sum())}/{G})", flush=True)

a1 = np.linspace(0.05, 3.0, 50)
cells = np.stack(np.meshgrid(a1, a1, indexing="ij"), -1).reshape(-1, 2)
for N in (128, 176, 512, 1024):
    X, y, _, _ = make_problem(N, 8, 2)
    for mode in ("reference", "logdet"):
        gp.nlml_grid(X, y, cells, likelihood=mode)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = gp.nlml_grid(X, y, cells, likelihood=mode)
        dt = time.perf_counter() - t
        print(f"N={N} {mode}: 2,500 cells in {dt*1e3:.2f} ms = {2500 * N**3 / 3 / dt / 1e12:.1f} TFLOP/s "
              f"(finite {int(np.isfinite(out).sum())})", flush=True)
