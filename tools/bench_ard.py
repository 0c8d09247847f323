"""Time of the ARD likelihood grid (2,500 cells, d=2) at several N: python tools/bench_ard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

a1, a2 = np.linspace(0.05, 3.0, 50), np.linspace(0.05, 3.0, 50)
cells = np.stack(np.meshgrid(a1, a2, indexing="ij"), -1).reshape(-1, 2)
gp = DeviceGP()
for N in (32, 64, 128, 176, 200, 512, 1024):
    X, y, _, _ = make_problem(N, 8, 2)
    gp.nlml_grid(X, y, cells)  # warm-up with the full cell count: the workspace (6.6 GB at N=512) is allocated here, not in the timed call
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = gp.nlml_grid(X, y, cells)
    dt = time.perf_counter() - t
    print(f"N={N}: 2,500 cells in {dt*1e3:.2f} ms  (finite cells: {int(np.isfinite(out).sum())})", flush=True)

# the batched blocked Cholesky at sizes the in-LDS kernel also handles (where is the cross-over?)
import ctypes as C
for N in (64, 128, 176, 512):
    X, y, _, _ = make_problem(N, 8, 2)
    Xd, yd, cd = gp._dev(X), gp._dev(y), gp._dev(cells)
    out = torch.empty(len(cells), dtype=torch.float32, device=gp.device)
    need = int(gp.lib.gpbo_nlml_grid_batched_workspace_bytes(N, len(cells)))
    work = torch.empty(need // 8 + 1, dtype=torch.float64, device=gp.device)
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        gp.lib.gpbo_nlml_grid_batched_f64(gp._ptr(Xd), gp._ptr(yd), N, 2, gp._ptr(cd), len(cells), 1e-4, gp._ptr(out),
                                          gp._ptr(work), need, gp._stream())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"batched route, N={N}: 2,500 cells in {dt*1e3:.2f} ms", flush=True)
