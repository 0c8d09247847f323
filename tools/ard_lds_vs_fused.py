"""The in-LDS likelihood kernel (nlml_grid_kernel: the bordered matrix of one cell in LDS, one barrier per column) against the
fused one (nlml_fused_kernel: 64-column panels on the matrix cores) between N = 33 and the in-LDS kernel's largest size:
2,500 cells, reference mode, kernel time by HIP events.  usage: python tools/ard_lds_vs_fused.py [d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

d = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a = np.linspace(0.05, 3.0, 50)
cells = np.tile(np.geomspace(0.2, 2.0, d), (2500, 1))
cells[:, :2] = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
gp = DeviceGP()
nmax = int(gp.lib.gpbo_nlml_grid_max_n())
for N in (16, 32, 48, 64, 65, 72, 80, 96, 112, 128, 144, 160, nmax):
    X, y, _, _ = make_problem(N, 8, d)
    Xd, yd, cd = gp._dev(X), gp._dev(y), gp._dev(cells)
    res = []
    gp.ARD_KERNEL = "lds"          # (the default, "wave", serves N <= 64 with the wave-per-cell kernel: third column)
    for thr in (0, nmax):          # 0: always the fused kernel; nmax: the in-LDS kernel
        gp.ARD_LDS_MAX_N = thr
        out = gp.nlml_grid_device(Xd, yd, cd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gp.nlml_grid_device(Xd, yd, cd)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 10, out.double().cpu().numpy()))
    fin = np.isfinite(res[0][1]) & np.isfinite(res[1][1])
    dv = float(np.max(np.abs(res[0][1][fin] - res[1][1][fin]) / np.maximum(1.0, np.abs(res[1][1][fin])))) if fin.any() else float("nan")
    wv = ""
    if N <= int(gp.lib.gpbo_nlml_grid_wave_max_n()):
        gp.ARD_KERNEL = "wave"
        w = gp.nlml_grid_device(Xd, yd, cd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gp.nlml_grid_device(Xd, yd, cd)
        e1.record()
        torch.cuda.synchronize()
        wn = w.double().cpu().numpy()
        wv = f", wave {e0.elapsed_time(e1) / 10:.3f} ms ({int((wn == res[0][1]).sum())} of 2,500 float32 cells equal to the fused kernel's)"
    print(f"d={d} N={N}: fused {res[0][0]:.3f} ms, in-LDS {res[1][0]:.3f} ms, max rel. difference of the float32 cells {dv:.1e}{wv}", flush=True)
