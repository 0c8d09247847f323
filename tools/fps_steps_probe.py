"""Per-member cost of the farthest-point order (gpbo_fps_order_f64 alone, hipEvents): the slope over J at fixed N separates the
selection steps from the fixed part (centroid, extension, gathers, launches).  python tools/fps_steps_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
gp = DeviceGP()
lib = gp.lib
for N, d in ((1024, 8), (2048, 8), (4096, 8), (4096, 16), (8192, 8), (8192, 16), (16384, 8)):
    X, y, Xs, ls = make_problem(N, 512, d)
    Xd, yd = gp._dev(X), gp._dev(y)
    wb = int(lib.gpbo_fps_order_workspace_bytes(N))
    w = torch.empty(wb // 8 + 1, dtype=torch.float64, device=gp.device)
    perm = torch.empty(N, dtype=torch.int64, device=gp.device)
    Xp, yp = torch.empty_like(Xd), torch.empty_like(yd)
    lsp = np.ascontiguousarray(ls).ctypes.data_as(C.c_void_p)
    res = []
    for J in (64, 128, 256, 512):
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = lib.gpbo_fps_order_f64(gp._ptr(Xd), gp._ptr(yd), N, d, lsp, J, gp._ptr(perm), gp._ptr(Xp), gp._ptr(yp), gp._ptr(w), wb, gp._stream())
            e1.record()
            assert st == 0
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append((J, min(ts)))
    slope = (res[-1][1] - res[0][1]) / (res[-1][0] - res[0][0]) * 1e3
    print(f"N={N} d={d}: " + ", ".join(f"J={J}: {t*1e3:.0f} us" for J, t in res) + f"  -> {slope:.2f} us per member, fixed {res[0][1]*1e3 - 64*slope:.0f} us", flush=True)
