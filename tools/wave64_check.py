"""The wave-per-cell likelihood kernel (csrc/ard_wave.hip) at N = 33 ... 64 called through the ABI: kernel time (2,500 cells,
HIP events), float32 cells against the fused kernel's and against the oracle on a sample.  usage: python tools/wave64_check.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem
from oracle import gp_oracle as O
gp = DeviceGP()
print("wave max n", gp.lib.gpbo_nlml_grid_wave_max_n())
a = np.linspace(0.05, 3.0, 50)
for d in (2, 8, 16):
    cells = np.tile(np.geomspace(0.2, 2.0, d), (2500, 1))
    cells[:, :2] = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
    for N in (33, 40, 48, 49, 56, 64):
        X, y, _, _ = make_problem(N, 8, d)
        Xd, yd, cd = gp._dev(X), gp._dev(y), gp._dev(cells)
        out = torch.empty(2500, dtype=torch.float32, device=gp.device)
        def run():
            st = gp.lib.gpbo_nlml_grid_wave_f64(gp._ptr(Xd), gp._ptr(yd), N, d, gp._ptr(cd), 2500, 1e-4, gp._ptr(out), gp._stream())
            assert st == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        w = out.cpu().numpy()
        f = gp.nlml_grid(X, y, cells)     # fused (N > 32)
        sel = np.linspace(0, 2499, 40).astype(int)
        ref = O.nlml_cells(X, y, cells[sel])
        fin = np.isfinite(ref)
        print(f"d={d} N={N}: wave {e0.elapsed_time(e1)/10:.3f} ms; equal float32 cells vs fused: {int((w==f).sum())}/2500, max rel diff {np.nanmax(np.abs(w-f)/np.maximum(1,np.abs(f))):.1e}; vs oracle {np.max(np.abs(w[sel][fin]-ref[fin])/np.maximum(1,np.abs(ref[fin]))):.1e}", flush=True)
