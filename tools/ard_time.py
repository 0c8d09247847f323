"""Kernel time of the likelihood grid (2,500 cells, HIP events on the launch stream): python tools/ard_time.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

d = int(os.environ.get("ARD_D", "2"))
a = np.linspace(0.05, 3.0, 50)
cells = np.tile(np.geomspace(0.2, 2.0, d), (2500, 1))
cells[:, :2] = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
gp = DeviceGP()
out = []
for N in [int(v) for v in sys.argv[1:]] or [176, 512, 1024]:
    X, y, _, _ = make_problem(N, 8, d)
    Xd, yd, cd = gp._dev(X), gp._dev(y), gp._dev(cells)
    gp.nlml_grid_device(Xd, yd, cd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gp.nlml_grid_device(Xd, yd, cd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f"N={N}: {ms:.3f} ms ({2500 * N**3 / 3 / ms / 1e9:.1f} TF)")
print(os.environ.get("GPBO_LIB", "libgpbo.so").split("/")[-1], f"d={d}", " | ".join(out), flush=True)
