"""Three calls of the likelihood grid (2,500 cells, the first is the warm-up that allocates the workspace) for the profiler:
python tools/ard_profile_one.py N [d] [likelihood]    (profiles/collect_ard.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

N = int(sys.argv[1])
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else "reference"
a = np.linspace(0.05, 3.0, 50)
if d == 2:
    cells = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
else:   # 2,500 cells of a d-feature search: the first two coordinates over the 50 x 50 grid, the others at geomspace(0.2, 2)
    cells = np.tile(np.geomspace(0.2, 2.0, d), (2500, 1))
    cells[:, :2] = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
X, y, _, _ = make_problem(N, 8, d)
gp = DeviceGP()
for _ in range(3):
    out = gp.nlml_grid(X, y, cells, likelihood=mode)
torch.cuda.synchronize()
print(f"N={N} d={d} {mode}: finite cells {int(np.isfinite(out).sum())} of {len(out)}")
