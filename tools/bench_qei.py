"""Timing of the q=8 Monte-Carlo qEI path at BASELINE config 5's per-GPU shape (d=8, N=2048, 512 samples)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import ard_length_scales, rff_objective, sobol_points

N, d, M, S = 2048, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17, 512
ls = ard_length_scales(d); X = sobol_points(0, N, d); y = rff_objective(X, ls); Xs = sobol_points(N, M, d)
Z = np.random.default_rng(7).standard_normal((S, 8))
gp = DeviceGP().factorise(X, y, ls)
Xsd, Zd = gp._dev(Xs), gp._dev(Z)
for _ in range(2):
    r = gp.score_qei(Xsd, Zd, float(y.min()))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    r = gp.score_qei(Xsd, Zd, float(y.min()))
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
r1 = gp.score(Xsd)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    r1 = gp.score(Xsd)
torch.cuda.synchronize(); dt1 = (time.perf_counter() - t) / 3
print(f"qEI N={N} M={M} q=8 S={S}: {dt*1e3:.1f} ms -> {M/dt:.3e} candidates/s ({M/8/dt:.3e} batches/s); "
      f"single-point LCB pass on the same candidates {dt1*1e3:.1f} ms; best batch {r.best_idx}, nan {r.nan_count}")
