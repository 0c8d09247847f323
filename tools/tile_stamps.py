"""Diagnostic (needs a diagnostics build of the library: GPBO_DIAG=1 bash bayesian_optimisation_amd/csrc/build.sh;
GPBO_SIGMA_VARIANT=6): per-tile s_memtime stamps of the variance kernel's first launch.
Prints the median duration (shader cycles, 100 MHz memtime ticks are converted by the caller's reading) per tile."""
import os, sys
os.environ["GPBO_SIGMA_VARIANT"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = 1 << 18
X, y, Xs, ls = make_problem(N, M, 8)
gp = DeviceGP(chunk=1 << 17).factorise(X, y, ls)
for _ in range(3):
    r = gp.score(Xs)
Np, chunk = gp.Np, 1 << 17
al = lambda v: (v + 255) // 256 * 256
off = 0                                                         # stamps land in chunk buffer 0 (free after chunk 0)
w = gp._work_post
nJ = Np // 128
T = min(8 * nJ * (nJ + 1) // 2, 1000)
raw = w[off // 8: off // 8 + 512 * 1024].cpu().numpy().reshape(512, 1024)
st = raw[:, :T]
Tall = 8 * nJ * (nJ + 1) // 2
clk = ((raw[:256, T - 1] - raw[:256, 0]) / (raw[:256, 1001] - raw[:256, 1000]) * 100e6) if Tall <= 1000 else np.array([np.nan])
print(f"shader clock during the kernel (s_memtime / s_memrealtime): median {np.median(clk)/1e9:.3f} GHz")
d = np.diff(st, axis=1)                                          # per-tile durations, memtime ticks
med = np.median(d[:256], axis=0)
tiles = [(jb, kt) for jb in range(nJ) for kt in range(8 * (jb + 1))][:T]
print("ticks per tile (median over the first 256 workgroups); tick = 10 ns @100 MHz")
for jb in range(min(nJ, 4)):
    row = [med[i] for i, (j, k) in enumerate(tiles[:-1]) if j == jb]
    print(f"jb={jb}:", " ".join(f"{v:5.0f}" for v in row))
print("total ticks per workgroup:", np.median(st[:256, -1] - st[:256, 0]))
