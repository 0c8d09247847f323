"""Timeline of one workgroup's first cell of the likelihood grid (diagnostics build with -DGPBO_ARD_STAMPS):
GPBO_LIB=ab_libs/ard_stamps.so python tools/ard_stamps.py N"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP
from bayesian_optimisation_amd.synthetic import make_problem

N = int(sys.argv[1]); d = int(os.environ.get("ARD_D", "2"))
a = np.linspace(0.05, 3.0, 50)
cells = np.tile(np.geomspace(0.2, 2.0, d), (2500, 1))
cells[:, :2] = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
gp = DeviceGP()
X, y, _, _ = make_problem(N, 8, d)
lib = C.CDLL(os.environ["GPBO_LIB"])
buf = np.zeros((4, 2048), dtype=np.uint64); cnt = np.zeros(4, dtype=np.int32)
gp.nlml_grid(X, y, cells)
lib.gpbo_diag_ard_stamps(buf.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))   # discard the warm-up's
gp.nlml_grid(X, y, cells)
lib.gpbo_diag_ard_stamps(buf.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
names = {1: "panel", 2: "staged", 3: "gemm", 4: "kgen", 5: "diag-in", 6: "potrf", 7: "pre-solve", 8: "solved", 9: "end-barrier"}
for w in (0, 3):
    t = [(int(v >> np.uint64(56)), int(v & np.uint64(0xffffffffffff))) for v in buf[w, :cnt[w]]]
    tot = {}
    for (tag0, t0), (tag1, t1) in zip(t[:-1], t[1:]):
        tot[names[tag1]] = tot.get(names[tag1], 0) + (t1 - t0)
    total = t[-1][1] - t[0][1]
    print(f"wave {w}: {cnt[w]} stamps, cell = {total} shader cycles (~{total / 2000:.1f} us at 2.0 GHz)")
    for k, v in tot.items():
        print(f"   until '{k}': {v / 2000:.1f} us ({100.0 * v / total:.1f} %)")
