"""One launch configuration of the fused factorisation's update tiles, repeated (for rocprofv3 passes):
python tools/ci_tile_one.py KIND K NTILES [Np] [reps]   KIND: 3 = 128 x 128, 4 = 256 x 128"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import _lib
lib = _lib.load()
GROUP = int(os.environ.get('CI_GROUP', '1'))
kind, K, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Np = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda", 0)
S = torch.rand(Np, 2 * Np, dtype=torch.float64, device=dev) * 1e-3
info = torch.zeros(1, dtype=torch.int32, device=dev)
th = 256 if kind == 4 else 128
tiles = []
rr = (K + th - 1) // th * th
while rr + th <= Np and len(tiles) < n:
    c = rr
    while c < Np + K and len(tiles) < n:
        tiles.append([kind, 0, K, rr, c, Np, K, 0]); c += 128
    rr += th
T = np.array(tiles, dtype=np.int32)
rc = lib.gpbo_cholinv_tiles_f64(C.c_void_p(S.data_ptr()), 2 * Np, Np, C.c_void_p(info.data_ptr()), -1, T.ctypes.data_as(C.c_void_p),
                                len(T), GROUP, reps, C.c_void_p(torch.cuda.current_stream().cuda_stream))
assert rc == 0
print("ran", len(T), "tiles x", reps)
