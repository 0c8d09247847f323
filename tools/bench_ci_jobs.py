"""Time of one kind of workgroup of the fused factorisation (gpbo_cholinv_tiles_f64): python tools/bench_ci_jobs.py [Np]
Each line: the tiles, time per launch, and - for the update tiles - the TFLOP/s the launch delivers."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import _lib
lib = _lib.load()
GROUP = int(os.environ.get('CI_GROUP', '1'))
Np = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
S = torch.rand(Np, 2 * Np, dtype=torch.float64, device=dev) * 1e-3
S[:, :Np] += torch.eye(Np, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SMALL, BIG, BIG256 = 2, 3, 4

def run(pair, tiles, reps=20):
    T = np.array(tiles, dtype=np.int32).reshape(-1, 8)
    S0 = S.clone()
    def go(n):
        rc = lib.gpbo_cholinv_tiles_f64(C.c_void_p(S0.data_ptr()), 2 * Np, Np, C.c_void_p(info.data_ptr()), pair,
                                        T.ctypes.data_as(C.c_void_p), len(T), GROUP, n, st)
        assert rc == 0, rc
    go(2)
    S0.copy_(S); torch.cuda.synchronize()
    # the call is synchronous: difference of two repetition counts removes the upload / sync overhead
    t0 = time.perf_counter(); go(5); t1 = time.perf_counter(); go(5 + reps); t2 = time.perf_counter()
    return ((t2 - t1) - (t1 - t0)) / reps * 1e6

def far_tiles(kind, K, n):
    th = 256 if kind == BIG256 else 128
    out = []
    rr = (K + th - 1) // th * th
    while rr + th <= Np and len(out) < n:
        c = rr
        while c < Np + K and len(out) < n:
            out.append([kind, 0, K, rr, c, Np, K, 0]); c += 128
        rr += th
    return out

NBS = [int(x) for x in os.environ.get('CI_NTILES', '128,256,512').split(',')]   # CI_NTILES=1,8,32,...: how a tile's time grows
for kind, nm, th in ((BIG, "big128", 128), (BIG256, "big256", 256)):                 # with the number of tiles beside it
    for K in (128, 256, 384, 512):
        for nb in NBS:
            tl = far_tiles(kind, K, nb)
            if len(tl) < nb:
                continue
            us = run(-1, tl)
            print(f"{nm} K={K} tiles={nb}: {us:8.2f} us/launch  {nb * th * 128 * K * 2 / us / 1e6:6.1f} TFLOP/s", flush=True)
for K in (128, 256):
    tl = [[SMALL, 0, K, 64 * b, c, K + 128, K, 0] for b in (K // 64, K // 64 + 1) for c in range(64 * b, Np + K, 64)]
    print(f"small-near K={K} ({len(tl)} tiles): {run(-1, tl):8.2f} us/launch", flush=True)
for p in (0, Np // 256):
    print(f"pair p={p} ({Np // 64} workgroups): {run(p, [], reps=5):8.2f} us/launch", flush=True)
