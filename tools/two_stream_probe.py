"""Does a critical-path launch keep its latency while long update tiles run on ANOTHER stream?  (the question a
two-stream factorisation - eliminations on one stream, far updates on a second - stands or falls with.)

The critical stream repeats PAIR(0)-only launches (Np/64 workgroups, ~35 us each, dependent by stream order); the bulk stream
repeats launches of 128 x 128 x K update tiles.  Both run through gpbo_cholinv_tiles_f64 (synchronous per call, so each
runs in its own host thread), on separate matrices.  Reported: microseconds per launch of each, alone and together, with
the critical stream at high priority and with the bulk stream confined to a subset of the compute units (CU mask).
usage: python tools/two_stream_probe.py [Np] [K]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bayesian_optimisation_amd import _lib

lib = _lib.load()
Np = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

torch.zeros(1, device=dev)   # PyTorch's HIP runtime is up
hip = C.CDLL(_lib.mapped_hip_runtimes()[0])   # the copy PyTorch already mapped: same handle, one runtime

# critical: a well-conditioned stacked matrix, so that repeating PAIR(0) on it stays finite
A = 2.0 * torch.eye(Np, dtype=torch.float64, device=dev)
A[:128, :] += 0.01
A[:, :128] += 0.01
S1 = torch.zeros(Np, 2 * Np, dtype=torch.float64, device=dev)
S1[:, :Np] = A
S2 = torch.rand(Np, 2 * Np, dtype=torch.float64, device=dev) * 1e-3
info = torch.zeros(2, dtype=torch.int32, device=dev)


def tiles(n):
    out, rr = [], (K + 127) // 128 * 128
    while rr + 128 <= Np and len(out) < n:
        c = rr
        while c < Np + K and len(out) < n:
            out.append([3, 0, K, rr, c, Np, K, 0])
            c += 128
        rr += 128
    assert len(out) == n, (len(out), n)
    return np.array(out, dtype=np.int32)


def crit(stream, reps, res):
    t0 = time.perf_counter()
    rc = lib.gpbo_cholinv_tiles_f64(C.c_void_p(S1.data_ptr()), 2 * Np, Np, C.c_void_p(info.data_ptr()), 0, None, 0, 1, reps, C.c_void_p(stream))
    res["crit"] = (time.perf_counter() - t0) / reps * 1e6
    assert rc == 0, rc


def bulk(stream, T, reps, res):
    t0 = time.perf_counter()
    rc = lib.gpbo_cholinv_tiles_f64(C.c_void_p(S2.data_ptr()), 2 * Np, Np, C.c_void_p(info.data_ptr() + 4), -1, T.ctypes.data_as(C.c_void_p),
                                    len(T), 1, reps, C.c_void_p(stream))
    res["bulk"] = (time.perf_counter() - t0) / reps * 1e6
    assert rc == 0, rc


def masked_stream(keep_of_32):
    """A stream whose kernels may use only `keep_of_32` of every 32 compute units (spread evenly whatever the numbering)."""
    words = (C.c_uint32 * 8)(*([(1 << keep_of_32) - 1] * 8))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    return s.value if rc == 0 else None


plain_a, plain_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
lo_pri, hi_pri = torch.cuda.Stream(dev, priority=0), torch.cuda.Stream(dev, priority=-1)
m24, m28 = masked_stream(24), masked_stream(28)
T1, T2 = tiles(256), tiles(512)
warm = {}
crit(plain_a.cuda_stream, 20, warm)
bulk(plain_b.cuda_stream, T1, 5, warm)
torch.cuda.synchronize()

r = {}
crit(plain_a.cuda_stream, 300, r)
print(f"critical alone: {r['crit']:.1f} us per PAIR launch ({Np // 64} workgroups)")
for name, T in (("256", T1), ("512", T2)):
    bulk(plain_b.cuda_stream, T, 40, r)
    print(f"bulk alone, {name} tiles of rank {K}: {r['bulk']:.1f} us per launch")
for name, s in (("24 of 32 CUs", m24), ("28 of 32 CUs", m28)):
    if s:
        bulk(s, T1, 40, r)
        print(f"bulk alone on {name}, 256 tiles: {r['bulk']:.1f} us per launch")
    else:
        print(f"CU-masked stream ({name}): hipExtStreamCreateWithCUMask failed")


def together(label, cs, bs, T):
    res = {}
    tb = threading.Thread(target=bulk, args=(bs, T, 600 if len(T) <= 256 else 300, res))
    tb.start()
    time.sleep(0.003)   # the bulk launches are running
    crit(cs, 400, res)   # both run for most of each other's time
    tb.join()
    print(f"{label}: critical {res['crit']:.1f} us per launch, bulk {res['bulk']:.1f} us per launch")


together("together, equal priority, bulk 256 tiles", plain_a.cuda_stream, plain_b.cuda_stream, T1)
together("together, critical stream high priority, bulk 256 tiles", hi_pri.cuda_stream, lo_pri.cuda_stream, T1)
together("together, critical stream high priority, bulk 512 tiles", hi_pri.cuda_stream, lo_pri.cuda_stream, T2)
for name, s in (("24 of 32 CUs", m24), ("28 of 32 CUs", m28)):
    if s:
        together(f"together, critical high priority, bulk confined to {name}, 256 tiles", hi_pri.cuda_stream, s, T1)
        together(f"together, critical high priority, bulk confined to {name}, 512 tiles", hi_pri.cuda_stream, s, T2)
assert int(info[0]) == 0, "PAIR(0) reported a bad pivot"
assert bool(torch.isfinite(S1[:128]).all())
