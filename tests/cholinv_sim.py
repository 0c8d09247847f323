"""CPU execution of the launch plan of the fused Cholesky + inverse factor (csrc/cholinv_plan.h) - shared by
tests/test_cholinv_plan_cpu.py (no GPU) and tests/test_gpu_cholinv.py (launch-by-launch comparison on the MI355X).
Test infrastructure only."""
import ctypes as C

import numpy as np

from bayesian_optimisation_amd import _lib

NONE, PAIR, SMALL, BIG = 0, 1, 2, 3


def get_plan(Np, opt=None):
    lib = _lib.load()
    o = (C.c_int32 * 4)(*(opt or [0, 0, 0, 0]))
    words = lib.gpbo_cholinv_plan(Np, C.cast(o, C.c_void_p), None, 0)
    assert words > 0 and words % 27 == 0
    buf = (C.c_int32 * words)()
    assert lib.gpbo_cholinv_plan(Np, C.cast(o, C.c_void_p), C.cast(buf, C.c_void_p), words) == words
    return np.frombuffer(buf, dtype=np.int32).reshape(-1, 3, 9).copy()


def tile_of(kind, Np, r0, wlim, t):
    lib = _lib.load()
    r, c = C.c_int32(), C.c_int32()
    assert lib.gpbo_cholinv_tile(kind, Np, r0, wlim, t, C.byref(r), C.byref(c)) == 0
    return r.value, c.value


class Tracker:
    """64 x 64 block read / write sets of the tiles of one launch."""

    def __init__(self, Np):
        self.nbr, self.nbc = Np // 64, 2 * Np // 64
        self.writer = -np.ones((self.nbr, self.nbc), dtype=np.int64)
        self.reads = []

    def write(self, tile_id, r, c):
        assert self.writer[r // 64, c // 64] == -1, f"two tiles of one launch write block ({r},{c})"
        self.writer[r // 64, c // 64] = tile_id

    def read(self, tile_id, r0, r1, c0, c1):
        self.reads.append((tile_id, r0 // 64, (r1 + 63) // 64, c0 // 64, (c1 + 63) // 64))

    def check(self):
        for tid, a, b, c, d in self.reads:
            w = self.writer[a:b, c:d]
            bad = (w != -1) & (w != tid)
            assert not bad.any(), f"tile {tid} reads a block that tile {w[bad][0]} of the same launch writes"


def run_plan(S, Np, plan, reads_tile_wide=True):
    """S: [Np x 2Np] = [A | 0].  Executes the plan in place.  Writes of a launch are applied after all its tiles have been
    computed from the state before the launch (what concurrent workgroups may or may not see is excluded by the tracker)."""
    tile_id = 0
    for launch in plan:
        tr = Tracker(Np)
        pending = []
        for kind, nblk, j, k0, K, r0, r1, wlim, t0 in launch:
            if kind == NONE or nblk == 0:
                continue
            if kind == PAIR:
                r0p = 128 * j
                D = S[r0p:r0p + 128, r0p:r0p + 128]
                D = np.tril(D) + np.tril(D, -1).T
                D[:64, 64:] = S[r0p:r0p + 64, r0p + 64:r0p + 128]  # the kernel reads A12 from the upper block
                D[64:, :64] = D[:64, 64:].T
                L = np.linalg.cholesky(D)
                Linv = np.tril(np.linalg.solve(L, np.eye(128)))
                nA = (Np - (r0p + 128)) // 64
                assert nblk == Np // 64
                for pt in range(nblk):
                    tile_id += 1
                    c0 = r0p + 128 + 64 * pt if pt < nA else Np + 64 * (pt - nA)
                    assert c0 + 64 <= 2 * Np
                    tr.read(tile_id, r0p, r0p + 128, r0p, r0p + 128)
                    ident = c0 >= Np and c0 - Np >= r0p
                    if ident:
                        X = np.zeros((128, 64))
                        blk = (c0 - Np - r0p) // 64
                        X[64 * blk:64 * blk + 64, :] = np.eye(64)
                    else:
                        X = S[r0p:r0p + 128, c0:c0 + 64]
                        tr.read(tile_id, r0p, r0p + 128, c0, c0 + 64)
                    Y = Linv @ X
                    for h in range(2):
                        pending.append((r0p + 64 * h, c0, Y[64 * h:64 * h + 64]))
                        tr.write(tile_id, r0p + 64 * h, c0)
            elif kind == SMALL:
                for t in range(nblk):
                    tile_id += 1
                    row0, col0 = tile_of(SMALL, Np, r0, wlim, t0 + t)
                    assert r0 <= row0 < r1 and row0 <= col0 and col0 + 64 <= Np + wlim
                    A = S[k0:k0 + K, row0:row0 + 64]
                    B = S[k0:k0 + K, col0:col0 + 64]
                    tr.read(tile_id, k0, k0 + K, row0, row0 + 64)
                    tr.read(tile_id, k0, k0 + K, col0, col0 + 64)
                    pending.append((row0, col0, S[row0:row0 + 64, col0:col0 + 64] - A.T @ B))
                    tr.write(tile_id, row0, col0)
            elif kind == BIG:
                for t in range(nblk):
                    tile_id += 1
                    row0, col0 = tile_of(BIG, Np, r0, wlim, t0 + t)
                    assert r0 <= row0 < r1 and col0 % 128 == 0 and col0 + 128 <= 2 * Np and row0 + 128 <= 2 * Np
                    A = S[k0:k0 + K, row0:row0 + 128]      # may run into the W half: in bounds, masked below
                    B = S[k0:k0 + K, col0:col0 + 128]
                    tr.read(tile_id, k0, k0 + K, row0, row0 + 128)
                    tr.read(tile_id, k0, k0 + K, col0, col0 + 128)
                    P = A.T @ B
                    for rb in range(2):
                        rr = row0 + 64 * rb
                        if rr >= r1:
                            continue
                        for cb in range(2):
                            cc = col0 + 64 * cb
                            live = (cc >= rr) if cc < Np else (cc < Np + wlim)
                            if live:
                                pending.append((rr, cc, S[rr:rr + 64, cc:cc + 64] - P[64 * rb:64 * rb + 64, 64 * cb:64 * cb + 64]))
                                tr.write(tile_id, rr, cc)
            else:
                raise AssertionError(kind)
        tr.check()
        for r, c, v in pending:
            S[r:r + 64, c:c + 64] = v
    return S


def spd(Np, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((Np, 4))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2 / 0.3 ** 2) + 1.01e-4 * np.eye(Np)
