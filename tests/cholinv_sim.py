"""CPU execution of the launch plan of the fused Cholesky + inverse factor (csrc/cholinv_plan.h) - shared by
tests/test_cholinv_plan_cpu.py (no GPU) and tests/test_gpu_cholinv.py (launch-by-launch comparison on the MI355X).
Test infrastructure only."""
import ctypes as C

import numpy as np

from bayesian_optimisation_amd import _lib

SMALL, BIG, BIG256 = 2, 3, 4
TILE_ROWS = {SMALL: 64, BIG: 128, BIG256: 256}


def get_plan(Np, opt=None):
    """(launches [n x 5]: pair, npair, tile0, ntile, tiles per workgroup;  tiles [m x 8]: kind, k0, K, row0, col0, r1, wlim, 0)"""
    lib = _lib.load()
    o = (C.c_int32 * 7)(*((list(opt or []) + [0] * 7)[:7]))
    nl, nt = C.c_int64(0), C.c_int64(0)
    assert lib.gpbo_cholinv_plan(Np, C.cast(o, C.c_void_p), C.byref(nl), C.byref(nt), None, None) == 0
    L = np.zeros((nl.value, 5), dtype=np.int32)
    T = np.zeros((max(nt.value, 1), 8), dtype=np.int32)
    assert lib.gpbo_cholinv_plan(Np, C.cast(o, C.c_void_p), C.byref(nl), C.byref(nt), L.ctypes.data_as(C.c_void_p),
                                 T.ctypes.data_as(C.c_void_p)) == 0
    return L, T[:nt.value]


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


def run_tile(S, Np, tile, tr, tile_id, pending):
    """The semantics of one update workgroup (csrc/cholinv.hip: upd_small / upd_big)."""
    kind, k0, K, row0, col0, r1, wlim, w = (int(v) for v in tile)
    H = TILE_ROWS[kind]
    Wd = (32 if w == 32 else 64) if kind == SMALL else 128
    assert K >= (32 if kind == SMALL else 128) and K % 32 == 0 and k0 + K <= row0 and col0 % Wd == 0
    assert col0 + Wd <= 2 * Np and row0 + H <= 2 * Np and r1 <= Np
    A = S[k0:k0 + K, row0:row0 + H]      # may run into the W half: in bounds, masked below
    B = S[k0:k0 + K, col0:col0 + Wd]
    tr.read(tile_id, k0, k0 + K, row0, row0 + H)
    tr.read(tile_id, k0, k0 + K, col0, col0 + Wd)
    P = A.T @ B
    for rb in range(H // 64):
        rr = row0 + 64 * rb
        if rr >= r1:
            continue
        if kind == SMALL:   # never masked; a 32-wide tile is half a 64 x 64 block (the tracker works on blocks: two
            # 32-wide tiles of one launch share a block they do not share any element of)
            pending.append((rr, col0, S[rr:rr + 64, col0:col0 + Wd] - P[64 * rb:64 * rb + 64, :]))
            if Wd == 64 or col0 % 64 == 0:
                tr.write(tile_id, rr, col0 // 64 * 64)
            continue
        for cb in range(Wd // 64):
            cc = col0 + 64 * cb
            live = (cc >= rr) if cc < Np else (cc < Np + wlim)
            if live:
                pending.append((rr, cc, S[rr:rr + 64, cc:cc + 64] - P[64 * rb:64 * rb + 64, 64 * cb:64 * cb + 64]))
                tr.write(tile_id, rr, cc)


def run_pair(S, Np, p, npair, tr, tile_id, pending):
    """The semantics of the PAIR workgroups of pair p (csrc/cholinv.hip: pair_body)."""
    r0p = 128 * p
    D = S[r0p:r0p + 128, r0p:r0p + 128]
    D = np.tril(D) + np.tril(D, -1).T
    D[:64, 64:] = S[r0p:r0p + 64, r0p + 64:r0p + 128]  # the kernel reads A12 from the upper block
    D[64:, :64] = D[:64, 64:].T
    L = np.linalg.cholesky(D)
    Linv = np.tril(np.linalg.solve(L, np.eye(128)))
    nA = (Np - (r0p + 128)) // 64
    assert npair == Np // 64
    for pt in range(npair):
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
    return tile_id


def run_plan(S, Np, plan, first=0, count=None):
    """S: [Np x 2Np] = [A | 0].  Executes launches [first, first + count) of the plan in place.  Writes of a launch are
    applied after all its tiles have been computed from the state before the launch (what concurrent workgroups may or
    may not see is excluded by the tracker)."""
    L, T = plan
    tile_id = 0
    last = len(L) if count is None else first + count
    for pair, npair, tile0, ntile, _group in L[first:last]:
        tr = Tracker(Np)
        pending = []
        if npair > 0:
            tile_id = run_pair(S, Np, int(pair), int(npair), tr, tile_id, pending)
        for t in T[tile0:tile0 + ntile]:
            tile_id += 1
            run_tile(S, Np, t, tr, tile_id, pending)
        tr.check()
        for r, c, v in pending:
            S[r:r + 64, c:c + v.shape[1]] = v
    return S


def spd(Np, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((Np, 4))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2 / 0.3 ** 2) + 1.01e-4 * np.eye(Np)
