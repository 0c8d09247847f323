"""The launch plan of the fused Cholesky + inverse factor (csrc/cholinv_plan.h), executed on the CPU.

libgpbo.so exports its plan as data (gpbo_cholinv_plan: host code, no GPU).  This test runs every workgroup of every
launch with NumPy, tile by tile, with the semantics the HIP workgroups implement (csrc/cholinv.hip), and checks
  * the result: W = inv(L) of LAPACK's Cholesky factor (what /root/reference/point_selector.py:89 gets from np.linalg.inv
    is W^T W), for several sizes, windows, far ranks and tile choices;
  * the schedule: inside one launch no tile reads a 64 x 64 block that another tile of the launch writes, and no two
    tiles write the same block - dependencies are carried by launch order alone, so this is what makes the GPU run
    race-free.
"""
import ctypes as C

import numpy as np
import pytest

from bayesian_optimisation_amd import _lib
from cholinv_sim import BIG, BIG256, SMALL, get_plan, run_plan, spd

OPTS = [(128, None), (256, None), (384, None), (512, None), (640, [1, 128, 3, 1]), (896, None), (896, [2, 256, 4, 2]),
        (1024, [3, 384, 3, 2]), (1152, [1, 256, 4, 1]), (1536, [2, 512, 4, 3]), (1664, None), (1664, [3, 256, 4, 3]), (2048, [1, 128, 3, 1, 0, 2]), (768, [1, 256, 3, 3, 0, 0, 64]),
        (1536, [1, 384, 3, 3, 0, 0, 64])]   # the last: the options N >= 7680 runs with, at a size the CPU executes quickly


@pytest.mark.parametrize("Np,opt", OPTS)
def test_plan_executed_on_the_cpu_gives_the_inverse_factor(Np, opt):
    A = spd(Np, Np)
    S = np.zeros((Np, 2 * Np))
    S[:, :Np] = A
    plan = get_plan(Np, opt)
    run_plan(S, Np, plan)
    W = S[:, Np:]
    L = np.linalg.cholesky(A)
    Winv = np.linalg.solve(L, np.eye(Np))
    assert np.array_equal(np.triu(W, 1), np.zeros_like(W))
    scale = np.abs(Winv).max()
    assert np.max(np.abs(W - Winv)) <= 1e-9 * scale
    # the upper block triangle of the left half holds L^T outside the pairs' 128 x 128 diagonal blocks
    R = S[:, :Np]
    for b in range(Np // 128 - 1):
        assert np.max(np.abs(R[128 * b:128 * b + 128, 128 * b + 128:] - L.T[128 * b:128 * b + 128, 128 * b + 128:])) <= 1e-9
    # and the product is the inverse the reference computes (point_selector.py:89)
    assert np.max(np.abs(W.T @ W @ A - np.eye(Np))) <= 1e-6


@pytest.mark.parametrize("Np", [1024, 4096, 8192])
def test_plan_shape_at_the_benchmark_sizes(Np):
    """Every pair of block rows gets exactly one PAIR launch, preceded (but for the first) by a NEAR launch of rank 128;
    every tile is well formed; the far updates have rank 256; workgroup counts stay inside one grid dimension."""
    L, T = get_plan(Np)
    pairs = [int(l[0]) for l in L if l[1] > 0]
    assert pairs == list(range(Np // 128))
    assert (L[:, 1] + L[:, 3] > 0).all() and (L[:, 1] + L[:, 3] < 2 ** 20).all()
    assert int(L[-1, 2] + L[-1, 3]) == len(T)
    assert set(np.unique(T[:, 0])) <= {SMALL, BIG, BIG256}
    assert (T[:, 2] >= 128).all() and (T[:, 2] % 128 == 0).all() and (T[:, 1] + T[:, 2] <= T[:, 3]).all()
    near = [T[l[2]:l[2] + l[3]] for l in L if l[1] == 0]
    assert len(near) == Np // 128 - 1 and all((n[:, 0] == SMALL).all() and (n[:, 2] == 128).all() for n in near)
    flops = (T[:, 2].astype(np.int64) * np.where(T[:, 0] == SMALL, 64 * 64, np.where(T[:, 0] == BIG, 128 * 128, 256 * 128))).sum()
    big = T[T[:, 0] != SMALL]
    far_share = (big[big[:, 2] >= 256][:, 2].astype(np.int64) * 128 * 128).sum() / max(flops, 1)
    if Np >= 4096:
        assert far_share > 0.7, far_share  # most of the work moves 16 bytes of target per >= 512 flop
    lib = _lib.load()
    n1, n2 = C.c_int64(0), C.c_int64(0)
    assert lib.gpbo_cholinv_plan(Np + 64, None, C.byref(n1), C.byref(n2), None, None) == -1
    bad = (C.c_int32 * 7)(0, 200, 0, 0, 0, 0, 0)  # far rank not a multiple of 128
    assert lib.gpbo_cholinv_plan(Np, C.cast(bad, C.c_void_p), C.byref(n1), C.byref(n2), None, None) == -1
