"""The launch plan of the fused Cholesky + inverse factor (csrc/cholinv_plan.h), executed on the CPU.

libgpbo.so exports its plan as data (gpbo_cholinv_plan / gpbo_cholinv_tile: host code, no GPU).  This test runs every job
of every launch with NumPy, tile by tile, with the semantics the HIP workgroups implement (csrc/cholinv.hip), and checks
  * the result: W = inv(L) of LAPACK's Cholesky factor (what /root/reference/point_selector.py:89 gets from np.linalg.inv
    is W^T W), for several sizes, group sizes and tile choices;
  * the schedule: inside one launch no tile reads a 64 x 64 block that another tile of the launch writes, and no two
    tiles write the same block - dependencies are carried by launch order alone, so this is what makes the GPU run
    race-free.
"""
import ctypes as C

import numpy as np
import pytest

from bayesian_optimisation_amd import _lib
from cholinv_sim import BIG, PAIR, SMALL, get_plan, run_plan, spd


@pytest.mark.parametrize("Np,opt", [(128, None), (256, None), (384, None), (512, None), (640, [0, 1, 0, 0]), (896, None),
                                    (1024, [0, 20, 0, 0])])
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
    """Every pair of block rows gets exactly one PAIR launch; workgroup counts stay inside one grid dimension."""
    plan = get_plan(Np)
    pairs = [int(l[0][2]) for l in plan if l[0][0] == PAIR]
    assert pairs == list(range(Np // 128))
    for l in plan:
        assert 0 < int(l[:, 1].sum()) < 2 ** 31
        for kind, nblk, j, k0, K, r0, r1, wlim, t0 in l:
            if kind in (SMALL, BIG):
                assert K % 32 == 0 and 0 < K <= 512 and r0 % 64 == 0 and r1 % 64 == 0 and k0 + K <= r0 and wlim <= r0
    lib = _lib.load()
    assert lib.gpbo_cholinv_plan(Np + 64, None, None, 0) == -1
