"""The fused Cholesky + inverse factor (gpbo_cholinv_f64, csrc/cholinv.hip) on the MI355X against LAPACK and against the
CPU execution of the same launch plan (tests/test_cholinv_plan_cpu.py), launch by launch.
Replaces np.linalg.inv(cov_meas) of /root/reference/point_selector.py:89."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bayesian_optimisation_amd import _lib  # noqa: E402
from cholinv_sim import get_plan, run_plan, spd  # noqa: E402


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"

    class Env:
        pass

    e = Env()
    e.torch, e.lib, e.dev = torch, _lib.load(), torch.device("cuda", 0)
    e.stream = lambda: C.c_void_p(torch.cuda.current_stream(e.dev).cuda_stream)
    e.p = lambda t: C.c_void_p(t.data_ptr())
    return e


def run_gpu(env, A, opt=None):
    t = env.torch
    Np = A.shape[0]
    S = np.zeros((Np, 2 * Np))
    S[:, :Np] = A
    dS = t.from_numpy(S).to(env.dev)
    info = t.full((1,), -7, dtype=t.int32, device=env.dev)
    o = (C.c_int32 * 5)(*(opt or [0, 0, 0, 0, 0]))
    st = env.lib.gpbo_cholinv_f64(env.p(dS), 2 * Np, Np, env.p(info), C.cast(o, C.c_void_p), env.stream())
    assert st == 0
    t.cuda.synchronize()
    return dS.cpu().numpy(), int(info.item())


@pytest.mark.parametrize("Np,opt", [(256, [0, 0, 0, 0]), (512, [0, 1, 0, 0]), (640, [0, 0, 0, 0])])
def test_every_launch_matches_the_cpu_execution_of_the_plan(env, Np, opt):
    A = spd(Np, 100 + Np)
    plan = get_plan(Np, opt)
    S = np.zeros((Np, 2 * Np))
    S[:, :Np] = A
    for n in range(1, len(plan) + 1):
        run_plan(S, Np, plan[n - 1:n])
        got, info = run_gpu(env, A, opt + [n])
        assert info == 0
        # live data only: blocks the CPU execution has written so far, plus everything still untouched, must agree;
        # the dead lower block triangle of the left half is never written by either
        err = np.max(np.abs(got - S))
        assert err <= 1e-9, f"launch {n} of {len(plan)} ({plan[n - 1][:, 0]}): max |diff| {err}"


@pytest.mark.parametrize("Np,opt", [(128, None), (384, None), (1024, None), (1024, [0, 50, 0, 0]), (2176, None),
                                    (2176, [0, 1, 0, 0]), (4096, None), (4224, None)])
def test_inverse_factor_vs_lapack(env, Np, opt):
    A = spd(Np, Np)
    got, info = run_gpu(env, A, (opt + [0]) if opt else None)
    assert info == 0
    W = got[:, Np:]
    L = np.linalg.cholesky(A)
    import scipy.linalg as sla

    Winv = sla.solve_triangular(L, np.eye(Np), lower=True)
    assert np.array_equal(np.triu(W, 1), np.zeros_like(W))
    assert np.max(np.abs(W - Winv)) <= 1e-9 * np.abs(Winv).max()
    assert np.max(np.abs((W.T @ W) @ A - np.eye(Np))) <= 1e-6


def test_bad_pivot_is_reported(env):
    Np = 384
    A = spd(Np, 3)
    A[200, 200] = -1.0  # the Schur complement at column 201 cannot be positive
    _, info = run_gpu(env, A)
    assert info == 201
