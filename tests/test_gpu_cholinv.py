"""The fused Cholesky + inverse factor (gpbo_cholinv_f64, csrc/cholinv.hip) on the MI355X against LAPACK and against the
CPU execution of the same launch plan (tests/cholinv_sim.py), launch by launch and tile kind by tile kind.
Replaces np.linalg.inv(cov_meas) of /root/reference/point_selector.py:89."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bayesian_optimisation_amd import _lib  # noqa: E402
from cholinv_sim import BIG, BIG256, SMALL, Tracker, get_plan, run_plan, run_tile, spd  # noqa: E402


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
    o = (C.c_int32 * 7)(*((list(opt or []) + [0] * 7)[:7]))
    st = env.lib.gpbo_cholinv_f64(env.p(dS), 2 * Np, Np, env.p(info), C.cast(o, C.c_void_p), env.stream())
    assert st == 0
    t.cuda.synchronize()
    return dS.cpu().numpy(), int(info.item())


@pytest.mark.parametrize("Np,opt", [(256, [0, 0, 0, 0]), (640, [1, 128, 3, 1]), (896, [2, 256, 4, 2]), (1152, [2, 256, 3, 2]), (2304, [1, 128, 3, 1])])
def test_every_launch_matches_the_cpu_execution_of_the_plan(env, Np, opt):
    A = spd(Np, 100 + Np)
    plan = get_plan(Np, opt)
    S = np.zeros((Np, 2 * Np))
    S[:, :Np] = A
    for n in range(1, len(plan[0]) + 1):
        run_plan(S, Np, plan, first=n - 1, count=1)
        got, info = run_gpu(env, A, opt + [n])
        assert info == 0
        err = np.max(np.abs(got - S))
        tol = 3e-11 * max(1.0, np.abs(S).max())  # entries of inv(L) reach ~1 / sqrt(jitter) = 100
        assert err <= tol, f"launch {n} of {len(plan[0])} ({plan[0][n - 1]}): max |diff| {err} > {tol}"


@pytest.mark.parametrize("kind,K", [(SMALL, 32), (SMALL, 128), (SMALL, 352), (-SMALL, 128), (-SMALL, 224), (BIG, 128), (BIG, 400 // 16 * 16), (BIG256, 128),
                                    (BIG256, 512)])
def test_one_tile_kind_against_numpy(env, kind, K):
    """A handful of tiles of one kind with sources [k0, k0 + K): live masks, ragged last row tile, W-side columns."""
    K = K // 32 * 32
    half = kind < 0          # 64 x 32 tiles
    kind = abs(kind)
    t = env.torch
    Np = 1152
    rng = np.random.default_rng(K + kind)
    S = rng.standard_normal((Np, 2 * Np))
    k0 = 64
    r0 = 640 if kind != SMALL else 576
    H = {SMALL: 64, BIG: 128, BIG256: 256}[kind]
    wlim = (k0 + K + 63) // 64 * 64
    tiles = []
    if kind == SMALL:
        for c in sorted({r0, r0 + 64, Np - 64, Np, Np + wlim - 64}):
            if half:
                tiles.append([kind, k0, K, r0, c, r0 + 64, wlim, 32])
                tiles.append([kind, k0, K, r0, c + 32, r0 + 64, wlim, 32])
            else:
                tiles.append([kind, k0, K, r0, c, r0 + 64, wlim, 0])
    else:
        for row0 in (r0, Np - 128):  # the second one is ragged for 256-row tiles
            for c in sorted({row0, min(row0 + 128, Np - 128), Np - 128, Np, Np + (wlim + 127) // 128 * 128 - 128}):
                tiles.append([kind, k0, K, row0, c, Np, wlim, 0])
    T = np.array(tiles, dtype=np.int32)
    ref = S.copy()
    tr, pending = Tracker(Np), []
    for i, tl in enumerate(T):
        run_tile(ref, Np, tl, tr, i + 1, pending)
    for r, c, v in pending:
        ref[r:r + 64, c:c + v.shape[1]] = v
    dS = t.from_numpy(S).to(env.dev)
    info = t.zeros(1, dtype=t.int32, device=env.dev)
    st = env.lib.gpbo_cholinv_tiles_f64(env.p(dS), 2 * Np, Np, env.p(info), -1, T.ctypes.data_as(C.c_void_p), len(T), 2, 1, env.stream())
    assert st == 0
    got = dS.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-11 * K


@pytest.mark.parametrize("Np,opt", [(128, None), (384, None), (1024, None), (1024, [1, 128, 3, 1]), (2176, None),
                                    (2176, [3, 512, 4, 2]), (4096, None), (4224, [2, 256, 4, 3]),
                                    (6144, None), (8192, None)])   # from 6144 up: far rank 384, 64-wide NEAR tiles
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


@pytest.mark.parametrize("N", [16384, 32768 + 200])
def test_factorise_at_sizes_no_other_test_reaches(N):
    """N = 16,384: the fused sweep with 128 pairs and far rank 384 (LAPACK on the host would take minutes: the inverse factor is
    checked by probing U^T K U = I with random vectors on the device).  N = 32,968: beyond the fused plan's cap (Np <= 32,768)
    gpbo_factorise_f64 must fall back to the two-pass chain (it used to return an argument error), same check."""
    import torch

    from bayesian_optimisation_amd import DeviceGP

    rng = np.random.default_rng(N)
    d = 3
    X = rng.uniform(0, 1, (N, d))
    ls = np.full(d, 0.05)          # short length scales: K is far from singular at any N, the check is about the kernels
    y = rng.standard_normal(N)
    gp = DeviceGP().factorise(X, y, ls)
    assert int(gp.info.item()) == 0 and gp.Np == (N + 127) // 128 * 128
    K, U = gp.K[:N, :N], gp.U[:N, :N]
    assert torch.equal(torch.tril(U, -1), torch.zeros_like(U))
    g = torch.Generator(device="cpu").manual_seed(N)
    Z = torch.randn(N, 4, dtype=torch.float64, generator=g).to(gp.device)
    R = U.T @ (K @ (U @ Z)) - Z                      # (U^T K U - I) Z
    assert float(R.abs().max()) <= 1e-9 * float(Z.abs().max()) * max(1.0, float(U.abs().max()))
    alpha = gp.alpha[:N]
    yd = gp._dev(y)
    assert float((K @ alpha - yd).abs().max()) <= 1e-8 * max(1.0, float(alpha.abs().max()))
    del K, U, Z, R
    # and the posterior of a few candidates against the oracle's formula evaluated with torch (no host factorisation)
    Xs = rng.uniform(0, 1, (2000, d))
    r = gp.score(Xs, dense=True)
    Xd, Xsd = gp._dev(X), gp._dev(Xs)
    ks = torch.exp(-0.5 * (((Xsd[:, None, :] - Xd[None, :, :]) / gp._dev(ls)) ** 2).sum(-1))
    mu = ks @ alpha
    assert float((r.mu - mu).abs().max()) <= 1e-9 * max(1.0, float(np.abs(y).max()))
    v = gp.U[:N, :N].T @ ks.T
    var = 1.000101 - (v * v).sum(0)
    assert float((r.sigma - var.abs().sqrt()).abs().max()) <= 1e-8
