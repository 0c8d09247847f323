"""The drop-in class bound through the host-pointer C entry points only: NumPy + ctypes, no PyTorch.

This is the ctypes stub a maintainer of the reference would write against include/gpbo.h
(gpbo_select_next_host_f64, gpbo_nlml_grid_host_f64): the arrays the reference already holds
(/root/reference/select_parameters.py:149-153, 285-289) go in as host pointers, `mean_func`, `cov_func`,
`acq_func_eval` and the selected index come back.  Every call allocates and frees its device buffers inside the
library, which costs a few milliseconds per step against the tensor-resident `PointSelector`; the numbers are the
same (same kernels).  Not available here: candidate sharding over several GPUs, the incremental factorisation,
fp32 / int8 screening - those need the device-pointer API behind `PointSelector`.
A process that uses both this binding and the PyTorch-based classes must `import torch` before its first call here
(bringing PyTorch's GPU context up after this library has initialised HIP was seen to dead-lock now and then; the
PyTorch-based classes refuse that order with a clear error).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .gp_device import JITTER_ASSEMBLY, JITTER_KERNEL
from .point_selector import PointSelector


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def nlml_grid(X, y, ls_cells, jitter: float = JITTER_KERNEL, lib=None, likelihood: str = "reference") -> np.ndarray:
    """-log marginal likelihood of every row of ls_cells [G x d]  (point_selector.py:111-156): float32 with the
    reference's det underflow (likelihood="reference") or fp64 with log det from the factor ("logdet")."""
    if likelihood not in ("reference", "logdet"):
        raise ValueError(f"likelihood must be 'reference' or 'logdet', got {likelihood!r}")
    lib = lib or _lib.load()
    X, y = _f64(X), _f64(y).reshape(-1)
    N, d = X.shape
    cells = _f64(np.asarray(ls_cells, dtype=np.float64).reshape(-1, d))
    logdet = likelihood == "logdet"
    out = np.empty(len(cells), dtype=np.float64 if logdet else np.float32)
    _lib.note_hip_use()
    fn = lib.gpbo_nlml_grid_logdet_host_f64 if logdet else lib.gpbo_nlml_grid_host_f64
    _lib.check(fn(_ptr(X), _ptr(y), N, d, _ptr(cells), len(cells), float(jitter), _ptr(out)),
               "gpbo_nlml_grid_logdet_host_f64" if logdet else "gpbo_nlml_grid_host_f64")
    return out


def select_next(X, y, ls, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best=None, xi: float = 0.0,
                dense: bool = True, want_cov_meas: bool = False, chunk: int = 0, lib=None) -> dict:
    """One surrogate step on host arrays.  Returns dict(best_val, best_idx, nan_count, info, mu, sigma, acq, cov_meas)."""
    lib = lib or _lib.load()
    X, y, Xs = _f64(X), _f64(y).reshape(-1), _f64(Xs)
    ls = _f64(np.asarray(ls, dtype=np.float64).reshape(-1))
    N, d = X.shape
    M = Xs.shape[0]
    if Xs.shape[1] != d or ls.size != d or y.size != N:
        raise ValueError("shapes: X (N, d), y (N,), ls (d,), Xs (M, d)")
    if acquisition == "lcb":
        kind, p0, p1 = _lib.ACQ_LCB, float(explore), 0.0
    elif acquisition == "ei":
        if f_best is None:
            raise ValueError("EI needs f_best (the incumbent minimum)")
        kind, p0, p1 = _lib.ACQ_EI, float(f_best), float(xi)
    else:
        raise ValueError(f"unknown acquisition {acquisition!r}")
    mu = np.empty(M) if dense else None
    sigma = np.empty(M) if dense else None
    acq = np.empty(M) if dense else None
    cov = np.empty((N, N)) if want_cov_meas else None
    res = (C.c_int64 * 4)()
    info = C.c_int32(0)
    diag_add = JITTER_KERNEL if Xs.shape == X.shape else 0.0          # point_selector.py:173 shape-coincidence quirk
    _lib.note_hip_use()
    st = lib.gpbo_select_next_host_f64(_ptr(X), _ptr(y), N, d, _ptr(ls), JITTER_KERNEL, JITTER_ASSEMBLY, _ptr(Xs), M,
                                       kind, p0, p1, diag_add, int(chunk), _ptr(mu), _ptr(sigma), _ptr(acq), _ptr(cov),
                                       C.cast(res, C.c_void_p), C.cast(C.pointer(info), C.c_void_p))
    _lib.check(st, "gpbo_select_next_host_f64")
    best_val = float(np.frombuffer(res, dtype=np.float64, count=1)[0])
    return dict(best_val=best_val, best_idx=int(res[1]), nan_count=int(res[2]), info=int(info.value), mu=mu, sigma=sigma,
                acq=acq, cov_meas=cov)


class _GridOnly:
    """What PointSelector.tune_kernel needs from its surrogate object: the likelihood grid."""

    def __init__(self, lib):
        self.lib = lib

    def nlml_grid(self, X, y, ls_cells, jitter: float = JITTER_KERNEL, likelihood: str = "reference"):
        return nlml_grid(X, y, ls_cells, jitter, self.lib, likelihood)


class PointSelectorHost(PointSelector):
    """`PointSelector` with the same attribute protocol (point_selector.py:13-207), on the host-pointer entry points."""

    def __init__(self, verbose: bool = False, chunk: int = 0, likelihood: str = "reference"):
        super().__init__(device=None, verbose=verbose, shard_candidates=False, likelihood=likelihood)
        self.lib = _lib.load()
        self._gp = _GridOnly(self.lib)
        self._chunk = int(chunk)
        self._inputs = None

    def _score(self, acquisition, want_cov_meas=False, **kw):
        X, y, ls, Xs = self._inputs
        r = select_next(X, y, ls, Xs, acquisition=acquisition, dense=True, want_cov_meas=want_cov_meas,
                        chunk=self._chunk, lib=self.lib, **kw)
        if r["info"] != 0:
            raise np.linalg.LinAlgError(
                f"covariance matrix is not positive definite (pivot {r['info']} of {len(X)}); "
                "the reference's np.linalg.inv would raise or return garbage here")
        return r

    def update_surrogate(self):
        """point_selector.py:42-102."""
        self.measured_pts = np.array(self.measured_pts)
        self.measured_vals = np.array(self.measured_vals)
        X = np.asarray(self.measured_pts, dtype=np.float64)
        y = np.asarray(self.measured_vals, dtype=np.float64)
        Xs = np.asarray(self.predicted_pts, dtype=np.float64)
        ls = self._select_kernel_params(X)
        self._inputs = (X, y, ls, Xs)
        r = self._score("lcb", want_cov_meas=True, explore=4.0)
        fd = [int(v) for v in self.feature_domain]
        self.cov_meas = r["cov_meas"]
        self.mean_func = r["mu"].reshape(fd)                              # :97
        self.cov_func = r["sigma"].reshape(fd)                            # :98 (a standard deviation)
        self._cached = {("lcb", 4.0, 0.0): (r["acq"].reshape(fd), (r["best_val"], r["best_idx"], r["nan_count"]))}
        self.cov_pred = self.cov_meas_pred = None                         # not materialised on this route
        self.last_update = "factorise"
        self.measured_pts = self.measured_pts.tolist()                    # :101-102
        self.measured_vals = self.measured_vals.tolist()

    def _finish(self, key, kind, **kw):
        fd = [int(v) for v in self.feature_domain]
        if key not in self._cached:   # another acquisition on the same data: one more pass through the kernels
            r = self._score(kind, **kw)
            self._cached[key] = (r["acq"].reshape(fd), (r["best_val"], r["best_idx"], r["nan_count"]))
        acq, (best_val, best_idx, nan_count) = self._cached[key]
        self.acq_func_eval = acq
        if nan_count > 0 or best_idx >= int(np.prod(fd)) or best_idx < 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0 (acquisition contains NaN)")
        return np.array(np.unravel_index(best_idx, fd), dtype=np.int64)

    def q_expected_improvement(self, n_samples=512, seed=7, xi=0.0):
        """q = 8 Monte-Carlo Expected Improvement on the host-pointer route (same definition and return value as
        PointSelector.q_expected_improvement: the (8, ndim) multi-indices of the first batch with the largest qEI)."""
        if self._inputs is None:
            raise RuntimeError("call update_surrogate() first")
        X, y, ls, Xs = self._inputs
        fd = [int(v) for v in self.feature_domain]
        M = int(np.prod(fd))
        if M % 8:
            raise ValueError("q_expected_improvement needs a candidate count that is a multiple of 8")
        Z = _f64(np.random.default_rng(seed).standard_normal((int(n_samples), 8)))
        X, y, Xs, ls = _f64(X), _f64(y).reshape(-1), _f64(Xs), _f64(np.asarray(ls, dtype=np.float64).reshape(-1))
        qei = np.empty(M // 8)
        res = (C.c_int64 * 4)()
        info = C.c_int32(0)
        _lib.note_hip_use()
        st = self.lib.gpbo_select_qei_host_f64(_ptr(X), _ptr(y), X.shape[0], X.shape[1], _ptr(ls), JITTER_KERNEL,
                                               JITTER_ASSEMBLY, _ptr(Xs), M, float(np.min(y)), float(xi), _ptr(Z),
                                               int(n_samples), self._chunk, _ptr(qei), C.cast(res, C.c_void_p),
                                               C.cast(C.pointer(info), C.c_void_p))
        _lib.check(st, "gpbo_select_qei_host_f64")
        if info.value != 0:
            raise np.linalg.LinAlgError(f"covariance matrix is not positive definite (pivot {info.value} of {len(X)})")
        self.acq_func_eval = qei
        if int(res[2]) > 0 or int(res[1]) >= M // 8 or int(res[1]) < 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0 (acquisition contains NaN)")
        flat = int(res[1]) * 8 + np.arange(8)
        return np.stack(np.unravel_index(flat, fd), axis=1).astype(np.int64)
