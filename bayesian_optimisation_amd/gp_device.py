"""Device-resident GP surrogate: host orchestration of the libgpbo kernels.

PyTorch-ROCm is used here for device memory, streams and (in distributed.py) the process group -
plumbing only; every number is produced by the HIP kernels behind the C ABI (include/gpbo.h).

Mirrors the arithmetic of PointSelector.update_surrogate / lower_confidence_bound
(/root/reference/point_selector.py:76-98, 197-207) with the factorisation done once per BO step and
the candidates streamed in chunks.  There is no CPU path: without libgpbo.so or a GPU this raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib

JITTER_KERNEL = 1e-4    # point_selector.py:193
JITTER_ASSEMBLY = 1e-6  # point_selector.py:78-79
PRIOR_VAR = (1.0 + JITTER_KERNEL) + JITTER_ASSEMBLY  # diagonal of cov_pred as the reference rounds it

DEFAULT_CHUNK = 1 << 17


def _torch():
    if _lib.hip_used_before_pytorch:
        # Seen on MI355X / ROCm 7: importing PyTorch and creating its GPU context AFTER this library has already
        # initialised HIP in the process (host-pointer binding used first) dead-locks now and then.  The other order
        # is the one every tensor-resident path takes and has never hung - so that is the only one allowed.
        raise _lib.GpboError(
            "libgpbo.so was used in this process before PyTorch was imported (PointSelectorHost / host_binding first). "
            "A process that needs both must `import torch` before its first use of bayesian_optimisation_amd.")
    import torch

    if not torch.cuda.is_available():
        raise _lib.GpboError("no GPU visible: the acquisition path runs only on the HIP kernels (no CPU fallback)")
    return torch


@dataclass
class ScoreResult:
    best_val: float
    best_idx: int
    nan_count: int
    mu: Optional[object] = None      # torch fp64 device tensors [M] when dense=True
    sigma: Optional[object] = None
    acq: Optional[object] = None


class DeviceGP:
    """One BO step's surrogate on one GPU: factorise once, then score any number of candidates."""

    def __init__(self, device=None, chunk: int = DEFAULT_CHUNK):
        torch = _torch()
        self.lib = _lib.load()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if chunk % _lib.CHUNK_GRANULE:
            raise ValueError(f"chunk must be a multiple of {_lib.CHUNK_GRANULE}")
        self.chunk = int(chunk)
        self.N = self.Np = self.d = 0
        self._work_post = None
        self._work_fact = None
        self.K = self.U = self.alpha = None
        self.perm = None             # factorise(order="fps"): device int64 [N], row of the factorisation -> the caller's row
        # the 32-byte result record and the factorisation's info word share one small buffer, so that a step can read
        # both back with a single device-to-host copy (each copy is a host synchronisation)
        self._status = torch.zeros(5, dtype=torch.int64, device=self.device)
        self._result = self._status[:4]
        self.info = self._status[4:5].view(torch.int32)[:1]
        self._profile = C.c_void_p(0)
        self.screen_cap = None       # fp32 screen: most survivors re-scored in fp64 before the plain fp64 pass takes over
        self.profile_active = True   # False: the next scoring calls record no events (bench.py samples every k-th step)

    # -- per-launch timing of the dominant kernel (bench.py) -----------------------------------------
    def enable_profile(self, capacity: int = 4096):
        p = C.c_void_p(0)
        _lib.check(self.lib.gpbo_profile_create(int(capacity), C.byref(p)), "gpbo_profile_create")
        self._profile = p

    def reset_profile(self):
        self.lib.gpbo_profile_reset(self._profile)

    def read_profile(self):
        """(total ms, launches, candidates) summed over the recorded sigma/acquisition launches."""
        ms, n, c = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.gpbo_profile_read(self._profile, C.byref(ms), C.byref(n), C.byref(c)), "gpbo_profile_read")
        return ms.value, n.value, c.value

    def read_profile_kstar(self):
        """(total ms, launches, candidates) of the K(X*,X) launches recorded beside the variance launches."""
        ms, n, c = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.gpbo_profile_read_kstar(self._profile, C.byref(ms), C.byref(n), C.byref(c)),
                   "gpbo_profile_read_kstar")
        return ms.value, n.value, c.value

    def read_profile_qei(self):
        """(total ms, launches, candidates) of the qEI launches recorded by score_qei()."""
        ms, n, c = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.gpbo_profile_read_qei(self._profile, C.byref(ms), C.byref(n), C.byref(c)), "gpbo_profile_read_qei")
        return ms.value, n.value, c.value

    # -- helpers -------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, arr):
        torch = self.torch
        if isinstance(arr, torch.Tensor):
            t = arr.to(device=self.device, dtype=torch.float64)
            return t if t.is_contiguous() else t.contiguous()
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    # -- factorisation (once per BO step) ---------------------------------------------------------
    def factorise(self, X, y, ls, jitter1: float = JITTER_KERNEL, jitter2: float = JITTER_ASSEMBLY,
                  check: bool = True, order: str = "arrival"):
        """K = k(X,X) + jitter; K = L L^T; U = L^-T; alpha = K^-1 y   (point_selector.py:79, 89-90).
        order="fps": the observations are factorised in their farthest-point order (gpbo_fps_order_f64: `bound_prefix()`
        members spread over the region the observations occupy, then the others in arrival order) instead of the order in
        which they arrived.  A GP's posterior does not depend on the order of its observations - every scoring call returns
        the same numbers within rounding - but score_bound() prunes by the FIRST observations of the factorisation, and a
        history sorted along an axis or begun inside one cluster is a poor prefix.  `perm` (device int64 [N]) maps a
        row of the factorisation to the caller's row; K / U / alpha / X / y of this object are in factorisation order
        (cov_meas_host() and observations_host() give the caller's order back)."""
        torch = self.torch
        if order not in ("arrival", "fps"):
            raise ValueError("order must be 'arrival' or 'fps'")
        Xd = self._dev(X)
        if Xd.dim() != 2:
            raise ValueError("X must be (N, d)")
        N, d = int(Xd.shape[0]), int(Xd.shape[1])
        if d > _lib.MAX_D_ANY:
            raise ValueError(f"d = {d} > {_lib.MAX_D_ANY} is not supported")
        yd = self._dev(y).reshape(-1)
        if yd.numel() != N:
            raise ValueError("y must have one value per row of X")
        ls_h = np.ascontiguousarray(np.asarray(ls, dtype=np.float64).reshape(-1))
        if ls_h.size != d:
            raise ValueError(f"length scales: expected {d} values, got {ls_h.size}")
        Np = int(self.lib.gpbo_padded_n(N))
        with torch.cuda.device(self.device):
            self.perm = None
            J = self.bound_prefix(Np)
            if order == "fps" and d <= _lib.MAX_D and J < N:
                # (fewer observations than one prefix, or the slow any-d kernels: the bound route is not taken anyway)
                wob = int(self.lib.gpbo_fps_order_workspace_bytes(N))
                if getattr(self, "_work_order", None) is None or self._work_order.numel() * 8 < wob:
                    self._work_order = torch.empty((wob + 7) // 8, dtype=torch.float64, device=self.device)
                perm = torch.empty(N, dtype=torch.int64, device=self.device)
                Xp, yp = torch.empty_like(Xd), torch.empty_like(yd)
                st = self.lib.gpbo_fps_order_f64(self._ptr(Xd), self._ptr(yd), N, d, ls_h.ctypes.data_as(C.c_void_p), J,
                                                 self._ptr(perm), self._ptr(Xp), self._ptr(yp), self._ptr(self._work_order),
                                                 wob, self._stream())
                _lib.check(st, "gpbo_fps_order_f64")
                # did the co-operative selection give up (a workgroup never scheduled) and install the arrival order?  The
                # flag stays on the device until somebody asks (order_fell_back(), last_screen): no synchronisation here
                if getattr(self, "_order_flag", None) is None:
                    self._order_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.gpbo_fps_order_status(self._ptr(self._work_order), N, self._ptr(self._order_flag),
                                                          self._stream()), "gpbo_fps_order_status")
                Xd, yd, self.perm = Xp, yp, perm
            self.X, self.y, self.ls_h = Xd, yd, ls_h
            self.N, self.Np, self.d = N, Np, d
            self.jitter1, self.jitter2 = float(jitter1), float(jitter2)
            self._owns_xy = False
            self.n_appended = 0  # columns of U built by append() since the last full factorisation
            if getattr(self, "K", None) is None or self.K.shape[0] != Np or self.U.shape[0] != Np:
                # the factor buffers (and the workspace) are kept from step to step: a BO loop refactorises
                # at the same padded size many times, and fresh 100-MB allocations cost more than the kernels
                self.K = torch.empty((Np, Np), dtype=torch.float64, device=self.device)
                self.U = torch.empty((Np, Np), dtype=torch.float64, device=self.device)
                self.alpha = torch.empty(Np, dtype=torch.float64, device=self.device)
                self._work_fact = None
            self._u32_valid = False
            self._u8_valid = False
            wbytes = int(self.lib.gpbo_factorise_workspace_bytes(Np))
            if self._work_fact is None or self._work_fact.numel() * 8 < wbytes:
                self._work_fact = torch.empty(wbytes // 8, dtype=torch.float64, device=self.device)
            work = self._work_fact
            st = self.lib.gpbo_factorise_f64(self._ptr(Xd), self._ptr(yd), N, d, ls_h.ctypes.data_as(C.c_void_p),
                                             jitter1, jitter2, Np, self._ptr(self.K), self._ptr(self.U),
                                             self._ptr(self.alpha), self._ptr(self.info), self._ptr(work), wbytes,
                                             self._stream())
            _lib.check(st, "gpbo_factorise_f64")
            if check:
                info = int(self.info.item())  # synchronises
                if info != 0:
                    # the failing pivot as a row of the CALLER's arrays (gpbo_select_next_host_f64 reports the same row)
                    row = info if self.perm is None or info > N else int(self.perm[info - 1].item()) + 1
                    raise np.linalg.LinAlgError(
                        f"covariance matrix is not positive definite (pivot {row} of {N}); "
                        "the reference's np.linalg.inv would raise or return garbage here")
        return self

    BOUND_PREFIX_FRACTION = 16  # first pass over the first Np / 16 observations' columns (1/256 of the variance product);
                                # survivors get a second bound from four times as many before the fp64 kernels see them

    BOUND_MIN_JITTER = 1e-6     # score_bound(): smallest jitter1 + jitter2 the route is taken for (see score_async_bound)

    def bound_prefix(self, Np: Optional[int] = None) -> int:
        """First-level prefix length of score_bound() at this padded size (= the farthest-point members of order="fps")."""
        Np = self.Np if Np is None else int(Np)
        return max(128, (Np // self.BOUND_PREFIX_FRACTION) // 128 * 128)

    @property
    def order(self) -> str:
        return "arrival" if self.perm is None else "fps"

    def order_fell_back(self) -> bool:
        """True when factorise(order="fps") came back with the ARRIVAL order because the co-operative selection gave up (one of
        its workgroups was never scheduled: bounded waits, csrc/subset.hip).  Exact either way - but candidate shards of one
        step must hold the same factorisation, so PointSelector votes on this flag.  Reads one device word (synchronises)."""
        if self.perm is None or getattr(self, "_order_flag", None) is None:
            return False
        return bool(int(self._order_flag.item()))

    def _perm_host(self):
        return None if self.perm is None else self.perm[: self.N].cpu().numpy()

    def observations_host(self):
        """(X [N x d], y [N]) in the CALLER's order (rows appended since the factorisation come last in both orders)."""
        Xf, yf = self.X[: self.N].cpu().numpy(), self.y[: self.N].cpu().numpy()
        p = self._perm_host()
        if p is None:
            return Xf, yf
        Xa, ya = np.empty_like(Xf), np.empty_like(yf)
        Xa[p], ya[p] = Xf, yf
        return Xa, ya

    def _need_unrolled_d(self, what: str):
        """d > 16 runs on the slow any-d kernels, which serve factorise() and score() only (point_selector.py:22: the
        reference's class takes any feature count; its call path is exactly those two)."""
        if self.d > _lib.MAX_D:
            raise ValueError(f"{what} needs d <= {_lib.MAX_D} (d = {self.d}): beyond that only factorise() and the plain fp64 "
                             "score() are available")

    # -- one more observation without refactorising ---------------------------------
    def _grow(self, Np_new: int):
        """Re-pad the factors into [Np_new x Np_new] buffers (identity on the new part of the diagonal)."""
        torch = self.torch
        Np = self.Np
        for name in ("K", "U"):
            old = getattr(self, name)
            new = torch.zeros((Np_new, Np_new), dtype=torch.float64, device=self.device)
            new[:Np, :Np] = old
            new.diagonal()[Np:] = 1.0
            setattr(self, name, new)
        alpha = torch.zeros(Np_new, dtype=torch.float64, device=self.device)
        alpha[:Np] = self.alpha
        self.alpha = alpha
        self.Np = Np_new
        self._work_post = None

    def append(self, x_new, y_new, check: bool = True):
        """Add one observation to the factorised surrogate in O(N^2): column N of U, alpha recomputed.
        The length scales and jitters stay those of the last factorise() - the caller decides when they may
        (the reference re-tunes them every iteration, point_selector.py:60-62, and then a full factorise() is due).
        An appended column goes through the explicit inverse factor (l = U^T k), so it carries cond(L) eps of
        relative error where the blocked Cholesky is backward stable: `n_appended` counts the columns built this way
        since the last full factorisation, for callers that want to refresh after a while (PointSelector does)."""
        self._need_unrolled_d("append()")
        torch = self.torch
        if self.N < 1:
            raise _lib.GpboError("append() needs a factorised surrogate")
        xn = self._dev(x_new).reshape(-1)
        if xn.numel() != self.d:
            raise ValueError(f"x_new: expected {self.d} coordinates, got {xn.numel()}")
        yn = self._dev(np.asarray([float(y_new)])) if not isinstance(y_new, torch.Tensor) else self._dev(y_new).reshape(-1)[:1]
        with torch.cuda.device(self.device):
            N = self.N
            if N + 1 > self.Np:
                self._grow(int(self.lib.gpbo_padded_n(N + 1)))
            if not self._owns_xy or self.X.shape[0] < N + 1:
                # private, padded copies: the caller's X / y tensors are never written to
                Xb = torch.zeros((self.Np, self.d), dtype=torch.float64, device=self.device)
                yb = torch.zeros(self.Np, dtype=torch.float64, device=self.device)
                Xb[:N] = self.X[:N]
                yb[:N] = self.y[:N]
                self.X, self.y, self._owns_xy = Xb, yb, True
            wbytes = int(self.lib.gpbo_append_workspace_bytes(self.Np))
            work = torch.empty(wbytes // 8, dtype=torch.float64, device=self.device)
            st = self.lib.gpbo_append_f64(self._ptr(self.X), self._ptr(self.y), N, self.d,
                                          self.ls_h.ctypes.data_as(C.c_void_p), self.jitter1, self.jitter2, self.Np,
                                          self._ptr(xn), self._ptr(yn), self._ptr(self.K), self._ptr(self.U),
                                          self._ptr(self.alpha), self._ptr(self.info), self._ptr(work), wbytes,
                                          self._stream())
            _lib.check(st, "gpbo_append_f64")
            self._u32_valid = False
            self._u8_valid = False
            if check:
                info = int(self.info.item())  # synchronises
                if info != 0:
                    raise np.linalg.LinAlgError(
                        f"appended observation makes the covariance matrix numerically singular (pivot {info}); "
                        "call factorise() on the full data instead")
            else:
                torch.cuda.current_stream(self.device).synchronize()  # xn / yn / work must outlive the kernels
            if self.perm is not None:   # the new row is last in both orders
                self.perm = torch.cat([self.perm[:N], torch.tensor([N], dtype=torch.int64, device=self.device)])
            self.N = N + 1
            self.n_appended += 1
            del work
        return self

    # -- persistence across jobs: the DAG's select_parameters jobs are separate processes --------------------
    def state_dict(self) -> dict:
        """Host copy of everything append()/score() need (the N x N part of the factors, not the padding)."""
        N = self.N
        st = dict(version=1, N=N, d=self.d, ls=np.array(self.ls_h), jitter1=self.jitter1, jitter2=self.jitter2,
                  n_appended=self.n_appended,
                  X=self.X[:N].cpu().numpy(), y=self.y[:N].cpu().numpy(), K=self.K[:N, :N].cpu().numpy(),
                  U=self.U[:N, :N].cpu().numpy(), alpha=self.alpha[:N].cpu().numpy())
        if self.perm is not None:   # everything above is in factorisation order; perm[i] = the caller's row of row i
            st["perm"] = self._perm_host()
        return st

    def load_state_dict(self, st: dict):
        torch = self.torch
        if int(st.get("version", 0)) != 1:
            raise ValueError("unknown surrogate state version")
        N, d = int(st["N"]), int(st["d"])
        Np = int(self.lib.gpbo_padded_n(N))
        with torch.cuda.device(self.device):
            self.N, self.Np, self.d = N, Np, d
            self.ls_h = np.ascontiguousarray(np.asarray(st["ls"], dtype=np.float64).reshape(-1))
            self.jitter1, self.jitter2 = float(st["jitter1"]), float(st["jitter2"])
            self.n_appended = int(st["n_appended"]) if "n_appended" in st else 0
            self.X = torch.zeros((Np, d), dtype=torch.float64, device=self.device)
            self.y = torch.zeros(Np, dtype=torch.float64, device=self.device)
            self.X[:N] = self._dev(st["X"])
            self.y[:N] = self._dev(st["y"]).reshape(-1)
            self._owns_xy = True
            for name in ("K", "U"):
                m = torch.zeros((Np, Np), dtype=torch.float64, device=self.device)
                m[:N, :N] = self._dev(st[name])
                m.diagonal()[N:] = 1.0
                setattr(self, name, m)
            self.alpha = torch.zeros(Np, dtype=torch.float64, device=self.device)
            self.alpha[:N] = self._dev(st["alpha"]).reshape(-1)
            self.info.zero_()
            self.perm = None
            if "perm" in st and st["perm"] is not None:
                p = np.ascontiguousarray(np.asarray(st["perm"], dtype=np.int64).reshape(-1))
                if p.size != N or not np.array_equal(np.sort(p), np.arange(N)):
                    raise ValueError("surrogate state: perm is not a permutation of the observations")
                self.perm = torch.from_numpy(p).to(self.device)
            self._u32_valid = False
            self._u8_valid = False
            self._work_post = None
        return self

    def save_state(self, path: str):
        """Atomic: written to a temporary file in the same directory and renamed over `path`, so a job killed
        mid-write (or a concurrent reader) never sees a torn file."""
        import os
        import tempfile

        path = str(path)
        if not path.endswith(".npz"):
            path += ".npz"  # what np.savez would have appended
        fd, tmp = tempfile.mkstemp(prefix=".gpbo_state_", suffix=".tmp", dir=os.path.dirname(os.path.abspath(path)))
        try:
            with os.fdopen(fd, "wb") as f:
                np.savez(f, **self.state_dict())
            # mkstemp creates 0600; np.savez(path) honoured the umask - a follow-up job under another account of the
            # group must still be able to read the state, as it could before the atomic rename was introduced
            um = os.umask(0)
            os.umask(um)
            os.chmod(tmp, 0o666 & ~um)
            os.replace(tmp, path)
        except BaseException:
            if os.path.exists(tmp):
                os.unlink(tmp)
            raise

    def load_state(self, path: str):
        with np.load(path) as z:
            return self.load_state_dict({k: z[k] for k in z.files})

    # -- scoring ------------------------------------------------------------------------------------
    def _ensure_post_workspace(self, M):
        torch = self.torch
        chunk = min(self.chunk, (M + _lib.CHUNK_GRANULE - 1) // _lib.CHUNK_GRANULE * _lib.CHUNK_GRANULE)
        need = int(self.lib.gpbo_posterior_workspace_bytes(self.Np, chunk, M))
        if need < 0:
            raise _lib.GpboError("gpbo_posterior_workspace_bytes: invalid sizes")
        if self._work_post is None or self._work_post.numel() * 8 < need:
            self._work_post = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
        return chunk, need

    def score_async(self, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best: Optional[float] = None,
                    xi: float = 0.0, dense: bool = False, idx_offset: int = 0, diag_add: float = 0.0,
                    prior_var: float = PRIOR_VAR):
        """Enqueue K(X*,X) + mu + sigma + acquisition + arg-max for all rows of Xs; no host sync.
        Returns (result_tensor[int64 x4 on device], mu, sigma, acq)."""
        torch = self.torch
        Xsd = self._dev(Xs)
        if Xsd.dim() != 2 or int(Xsd.shape[1]) != self.d:
            raise ValueError("Xs must be (M, d) with the same d as X")
        M = int(Xsd.shape[0])
        if acquisition == "lcb":
            kind, p0, p1 = _lib.ACQ_LCB, float(explore), 0.0
        elif acquisition == "ei":
            if f_best is None:
                raise ValueError("EI needs f_best (the incumbent minimum)")
            kind, p0, p1 = _lib.ACQ_EI, float(f_best), float(xi)
        else:
            raise ValueError(f"unknown acquisition {acquisition!r}")
        if diag_add != 0.0 and self.perm is not None:
            # the N == M quirk (point_selector.py:173) adds to entry (i, i) of k(X, X*): candidate i against the caller's
            # observation i, which is not row i of a permuted factorisation
            raise ValueError("diag_add (the N == M shape quirk) needs factorise(order='arrival')")
        with torch.cuda.device(self.device):
            chunk, wbytes = self._ensure_post_workspace(M)
            mu = sigma = acq = None
            if dense:
                mu = torch.empty(M, dtype=torch.float64, device=self.device)
                sigma = torch.empty(M, dtype=torch.float64, device=self.device)
                acq = torch.empty(M, dtype=torch.float64, device=self.device)
            st = self.lib.gpbo_posterior_acq_f64(
                self._ptr(Xsd), M, self._ptr(self.X), self.N, self.Np, self.d,
                self.ls_h.ctypes.data_as(C.c_void_p), self._ptr(self.U), self._ptr(self.alpha), prior_var,
                kind, p0, p1, float(diag_add), int(idx_offset), chunk, self._ptr(mu), self._ptr(sigma),
                self._ptr(acq), self._ptr(self._result), self._ptr(self._work_post), wbytes,
                self._profile if self.profile_active else None,
                self._stream())
            _lib.check(st, "gpbo_posterior_acq_f64")
        self._keep = Xsd  # keep the candidate tensor alive until the stream has consumed it
        return self._result, mu, sigma, acq

    # -- fp32-screened scoring (BASELINE config 4): fp32 variance product for all, fp64 re-score of the survivors ----
    SCREEN_TAU0 = 1e-4          # first guess of |var64 - var32|; checked and raised per call (gpbo_rescore_f64)
    SCREEN_SAMPLE = 1024        # about this many evenly strided non-survivors are re-scored as well, to check tau
    SCREEN_CHUNK64 = 1 << 14    # candidates per fp64 launch of the re-scoring

    def prepare_f32(self):
        """Round U to fp32 (re-padded to a multiple of 256) for the fp32 variance screen."""
        torch = self.torch
        Np32 = int(self.lib.gpbo_padded_n_f32(self.N))
        with torch.cuda.device(self.device):
            if getattr(self, "U32", None) is None or self.U32.shape[0] != Np32:
                self.U32 = torch.empty((Np32, Np32), dtype=torch.float32, device=self.device)
            st = self.lib.gpbo_prepare_f32(self._ptr(self.U), None, self.Np, self._ptr(self.U32), None, Np32,
                                           self._stream())
            _lib.check(st, "gpbo_prepare_f32")
        if getattr(self, "Np32", 0) != Np32:
            self._work_post32 = None
        self.Np32 = Np32
        self._u32_valid = True
        return self

    def prepare_i8(self):
        """Column scales and int8 MFMA fragments of U for the int8-sliced variance screen (once per factorisation)."""
        torch = self.torch
        if self.Np > _lib.I8_MAX_N:
            raise _lib.GpboError(f"the int8-sliced screen needs N <= {_lib.I8_MAX_N} (int32 accumulators)")
        with torch.cuda.device(self.device):
            need = int(self.lib.gpbo_prepare_i8_bytes(self.Np))
            if need < 0:
                raise _lib.GpboError("gpbo_prepare_i8_bytes: invalid size")
            if getattr(self, "U8", None) is None or self.U8.numel() < need:
                self.U8 = torch.empty(need, dtype=torch.uint8, device=self.device)
            st = self.lib.gpbo_prepare_i8(self._ptr(self.U), self.Np, self._ptr(self.U8), need, self._stream())
            _lib.check(st, "gpbo_prepare_i8")
        self._u8_valid = True
        return self

    SCREEN_TAU0_I8 = 1e-9       # int8-sliced screen: |var64 - var_i8| is ~1e-11; checked per call like the fp32 one
    SCREEN_TAU0_I8C = 1e-3      # coarse int8 screen (three digits per operand): |var64 - var| ~ 2e-4 at N = 4096

    def _score_screened(self, mode, Xs, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var):
        self._need_unrolled_d(f"the {mode} screen")
        """A reduced-cost pass over all rows of Xs (mode "f32": fp32 matrix cores; "i8": int8 slices on the integer
        matrix cores), then the fp64 decision (gpbo_rescore_f64): the result record holds the fp64 kernels' maximum and
        its lowest index.  Dense outputs (mu exactly the fp64 path's; sigma / acq with the screen's variance) are
        float64 tensors.  Unlike score_async this call synchronises (the survivor count is read back).
        `last_screen` keeps the statistics of the call."""
        torch = self.torch
        Xsd = self._dev(Xs)
        if Xsd.dim() != 2 or int(Xsd.shape[1]) != self.d:
            raise ValueError("Xs must be (M, d) with the same d as X")
        M = int(Xsd.shape[0])
        if acquisition == "lcb":
            kind, p0, p1 = _lib.ACQ_LCB, float(explore), 0.0
        elif acquisition == "ei":
            if f_best is None:
                raise ValueError("EI needs f_best (the incumbent minimum)")
            kind, p0, p1 = _lib.ACQ_EI, float(f_best), float(xi)
        else:
            raise ValueError(f"unknown acquisition {acquisition!r}")
        if diag_add != 0.0:
            # N == M shape quirk (point_selector.py:173): gathered rows lose the index the quirk is keyed on
            self.last_screen = dict(fallback=True, reason="diag_add")
            return self.score_async(Xsd, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)
        if mode == "f32" and (not getattr(self, "_u32_valid", False) or getattr(self, "U32", None) is None):
            self.prepare_f32()
        if mode in ("i8", "i8c") and (not getattr(self, "_u8_valid", False) or getattr(self, "U8", None) is None):
            self.prepare_i8()
        with torch.cuda.device(self.device):
            if getattr(self, "_screen_mu", None) is None or self._screen_mu.numel() < M:
                self._screen_mu = torch.empty(M, dtype=torch.float64, device=self.device)
                self._screen_var = torch.empty(M, dtype=torch.float64, device=self.device)
            mu = sigma = acq = None
            if dense:
                mu, sigma, acq = (torch.empty(M, dtype=torch.float64, device=self.device) for _ in range(3))
            mu_w = mu if dense else self._screen_mu
            prof = self._profile if self.profile_active else None
            lsp = self.ls_h.ctypes.data_as(C.c_void_p)
            if mode == "f32":
                chunk = min(self.chunk, (M + 1023) // 1024 * 1024)
                chunk = (chunk + 1023) // 1024 * 1024
                need = int(self.lib.gpbo_posterior_workspace_bytes_f32(self.Np32, chunk, M))
            else:
                chunk = min(self.chunk, (M + _lib.CHUNK_GRANULE - 1) // _lib.CHUNK_GRANULE * _lib.CHUNK_GRANULE)
                need = int(self.lib.gpbo_posterior_workspace_bytes_i8(self.Np, chunk, M))
            if need < 0:
                raise _lib.GpboError("screen workspace: invalid sizes")
            if getattr(self, "_work_screen", None) is None or self._work_screen.numel() * 8 < need:
                self._work_screen = None
                self._work_screen = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
            if mode == "f32":
                st = self.lib.gpbo_posterior_acq_f32(
                    self._ptr(Xsd), M, self._ptr(self.X), self.N, self.Np32, self.d, lsp, self._ptr(self.U32),
                    self._ptr(self.alpha), prior_var, kind, p0, p1, 0.0, int(idx_offset), chunk, self._ptr(mu_w),
                    self._ptr(sigma), self._ptr(acq), self._ptr(self._screen_var), self._ptr(self._result),
                    self._ptr(self._work_screen), need, prof, self._stream())
                _lib.check(st, "gpbo_posterior_acq_f32")
                tau0 = float(self.SCREEN_TAU0)
            else:
                fn = self.lib.gpbo_posterior_acq_i8c if mode == "i8c" else self.lib.gpbo_posterior_acq_i8
                st = fn(
                    self._ptr(Xsd), M, self._ptr(self.X), self.N, self.Np, self.d, lsp, self._ptr(self.U8),
                    self._ptr(self.alpha), prior_var, kind, p0, p1, int(idx_offset), chunk, self._ptr(mu_w),
                    self._ptr(sigma), self._ptr(acq), self._ptr(self._screen_var), self._ptr(self._result),
                    self._ptr(self._work_screen), need, prof, self._stream())
                _lib.check(st, "gpbo_posterior_acq_" + mode)
                tau0 = float(self.SCREEN_TAU0_I8C if mode == "i8c" else self.SCREEN_TAU0_I8)
            cap = self.screen_cap if self.screen_cap else max(4096, min(M, max(M // 16, 1 << 16)))
            chunk64 = self.SCREEN_CHUNK64
            rbytes = int(self.lib.gpbo_rescore_workspace_bytes(self.Np, cap, chunk64))
            if rbytes < 0:
                raise _lib.GpboError("gpbo_rescore_workspace_bytes: invalid sizes")
            if getattr(self, "_work_rescore", None) is None or self._work_rescore.numel() * 8 < rbytes:
                self._work_rescore = torch.empty((rbytes + 7) // 8, dtype=torch.float64, device=self.device)
            stats = _lib.ScreenStats()
            stride = max(1, M // self.SCREEN_SAMPLE)
            st = self.lib.gpbo_rescore_f64(
                self._ptr(Xsd), M, self._ptr(mu_w), self._ptr(self._screen_var), self._ptr(self.X), self.N, self.Np,
                self.d, lsp, self._ptr(self.U), self._ptr(self.alpha), prior_var, kind,
                p0, p1, int(idx_offset), tau0, stride, cap, chunk64, self._ptr(self._result),
                C.byref(stats), self._ptr(self._work_rescore), rbytes, self._stream())
            _lib.check(st, "gpbo_rescore_f64")
            self.last_screen = dict(mode=mode, survivors=int(stats.survivors), rescored=int(stats.rescored),
                                    rounds=int(stats.rounds), fallback=bool(stats.fallback), tau=float(stats.tau),
                                    err_max=float(stats.err_max), candidates=M)
        self._keep = Xsd
        if stats.fallback:
            # too many candidates could still be the maximum (or tau did not settle): the plain fp64 pass decides
            return self.score_async(Xsd, acquisition, explore, f_best, xi, dense, idx_offset, 0.0, prior_var)
        return self._result, mu, sigma, acq

    def score_async_f32(self, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best: Optional[float] = None,
                        xi: float = 0.0, dense: bool = False, idx_offset: int = 0, diag_add: float = 0.0,
                        prior_var: float = PRIOR_VAR):
        """fp32 variance screen + fp64 decision (BASELINE config 4); see _score_screened."""
        return self._score_screened("f32", Xs, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)

    def score_async_i8(self, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best: Optional[float] = None,
                       xi: float = 0.0, dense: bool = False, idx_offset: int = 0, diag_add: float = 0.0,
                       prior_var: float = PRIOR_VAR):
        """int8-sliced variance screen (|dsigma| ~ 1e-10) + fp64 decision; see _score_screened."""
        return self._score_screened("i8", Xs, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)

    def score_i8(self, Xs, **kw) -> ScoreResult:
        res, mu, sigma, acq = self.score_async_i8(Xs, **kw)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, mu, sigma, acq)

    def score_async_i8c(self, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best: Optional[float] = None,
                        xi: float = 0.0, dense: bool = False, idx_offset: int = 0, diag_add: float = 0.0,
                        prior_var: float = PRIOR_VAR):
        """Coarse int8 screen (three digits per operand, six slice products, |dsigma^2| ~ 2e-4) + fp64 decision: the
        cheapest pass that still leaves only a handful of candidates for the fp64 kernels; see _score_screened."""
        return self._score_screened("i8c", Xs, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)

    def score_i8c(self, Xs, **kw) -> ScoreResult:
        res, mu, sigma, acq = self.score_async_i8c(Xs, **kw)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, mu, sigma, acq)

    def score_f32(self, Xs, **kw) -> ScoreResult:
        res, mu, sigma, acq = self.score_async_f32(Xs, **kw)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, mu, sigma, acq)

    # -- prefix-bound screen: exact branch and bound, all fp64 ---------------------------------------------------
    ARD_KEEP_WORKSPACE_BYTES = 1 << 28   # the batched ARD workspace is dropped after a call when larger than this
    def score_async_bound(self, Xs, acquisition: str = "lcb", explore: float = 4.0, f_best: Optional[float] = None,
                          xi: float = 0.0, dense: bool = False, idx_offset: int = 0, diag_add: float = 0.0,
                          prior_var: float = PRIOR_VAR, prefix: Optional[int] = None, prefix2: Optional[int] = None):
        """The selected point WITHOUT the variance of every candidate - exact, all fp64.  The squared norm of the first J
        components of v_c = U^T k_c is the variance reduction from the first J observations alone, so
        sqrt(prior_var - |v_c[:J]|^2) >= cov_func_c and (both acquisitions increase with sigma) an UPPER bound of the
        acquisition follows at (J / N)^2 of the cost.  Candidates whose bound is below the best exact value of a sample
        cannot be the maximum nor tie with it; the others go through the fp64 kernels, which decide
        (gpbo_posterior_prefix_f64 + gpbo_bound_select_f64).  The mean is still computed for every candidate.
        Falls back to score_async when the bound cannot be used (dense outputs wanted, LCB with a negative weight, the
        N == M diagonal quirk, fewer than 3 column blocks) or does not separate the candidates (flat mean, ties).
        "The first J observations" are those of THIS object's factorisation: factorise(order="fps") puts J well-spread
        members first, so that the pruning does not depend on the order in which the observations arrived; with the
        default order the literal arrival prefix is used (exact as well; prunes little when the history is sorted or
        clustered).  Bounds and exact values come from the same U, the same K(X*,X) kernel and the same variance
        kernel: |v[:J]|^2 is a partial sum of the squares score() adds up.
        Synchronises; `last_screen` keeps the statistics."""
        torch = self.torch
        if self.d > _lib.MAX_D:   # the slow any-d kernels serve the plain pass only
            self.last_screen = dict(mode="bound", fallback=True, reason="d > 16")
            return self.score_async(Xs, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)
        Xsd = self._dev(Xs)
        if Xsd.dim() != 2 or int(Xsd.shape[1]) != self.d:
            raise ValueError("Xs must be (M, d) with the same d as X")
        M = int(Xsd.shape[0])
        if acquisition == "lcb":
            kind, p0, p1 = _lib.ACQ_LCB, float(explore), 0.0
        elif acquisition == "ei":
            if f_best is None:
                raise ValueError("EI needs f_best (the incumbent minimum)")
            kind, p0, p1 = _lib.ACQ_EI, float(f_best), float(xi)
        else:
            raise ValueError(f"unknown acquisition {acquisition!r}")
        J = int(prefix) if prefix else self.bound_prefix()
        J2 = int(prefix2) if prefix2 is not None else (4 * J if 8 * J <= self.Np else 0)   # second level: survivors only
        reason = None
        if dense:
            reason = "dense outputs"
        elif diag_add != 0.0:
            reason = "diag_add"
        elif kind == _lib.ACQ_LCB and not p0 >= 0.0:
            reason = "negative explore weight"
        elif J % 128 or J < 128 or 2 * J > self.Np:
            reason = "too few column blocks"
        elif not (self.jitter1 + self.jitter2 >= self.BOUND_MIN_JITTER
                  and prior_var >= (1.0 + self.jitter1) + self.jitter2 - 1e-15):
            # The plain pass takes sqrt(|var|) (point_selector.py:98): a NEGATIVE computed variance counts by its magnitude,
            # which no prefix can bound.  With K = k(X,X) + tau I and prior_var >= 1 + tau the true variance is >= tau
            # (a candidate on top of an observation: 2 tau - tau^2 (K^-1)_ii), so the computed one is negative only if its
            # rounding error exceeds tau; the route is taken for tau >= 1e-6 (the reference: 1.01e-4, errors seen: 1e-9).
            reason = "jitter too small for the |var| rule"
        if reason:
            self.last_screen = dict(mode="bound", fallback=True, reason=reason)
            return self.score_async(Xsd, acquisition, explore, f_best, xi, dense, idx_offset, diag_add, prior_var)
        with torch.cuda.device(self.device):
            chunk, wbytes = self._ensure_post_workspace(M)
            if getattr(self, "_bound_ub", None) is None or self._bound_ub.numel() < M:
                self._bound_ub = torch.empty(M, dtype=torch.float64, device=self.device)
            lsp = self.ls_h.ctypes.data_as(C.c_void_p)
            st = self.lib.gpbo_posterior_prefix_f64(
                self._ptr(Xsd), M, self._ptr(self.X), self.N, self.Np, self.d, lsp, self._ptr(self.U),
                self._ptr(self.alpha), prior_var, kind, p0, p1, int(idx_offset), chunk, J, None, None,
                self._ptr(self._bound_ub), self._ptr(self._result), self._ptr(self._work_post), wbytes,
                self._profile if self.profile_active else None, self._stream())
            _lib.check(st, "gpbo_posterior_prefix_f64")
            # every first-level survivor may go on to the second-level bound (N/4 rows: 1/16 of a plain pass per candidate -
            # cheaper than falling back even if ALL of them survive); LCB(explore=10) on the headline workload leaves 256k
            # of 2^21: 24 ms through the second level against 518 ms through the plain pass (tools/bound_cap_probe.py)
            cap = self.screen_cap if self.screen_cap else max(4096, M)
            chunk64 = self.SCREEN_CHUNK64
            rbytes = int(self.lib.gpbo_rescore_workspace_bytes(self.Np, cap, chunk64))
            if rbytes < 0:
                raise _lib.GpboError("gpbo_rescore_workspace_bytes: invalid sizes")
            if getattr(self, "_work_rescore", None) is None or self._work_rescore.numel() * 8 < rbytes:
                self._work_rescore = torch.empty((rbytes + 7) // 8, dtype=torch.float64, device=self.device)
            stats = _lib.ScreenStats()
            stride = max(1, M // self.SCREEN_SAMPLE)
            st = self.lib.gpbo_bound_select_f64(
                self._ptr(Xsd), M, self._ptr(self._bound_ub), self._ptr(self.X), self.N, self.Np, self.d, lsp,
                self._ptr(self.U), self._ptr(self.alpha), prior_var, kind, p0, p1, int(idx_offset), stride, cap,
                chunk64, J2, self._ptr(self._result), C.byref(stats), self._ptr(self._work_rescore), rbytes,
                self._stream())
            _lib.check(st, "gpbo_bound_select_f64")
            self.last_screen = dict(mode="bound", order="arrival (fps fell back)" if self.order_fell_back() else self.order, prefix=J, prefix2=J2, survivors=int(stats.survivors),
                                    rescored=int(stats.rescored), rounds=int(stats.rounds), fallback=bool(stats.fallback),
                                    threshold=float(stats.tau), candidates=M)
        self._keep = Xsd
        if stats.fallback:
            return self.score_async(Xsd, acquisition, explore, f_best, xi, False, idx_offset, 0.0, prior_var)
        return self._result, None, None, None

    def score_bound(self, Xs, **kw) -> ScoreResult:
        res, mu, sigma, acq = self.score_async_bound(Xs, **kw)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, mu, sigma, acq)

    # -- q = 8 Monte-Carlo Expected Improvement (BASELINE config 5) ---------------------------------------
    def score_qei_async(self, Xs, Z, f_best: float, xi: float = 0.0, dense: bool = False, batch_offset: int = 0,
                        prior_var: float = PRIOR_VAR):
        """Enqueue qEI over consecutive batches of 8 rows of Xs; Z = [S x 8] base samples; no host sync.
        Returns (result_tensor, qei or None); the result's best_idx is a BATCH index."""
        self._need_unrolled_d("qEI")
        torch = self.torch
        Xsd, Zd = self._dev(Xs), self._dev(Z)
        M, S = int(Xsd.shape[0]), int(Zd.shape[0])
        if M % 8 or int(Zd.shape[1]) != 8:
            raise ValueError("qEI needs M % 8 == 0 and base samples of shape (S, 8)")
        with torch.cuda.device(self.device):
            chunk = min(self.chunk, (M + _lib.CHUNK_GRANULE - 1) // _lib.CHUNK_GRANULE * _lib.CHUNK_GRANULE)
            need = int(self.lib.gpbo_qei_workspace_bytes(self.Np, chunk, M))
            if need < 0:
                raise _lib.GpboError("gpbo_qei_workspace_bytes: invalid sizes")
            if getattr(self, "_work_qei", None) is None or self._work_qei.numel() * 8 < need:
                self._work_qei = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
            qei = torch.empty(M // 8, dtype=torch.float64, device=self.device) if dense else None
            st = self.lib.gpbo_posterior_qei_f64(
                self._ptr(Xsd), M, self._ptr(self.X), self.N, self.Np, self.d, self.ls_h.ctypes.data_as(C.c_void_p),
                self._ptr(self.U), self._ptr(self.alpha), prior_var, float(f_best), float(xi), self._ptr(Zd), S,
                int(batch_offset), chunk, self._ptr(qei), self._ptr(self._result), self._ptr(self._work_qei), need,
                self._profile if self.profile_active else None, self._stream())
            _lib.check(st, "gpbo_posterior_qei_f64")
        self._keep = (Xsd, Zd)
        return self._result, qei

    def score_qei(self, Xs, Z, f_best: float, xi: float = 0.0, dense: bool = False, batch_offset: int = 0,
                  prior_var: float = PRIOR_VAR) -> ScoreResult:
        """qEI over consecutive batches of 8 rows of Xs; Z = [S x 8] base samples.  best_idx is a BATCH index."""
        res, qei = self.score_qei_async(Xs, Z, f_best, xi, dense, batch_offset, prior_var)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, None, None, qei)

    @property
    def status(self):
        """Device int64[5]: the gpbo_result record of the last scoring call + the info word of the last factorisation
        (what distributed.allreduce_status gathers over the ranks)."""
        return self._status

    def read_result_and_info(self) -> tuple:
        """(best value, best index, NaN count, info of the last factorise/append) with one device-to-host copy."""
        r = self._status.cpu()  # synchronises
        best_val = float(r[:1].view(self.torch.float64)[0])
        return best_val, int(r[1]), int(r[2]), int(r[4:5].view(self.torch.int32)[0])

    def read_result(self, result_tensor) -> tuple:
        r = result_tensor.cpu()  # synchronises
        best_val = float(r[:1].view(self.torch.float64)[0])
        return best_val, int(r[1]), int(r[2])

    def score(self, Xs, **kw) -> ScoreResult:
        res, mu, sigma, acq = self.score_async(Xs, **kw)
        v, i, n = self.read_result(res)
        return ScoreResult(v, i, n, mu, sigma, acq)

    def acquisition_on_posterior(self, mu, sigma, acquisition: str = "lcb", explore: float = 4.0,
                                 f_best: Optional[float] = None, xi: float = 0.0, idx_offset: int = 0) -> ScoreResult:
        """Second acquisition on dense device mu/sigma (lower_confidence_bound(explore) after the fact)."""
        torch = self.torch
        M = int(mu.numel())
        kind, p0, p1 = (_lib.ACQ_LCB, float(explore), 0.0) if acquisition == "lcb" else (_lib.ACQ_EI, float(f_best), float(xi))
        with torch.cuda.device(self.device):
            acq = torch.empty(M, dtype=torch.float64, device=self.device)
            wbytes = int(self.lib.gpbo_acq_workspace_bytes())
            work = torch.empty(wbytes // 8 + 32, dtype=torch.float64, device=self.device)
            st = self.lib.gpbo_acq_argmax_f64(self._ptr(mu), self._ptr(sigma), M, kind, p0, p1, int(idx_offset),
                                              self._ptr(acq), self._ptr(self._result), self._ptr(work), wbytes,
                                              self._stream())
            _lib.check(st, "gpbo_acq_argmax_f64")
            v, i, n = self.read_result(self._result)
        return ScoreResult(v, i, n, mu, sigma, acq)

    # -- ARD grid ------------------------------------------------------------------------------------
    ARD_KERNEL = "wave"  # N <= gpbo_nlml_grid_wave_max_n() (64): "wave" = a wave per cell, the matrix in registers (csrc/ard_wave.hip,
                         # round 5: 2,500 cells at N = 16 / 32 / 48 / 64, d = 2: 0.014 / 0.026 / 0.049 / 0.068 ms);
                         # "lds" = the workgroup-per-cell kernel of round 2 up to ARD_LDS_MAX_N, the fused kernel beyond (the
                         # routing until the end of round 5: 0.044 / 0.116 / 0.125 / 0.122 ms; tools/ard_lds_vs_fused.py).
                         # The float32 cells of the three kernels are equal.
    ARD_LDS_MAX_N = 32

    def nlml_grid(self, X, y, ls_cells, jitter: float = JITTER_KERNEL, likelihood: str = "reference") -> np.ndarray:
        """-log marginal likelihood of every row of ls_cells [G x d]  (point_selector.py:111-156), as a host array.

        likelihood="reference": the reference's value - float32, log det K taken as np.log(np.linalg.det(K)), which
        underflows to -inf beyond N ~ 100 (reproduced for parity).  likelihood="logdet": fp64, log det K = 2 sum log L_ii
        from the factor - finite at any N, NaN where a pivot is not positive (a documented departure)."""
        res = self.nlml_grid_device(X, y, ls_cells, jitter, likelihood).cpu().numpy()   # synchronises: the workspace is idle
        # the scratch slots can be GiBs (512 x (N + 16) x N x 8 bytes): kept between the calls of one search (a
        # coordinate-wise ARD search calls this once per axis and sweep) only while they are small
        if getattr(self, "_work_ard", None) is not None and self._work_ard.numel() * 8 > self.ARD_KEEP_WORKSPACE_BYTES:
            self._work_ard = None
        return res

    def nlml_grid_device(self, X, y, ls_cells, jitter: float = JITTER_KERNEL, likelihood: str = "reference"):
        """The same grid left on the device (float32 / float64 tensor [G]); enqueues on the current stream, no read-back."""
        if likelihood not in ("reference", "logdet"):
            raise ValueError(f"likelihood must be 'reference' or 'logdet', got {likelihood!r}")
        torch = self.torch
        Xd, yd = self._dev(X), self._dev(y).reshape(-1)
        N, d = int(Xd.shape[0]), int(Xd.shape[1])
        cells = ls_cells if isinstance(ls_cells, torch.Tensor) else np.asarray(ls_cells, dtype=np.float64).reshape(-1, d)
        cells = self._dev(cells).reshape(-1, d)
        G = int(cells.shape[0])
        logdet = likelihood == "logdet"
        with torch.cuda.device(self.device):
            wave = self.ARD_KERNEL == "wave" and N <= int(self.lib.gpbo_nlml_grid_wave_max_n())
            if wave:
                # the reference's own sizes: a wave per cell, the matrix in registers (csrc/ard_wave.hip), both likelihood modes;
                # pinned against the reference's float32 ties (golden g4_ard_n2) like the in-LDS kernel it replaced
                out = torch.empty(G, dtype=torch.float64 if logdet else torch.float32, device=self.device)
                fn = self.lib.gpbo_nlml_grid_wave_logdet_f64 if logdet else self.lib.gpbo_nlml_grid_wave_f64
                st = fn(self._ptr(Xd), self._ptr(yd), N, d, self._ptr(cells), G, float(jitter), self._ptr(out), self._stream())
                _lib.check(st, "gpbo_nlml_grid_wave_logdet_f64" if logdet else "gpbo_nlml_grid_wave_f64")
            elif N > self.ARD_LDS_MAX_N or logdet:
                # one persistent workgroup per cell, the whole factorisation in one launch (csrc/ard.hip, round 5)
                out = torch.empty(G, dtype=torch.float64 if logdet else torch.float32, device=self.device)
                need = int(self.lib.gpbo_nlml_grid_batched_workspace_bytes(N, G))
                if need < 0:
                    raise _lib.GpboError("gpbo_nlml_grid_batched_workspace_bytes: invalid sizes")
                if getattr(self, "_work_ard", None) is None or self._work_ard.numel() * 8 < need:
                    self._work_ard = None
                    self._work_ard = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
                fn = self.lib.gpbo_nlml_grid_batched_logdet_f64 if logdet else self.lib.gpbo_nlml_grid_batched_f64
                st = fn(self._ptr(Xd), self._ptr(yd), N, d, self._ptr(cells), G, float(jitter), self._ptr(out),
                        self._ptr(self._work_ard), need, self._stream())
                _lib.check(st, "gpbo_nlml_grid_batched_logdet_f64" if logdet else "gpbo_nlml_grid_batched_f64")
            else:
                out = torch.empty(G, dtype=torch.float32, device=self.device)
                st = self.lib.gpbo_nlml_grid_f64(self._ptr(Xd), self._ptr(yd), N, d, self._ptr(cells), G, float(jitter),
                                                 self._ptr(out), self._stream())
                _lib.check(st, "gpbo_nlml_grid_f64")
            self._keep_ard = (Xd, yd, cells)   # alive until the stream has consumed them
        return out

    # -- dense covariance blocks for inspection (small problems only) --------------------------------------
    def cov_meas_host(self) -> np.ndarray:
        """cov_meas (point_selector.py:79) in the caller's order of the observations."""
        K = self.K[: self.N, : self.N].cpu().numpy()
        p = self._perm_host()
        if p is None:
            return K
        Ka = np.empty_like(K)
        Ka[np.ix_(p, p)] = K
        return Ka

    def kxx_host(self, P, ls, jitter1, jitter2) -> np.ndarray:
        """k(P,P) with the reference's jitter, as a host array (used for `cov_pred` on small grids)."""
        torch = self.torch
        Pd = self._dev(P)
        n, d = int(Pd.shape[0]), int(Pd.shape[1])
        npad = int(self.lib.gpbo_padded_n(n))
        ls_h = np.ascontiguousarray(np.asarray(ls, dtype=np.float64).reshape(-1))
        with torch.cuda.device(self.device):
            Kp = torch.empty((npad, npad), dtype=torch.float64, device=self.device)
            st = self.lib.gpbo_kxx_f64(self._ptr(Pd), n, d, ls_h.ctypes.data_as(C.c_void_p), jitter1, jitter2,
                                       self._ptr(Kp), npad, self._stream())
            _lib.check(st, "gpbo_kxx_f64")
            return Kp[:n, :n].cpu().numpy()

    def cov_meas_pred_host(self, Xs, diag_add: float = 0.0) -> np.ndarray:
        """K(X*, X) as an (M, N) host array  (point_selector.py:81); small problems only."""
        torch = self.torch
        Xsd = self._dev(Xs)
        M = int(Xsd.shape[0])
        ldk = (M + _lib.CHUNK_GRANULE - 1) // _lib.CHUNK_GRANULE * _lib.CHUNK_GRANULE
        with torch.cuda.device(self.device):
            kst = torch.empty((self.Np, ldk), dtype=torch.float64, device=self.device)
            mup = torch.empty((self.Np // 64, ldk), dtype=torch.float64, device=self.device)
            xsc = torch.empty((self.Np, self.d), dtype=torch.float64, device=self.device)
            lsp = self.ls_h.ctypes.data_as(C.c_void_p)
            _lib.check(self.lib.gpbo_scale_points_f64(self._ptr(self.X), self.N, self.Np, self.d, lsp, self._ptr(xsc),
                                                      self._stream()), "gpbo_scale_points_f64")
            st = self.lib.gpbo_kstar_mu_f64(self._ptr(Xsd), M, self._ptr(xsc), self.N, self.Np, self.d, lsp,
                                            self._ptr(self.alpha), float(diag_add), 0, self._ptr(kst), ldk,
                                            self._ptr(mup), self._stream())
            _lib.check(st, "gpbo_kstar_mu_f64")
            Kf = kst[: self.N, :M].t().contiguous().cpu().numpy()
        p = self._perm_host()
        if p is None:
            return Kf
        Ka = np.empty_like(Kf)   # columns back in the caller's order of the observations
        Ka[:, p] = Kf
        return Ka
