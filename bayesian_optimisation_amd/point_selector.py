"""Drop-in replacement for the reference's `PointSelector` running on the MI355X kernels.

Same attribute protocol and call sequence as /root/reference/point_selector.py:13-207 as driven by
/root/reference/select_parameters.py:146-157 and :282-293:

    ps = PointSelector(); ps.name = ...; ps.iteration = ...
    ps.measured_pts, ps.measured_vals, ps.feature_domain, ps.predicted_pts, ps.length_scales = ...
    ps.update_surrogate()
    idx = ps.lower_confidence_bound()          # np.ndarray[int64], one entry per feature axis
    ps.mean_func, ps.cov_func, ps.acq_func_eval, ps.kernel_params

Differences from the reference, all deliberate:
  * `cov_pred` (the M x M predictive covariance, :78) is never needed - only its diagonal is - and
    is materialised only when M <= COV_PRED_MAX_M; `cov_meas_pred` (:81) only when M*N is small.
  * Failures raise instead of returning garbage: numpy.linalg.LinAlgError when K is not positive
    definite, IndexError when the acquisition contains NaN (the reference's own failure at :207).
  * Extra, not in the reference: `precision="fp32"` / `"i8"` / `"i8c"` (fp64 factorisation and means; the N^2 variance
    product as an fp32 / int8-sliced / three-digit int8 SCREEN).  In these modes EVERY acquisition call - LCB with any
    `explore`, EI - returns the index the fp64 kernels decide (screen + fp64 re-score of the survivors with THAT
    acquisition), `mean_func` is the fp64 kernels' bit for bit, and `cov_func` / `acq_func_eval` are the SCREEN's:
    |cov_func - fp64| <= SCREEN_SIGMA_TOL[precision] (5e-3 for fp32 and i8c, 2e-9 for i8; also in `last_screen`), so the
    arg-max of `acq_func_eval` itself can differ from the returned index when two candidates are closer than that.
    A caller that plots or post-processes these arrays at the reference's 1e-8 should stay with precision="fp64".
    The screens' tolerance is verified per call on every re-scored candidate and a strided sample, not proven for each
    candidate: the route that is exact by construction is `dense_outputs=False` (the prefix bound).
    Also extra: `expected_improvement(xi)`,
    `q_expected_improvement()`, `dense_outputs=False` (next point only: the dense attributes stay None and the acquisition
    calls go through the exact prefix bound), `kernel_params` may be preset (then no
    ARD search runs), optional multi-GPU candidate sharding when torch.distributed is initialised,
    `incremental=True` / `state_path=...` (append new observations to the previous factorisation in O(N^2)
    while the length scales stay the same, within one process or across jobs through a state file).
There is no CPU implementation behind this class.
"""
from __future__ import annotations

import numpy as np

from . import distributed as D
from .gp_device import JITTER_ASSEMBLY, JITTER_KERNEL, DeviceGP

COV_PRED_MAX_M = 4096          # cov_pred is M x M: 128 MiB at this size
COV_MEAS_PRED_MAX = 1 << 24    # entries of the (M, N) cross covariance kept for inspection
MAX_APPEND_ROWS = 64           # more new rows than this: a fresh factorisation is cheaper than row-by-row appends
MAX_APPENDED_COLUMNS = 256     # columns built by appends since the last full factorisation before a refresh is due
# |cov_func - fp64 sigma| of the screened precisions (asserted in tests/test_gpu_parity.py, test_gpu_i8.py, test_gpu_i8c.py)
SCREEN_SIGMA_TOL = {"fp32": 5e-3, "i8": 2e-9, "i8c": 5e-3}


def _plot_hooks():
    """The reference star-imports plot_utils (point_selector.py:4) and calls plot_ARD_LL{,_1d} from
    tune_kernel (:146,163).  If that module is importable (the DAG's working directory) use it."""
    try:
        import plot_utils  # type: ignore

        return getattr(plot_utils, "plot_ARD_LL", None), getattr(plot_utils, "plot_ARD_LL_1d", None)
    except Exception:  # noqa: BLE001 - plotting is a side output, never a reason to fail the step
        return None, None


class PointSelector:
    def __init__(self, device=None, verbose: bool = False, shard_candidates: bool = True, precision: str = "fp64",
                 incremental: bool = False, state_path=None, dense_outputs: bool = True, likelihood: str = "reference"):
        # attribute protocol of point_selector.py:15-40
        self.feature_domain = None
        self.predicted_pts = None
        self.measured_vals = []
        self.measured_pts = []
        self.mean_func = None
        self.cov_func = None
        self.acq_func_eval = None
        self.hyperparam_obj = []
        self.length_scales = None
        self.kernel_params = None
        self.gradient_steps = 0.001
        self.iteration = None
        self.name = None
        # cov_pred / cov_meas / cov_meas_pred (:38-40) are properties below: built on first access
        self._lazy = {}
        self._cov = {"cov_pred": None, "cov_meas": None, "cov_meas_pred": None}
        # build-specific
        self.nlogml = None
        self._device = device
        self._verbose = verbose
        self._shard = shard_candidates
        if precision not in ("fp64", "fp32", "i8", "i8c"):
            raise ValueError("precision must be 'fp64' (reference arithmetic), 'fp32', 'i8' or 'i8c' (fp64 factorisation, "
                             "means and decision; variance product screened in fp32 / in int8 slices / in three int8 digits)")
        self._precision = precision
        # likelihood="reference" (default): tune_kernel's grid holds the reference's float32 values, det underflow
        # included (point_selector.py:117-119: -inf beyond N ~ 100, where the search then returns its first cell);
        # "logdet" (not in the reference): fp64 grid with log det K from the Cholesky factor - finite at any N.
        if likelihood not in ("reference", "logdet"):
            raise ValueError("likelihood must be 'reference' (the reference's np.log(np.linalg.det(K))) or 'logdet'")
        self._likelihood = likelihood
        # dense_outputs=False (not in the reference): a caller that needs the next point only.  mean_func / cov_func /
        # acq_func_eval stay None and the acquisition calls return the same multi-index through the exact prefix bound
        # (DeviceGP.score_bound: fp64 branch and bound, the full pass when the bound does not separate the candidates).
        self._dense = bool(dense_outputs)
        self._gp = None
        self._mu_dev = self._sigma_dev = None
        self._cached = None  # (kind, p0, p1) -> (acq ndarray, flat index)
        self._preset_kernel_params = False
        self._ls_cells = None          # explicit [G x d] cell list for d > 2 (set_length_scale_cells)
        self.ard_sweeps = 2            # passes of the coordinate-wise search when length_scales holds d > 2 axes
        # consecutive iterations differ by one observed row (select_parameters.py:163,299)
        self._incremental = bool(incremental) or state_path is not None
        self._state_path = state_path
        self._inc = None               # (X, y, ls) of the factorisation held by self._gp
        self.last_update = None        # "factorise" | "append": what the last update_surrogate() did
        self.last_screen = None        # screened precisions: DeviceGP.last_screen of the last acquisition + sigma_abs_tol
        self._screen_ctx = None        # screened precisions: (candidate shard on the device, diag_add)

    # ------------------------------------------------------------------------------------------
    def _cov_get(self, name):
        make = self._lazy.pop(name, None)
        if make is not None:
            self._cov[name] = make()
        return self._cov[name]

    def _cov_set(self, name, value):
        self._lazy.pop(name, None)
        self._cov[name] = value

    cov_pred = property(lambda self: self._cov_get("cov_pred"), lambda self, v: self._cov_set("cov_pred", v),
                        doc="k(X*,X*) + 1e-6 I (point_selector.py:78); None above COV_PRED_MAX_M candidates")
    cov_meas = property(lambda self: self._cov_get("cov_meas"), lambda self, v: self._cov_set("cov_meas", v),
                        doc="k(X,X) + 1e-6 I (point_selector.py:79)")
    cov_meas_pred = property(lambda self: self._cov_get("cov_meas_pred"),
                             lambda self, v: self._cov_set("cov_meas_pred", v),
                             doc="k(X,X*).T (point_selector.py:81); None above COV_MEAS_PRED_MAX entries")

    def _log(self, *a):
        if self._verbose:
            print(*a)

    def _world(self):
        if not self._shard:
            return 1, 0
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
        return 1, 0

    def set_length_scale_cells(self, cells):
        """d > 2 (not in the reference, whose tune_kernel builds a 1-D or 2-D grid only, point_selector.py:122-163):
        an explicit list of length-scale vectors [G x d]; tune_kernel() evaluates the reference's likelihood
        (:111-120) in every cell on the GPU and keeps the FIRST minimum (np.argwhere(g == amin)[0], as :141,159)."""
        self.length_scales = None
        self._ls_cells = np.ascontiguousarray(np.asarray(cells, dtype=np.float64))
        if self._ls_cells.ndim != 2:
            raise ValueError("cells must be a [G x d] array of length-scale vectors")
        self._preset_kernel_params = False

    def set_kernel_params(self, kernel_params):
        """Preset length scales: update_surrogate() then skips the ARD grid search (needed for d > 2,
        where the reference's tune_kernel cannot run at all)."""
        self.kernel_params = np.asarray(kernel_params, dtype=np.float64)
        self._preset_kernel_params = True

    # ------------------------------------------------------------------------------------------
    def update_surrogate(self):
        """point_selector.py:42-102."""
        self.measured_pts = np.array(self.measured_pts)
        self.measured_vals = np.array(self.measured_vals)
        X = np.asarray(self.measured_pts, dtype=np.float64)
        y = np.asarray(self.measured_vals, dtype=np.float64)
        Xs = np.asarray(self.predicted_pts, dtype=np.float64)
        if self._gp is None:
            self._gp = DeviceGP(self._device)
        gp = self._gp

        ls = self._select_kernel_params(X)

        self._factorise_or_append(gp, X, y, ls)                           # :79, :89 (raises LinAlgError)

        M, N = len(Xs), len(X)
        diag_add = JITTER_KERNEL if Xs.shape == X.shape else 0.0          # :173 shape-coincidence quirk
        world, rank = self._world()
        lo, hi = D.shard_bounds(M, world, rank)
        if not self._dense:
            self.mean_func = self.cov_func = self.acq_func_eval = None
            self._mu_dev = self._sigma_dev = None
            self._cached = {}
            self._lo_hi = (lo, hi)
            self._select_only = (gp._dev(Xs[lo:hi]), diag_add)
            self._cov = {"cov_pred": None, "cov_meas": None, "cov_meas_pred": None}
            self._lazy = {"cov_meas": gp.cov_meas_host}
            self.measured_pts = self.measured_pts.tolist()
            self.measured_vals = self.measured_vals.tolist()
            return
        if self._precision in ("fp32", "i8", "i8c"):   # screened variance product (fp32: BASELINE config 4's mode), fp64 decision
            score = {"fp32": gp.score_f32, "i8": gp.score_i8, "i8c": gp.score_i8c}[self._precision]
            self._screen_ctx = (gp._dev(Xs[lo:hi]), diag_add)
            res = score(self._screen_ctx[0], acquisition="lcb", explore=4.0, dense=True, idx_offset=lo, diag_add=diag_add)
            self.last_screen = dict(gp.last_screen, sigma_abs_tol=SCREEN_SIGMA_TOL[self._precision])
        else:
            self._screen_ctx = None
            res = gp.score(Xs[lo:hi], acquisition="lcb", explore=4.0, dense=True, idx_offset=lo, diag_add=diag_add)
        self._mu_dev, self._sigma_dev = res.mu, res.sigma
        best = D.allreduce_argmax(res.best_val, res.best_idx, res.nan_count)
        # sharded: the three dense arrays are gathered on the device (one collective), then copied to the host once
        mu, sigma, acq = (t.cpu().numpy() for t in D.gather_concat_tensors([res.mu, res.sigma, res.acq], M))
        fd = [int(v) for v in self.feature_domain]
        self.mean_func = mu.reshape(fd)                                   # :97
        self.cov_func = sigma.reshape(fd)                                 # :98 (a standard deviation)
        self._cached = {("lcb", 4.0, 0.0): (acq.reshape(fd), best)}
        self._lo_hi = (lo, hi)

        # The three covariance attributes of point_selector.py:38-40 are read by nobody on the reference's call path
        # (select_parameters.py reads mean_func / cov_func / acq_func_eval only): they are copied out of the device
        # on first access instead of on every call (the 2,500 x 2,500 cov_pred copy was 6.4 of 7.9 ms of a C1 step).
        self._cov = {"cov_pred": None, "cov_meas": None, "cov_meas_pred": None}
        self._lazy = {"cov_meas": gp.cov_meas_host}
        if M * N <= COV_MEAS_PRED_MAX:
            self._lazy["cov_meas_pred"] = lambda: gp.cov_meas_pred_host(Xs, diag_add)
        if M <= COV_PRED_MAX_M:
            self._lazy["cov_pred"] = lambda: gp.kxx_host(Xs, ls, JITTER_KERNEL, JITTER_ASSEMBLY)

        self.measured_pts = self.measured_pts.tolist()                    # :101-102
        self.measured_vals = self.measured_vals.tolist()

    def _select_kernel_params(self, X) -> np.ndarray:
        """point_selector.py:60-73: preset, ARD grid search (n > 1) or the middle of each length-scale axis."""
        if self._preset_kernel_params:
            pass
        elif len(X[:, 0]) > 1:                                           # :60
            self.tune_kernel()
        elif self._ls_cells is not None:                                 # one observation: the middle cell of the list
            self.kernel_params = np.array(self._ls_cells[len(self._ls_cells) // 2])
        elif X.shape[1] > 2:                                             # the middle of every axis, as :63-73 does
            self.kernel_params = np.array([np.asarray(a, dtype=np.float64)[len(a) // 2] for a in self.length_scales])
        else:                                                            # :63-73
            if len(self.length_scales) == 2:
                a1, a2 = self.length_scales[0], self.length_scales[1]
                self.kernel_params = np.array([a1[len(a1) // 2], a2[len(a2) // 2]])
            else:
                self.kernel_params = np.array([self.length_scales[len(self.length_scales) // 2]])
        return np.asarray(self.kernel_params, dtype=np.float64).reshape(-1)

    def _factorise_or_append(self, gp, X, y, ls):
        """Full factorisation, or - with incremental=True / a state file - O(N^2) appends when the new data are
        the old data plus a few rows and the length scales are bit-identical to those of the held factors."""
        import os

        appended = False
        world, rank = self._world()
        if self._incremental:
            if self._inc is None and self._state_path is not None and os.path.exists(self._state_path):
                try:
                    gp.load_state(self._state_path)
                    self._inc = (*gp.observations_host(), np.array(gp.ls_h))   # in the caller's order
                except Exception as exc:  # noqa: BLE001 - an unreadable state file only costs the shortcut
                    self._log(f"state file {self._state_path!r} ignored: {exc}")
                    self._inc = None
            can_append = False
            if self._inc is not None:
                X0, y0, ls0 = self._inc
                n0 = len(X0)
                can_append = (n0 < len(X) <= n0 + MAX_APPEND_ROWS and gp.N == n0 and X0.shape[1:] == X.shape[1:]
                              and gp.n_appended + (len(X) - n0) <= MAX_APPENDED_COLUMNS
                              and ls0.shape == ls.shape and np.array_equal(ls0, ls)
                              and gp.jitter1 == JITTER_KERNEL and gp.jitter2 == JITTER_ASSEMBLY
                              and np.array_equal(X[:n0], X0) and np.array_equal(y[:n0], y0)
                              # (a permuted factorisation cannot carry the N == M quirk, which is keyed on the arrival index)
                              and (gp.perm is None or np.shape(self.predicted_pts) != np.shape(X)))
            # the route is a collective decision: a rank that appends while another refactorises would hold factors
            # that differ at rounding level, and the lowest-index tie rule across shards assumes identical factors
            if D.all_agree(can_append) if self._shard else can_append:
                try:
                    for i in range(n0, len(X)):
                        gp.append(X[i], y[i])
                    appended = True
                except np.linalg.LinAlgError:
                    pass  # numerically singular through the update: the full route decides (and raises if so)
                if self._shard:
                    appended = D.all_agree(appended)
        if not appended:
            # next point only (dense_outputs=False): the acquisition calls prune by the first observations of the
            # factorisation (DeviceGP.score_bound), so those are chosen to cover the region whatever the order of the
            # history; not with the N == M quirk, which is keyed on the arrival index (:173)
            fps = not self._dense and np.shape(self.predicted_pts) != np.shape(X)
            gp.factorise(X, y, ls, JITTER_KERNEL, JITTER_ASSEMBLY, check=True, order="fps" if fps else "arrival")
            if fps and self._shard and self._world()[0] > 1:
                # the order is part of the factorisation: if the selection fell back to the arrival order on ANY rank, every
                # rank refactorises in arrival order (identical factors on all shards: the cross-shard tie rule needs them)
                if not D.all_agree(not gp.order_fell_back()):
                    gp.factorise(X, y, ls, JITTER_KERNEL, JITTER_ASSEMBLY, check=True, order="arrival")
        self.last_update = "append" if appended else "factorise"
        if self._incremental:
            self._inc = (X.copy(), y.copy(), ls.copy())
            if self._state_path is not None and rank == 0:   # one writer; DeviceGP.save_state renames atomically
                gp.save_state(self._state_path)

    @staticmethod
    def _gather(local: np.ndarray, M: int, world: int) -> np.ndarray:
        return D.gather_concat(local, M)

    def _nlml_cells(self, X, y, cells) -> np.ndarray:
        """The likelihood of every grid cell; with several ranks each evaluates a contiguous block of cells
        (independent factorisations) and the float32 values are concatenated on every rank."""
        world, rank = self._world()
        kw = {} if self._likelihood == "reference" else {"likelihood": self._likelihood}
        if world == 1 or len(cells) < world:
            return self._gp.nlml_grid(X, y, cells, **kw)
        lo, hi = D.shard_bounds(len(cells), world, rank)
        return D.gather_concat(self._gp.nlml_grid(X, y, cells[lo:hi], **kw), len(cells))

    # ------------------------------------------------------------------------------------------
    def tune_kernel(self):
        """ARD grid search, point_selector.py:104-163: float32 nlml grid on the GPU, first row-major
        minimum on the host (np.argwhere(g == amin)[0]; NaN -> IndexError, as in the reference)."""
        X = np.asarray(self.measured_pts, dtype=np.float64)
        y = np.asarray(self.measured_vals, dtype=np.float64)
        plot2, plot1 = _plot_hooks()
        if self._gp is None:
            self._gp = DeviceGP(self._device)
        if self._ls_cells is not None:
            # explicit cell list (any d): the reference's likelihood in every cell, first minimum wins
            if self._ls_cells.shape[1] != X.shape[1]:
                raise ValueError(f"length-scale cells have {self._ls_cells.shape[1]} columns, the observations {X.shape[1]}")
            nlogml = self._nlml_cells(X, y, self._ls_cells)
            self.kernel_params = np.array(self._ls_cells[np.argwhere(nlogml == np.amin(nlogml))[0][0]])
            self.nlogml = nlogml
            return
        if X.shape[1] > 2:
            # d > 2: `length_scales` holds one search axis per feature.  The full Cartesian grid has prod(G_k) cells, so
            # the axes are searched one at a time from the middle of every axis (the reference's choice when it cannot
            # tune, :63-73), first minimum per axis, `ard_sweeps` passes.
            axes = [np.asarray(a, dtype=np.float64).reshape(-1) for a in self.length_scales]
            if len(axes) != X.shape[1]:
                raise ValueError(f"length_scales must hold one axis per feature ({X.shape[1]}), got {len(axes)}")
            ls = np.array([a[len(a) // 2] for a in axes])
            grids = [None] * len(axes)
            for _ in range(int(self.ard_sweeps)):
                for k, a in enumerate(axes):
                    cells = np.tile(ls, (len(a), 1))
                    cells[:, k] = a
                    g = self._nlml_cells(X, y, cells)
                    grids[k] = g
                    ls[k] = a[np.argwhere(g == np.amin(g))[0][0]]
            self.kernel_params = ls
            self.nlogml = grids
            return
        if len(self.length_scales) == 2:                                  # :122
            axis1 = np.asarray(self.length_scales[0], dtype=np.float64)
            axis2 = np.asarray(self.length_scales[1], dtype=np.float64)
            cells = np.stack(np.meshgrid(axis1, axis2, indexing="ij"), -1).reshape(-1, 2)
            nlogml = self._nlml_cells(X, y, cells).reshape(len(axis1), len(axis2))
            min_idx = np.argwhere(nlogml == np.amin(nlogml))[0]           # :141
            self.kernel_params = np.array([axis1[min_idx[0]], axis2[min_idx[1]]])
            self.nlogml = nlogml
            if plot2 is not None:
                try:
                    plot2(nlogml, self.kernel_params, self.length_scales, self.name, self.iteration)
                except Exception:  # noqa: BLE001
                    pass
        else:
            axis = np.asarray(self.length_scales, dtype=np.float64)
            nlogml = self._nlml_cells(X, y, axis.reshape(-1, 1))
            min_idx = np.argwhere(nlogml == np.amin(nlogml))[0]           # :159
            self.kernel_params = np.array([axis[min_idx]])                # shape (1, 1), as at :161
            self.nlogml = nlogml
            if plot1 is not None:
                try:
                    plot1(nlogml, self.kernel_params, self.length_scales, self.name, self.iteration)
                except Exception:  # noqa: BLE001
                    pass

    # ------------------------------------------------------------------------------------------
    def _finish(self, key, kind, **kw):
        fd = [int(v) for v in self.feature_domain]
        if not self._dense:
            if key not in self._cached:
                Xd, diag_add = self._select_only
                res = self._gp.score_bound(Xd, acquisition=kind, idx_offset=self._lo_hi[0], diag_add=diag_add, **kw)
                self._cached[key] = (None, D.allreduce_argmax(res.best_val, res.best_idx, res.nan_count))
        elif key not in self._cached:
            lo, hi = self._lo_hi
            if self._screen_ctx is not None:
                # screened precision: the stored sigma is the screen's, so the decision for THIS acquisition is made the
                # way the cached LCB(4) one was - a screen pass with it, then the fp64 kernels on every survivor
                score = {"fp32": self._gp.score_f32, "i8": self._gp.score_i8, "i8c": self._gp.score_i8c}[self._precision]
                Xd, diag_add = self._screen_ctx
                res = score(Xd, acquisition=kind, dense=True, idx_offset=lo, diag_add=diag_add, **kw)
                self.last_screen = dict(self._gp.last_screen, sigma_abs_tol=SCREEN_SIGMA_TOL[self._precision])
            else:
                res = self._gp.acquisition_on_posterior(self._mu_dev, self._sigma_dev, acquisition=kind,
                                                        idx_offset=lo, **kw)
            best = D.allreduce_argmax(res.best_val, res.best_idx, res.nan_count)
            acq = D.gather_concat_tensors([res.acq], int(np.prod(fd)))[0].cpu().numpy()
            self._cached[key] = (acq.reshape(fd), best)
        acq, (best_val, best_idx, nan_count) = self._cached[key]
        self.acq_func_eval = acq
        if nan_count > 0 or best_idx >= int(np.prod(fd)):
            # the reference: amax is NaN, the comparison is empty, [0] raises (point_selector.py:207)
            raise IndexError("index 0 is out of bounds for axis 0 with size 0 (acquisition contains NaN)")
        return np.array(np.unravel_index(best_idx, fd), dtype=np.int64)

    def lower_confidence_bound(self, explore=4):
        """point_selector.py:197-207: acq = explore*sigma - mu; first row-major arg-max as a multi-index."""
        if self._cached is None:
            raise RuntimeError("call update_surrogate() first")
        return self._finish(("lcb", float(explore), 0.0), "lcb", explore=float(explore))

    def q_expected_improvement(self, n_samples=512, seed=7, xi=0.0):
        """Not in the reference: q = 8 Monte-Carlo Expected Improvement.  The candidates are grouped
        consecutively (row-major order of `predicted_pts`) into batches of 8; returns the (8, ndim) multi-indices
        of the first batch with the largest qEI, and leaves the per-batch values in `acq_func_eval` (1-D).
        Fixed base samples: default_rng(seed).standard_normal((n_samples, 8)).  Batches are sharded over the
        ranks like single candidates are."""
        if self._cached is None:
            raise RuntimeError("call update_surrogate() first")
        fd = [int(v) for v in self.feature_domain]
        M = int(np.prod(fd))
        if M % 8:
            raise ValueError("q_expected_improvement needs a candidate count that is a multiple of 8")
        Z = np.random.default_rng(seed).standard_normal((int(n_samples), 8))
        f_best = float(np.min(np.asarray(self.measured_vals, dtype=np.float64)))
        Xs = np.asarray(self.predicted_pts, dtype=np.float64)
        world, rank = self._world()
        blo, bhi = D.shard_bounds(M // 8, world, rank)          # whole batches per rank, contiguous
        res = self._gp.score_qei(Xs[blo * 8: bhi * 8], Z, f_best, xi=float(xi), dense=True, batch_offset=blo)
        best_val, best_idx, nan_count = D.allreduce_argmax(res.best_val, res.best_idx, res.nan_count)
        qei = D.gather_concat_tensors([res.acq], M // 8)[0].cpu().numpy()
        self.acq_func_eval = qei
        if nan_count > 0 or best_idx >= M // 8:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0 (acquisition contains NaN)")
        res.best_idx = best_idx
        flat = res.best_idx * 8 + np.arange(8)
        return np.stack(np.unravel_index(flat, fd), axis=1).astype(np.int64)

    def expected_improvement(self, xi=0.0):
        """Not in the reference (docs/README.md:363-365 'future work'): EI for minimisation,
        f_best = min(measured_vals)."""
        if self._cached is None:
            raise RuntimeError("call update_surrogate() first")
        f_best = float(np.min(np.asarray(self.measured_vals, dtype=np.float64)))
        return self._finish(("ei", f_best, float(xi)), "ei", f_best=f_best, xi=float(xi))
