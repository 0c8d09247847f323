"""CPU oracle for the GP acquisition path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module; nothing under `bayesian_optimisation_amd/` does, and the product path raises when the HIP
library is missing instead of falling back to anything here.

It restates, in NumPy fp64, the algorithm of the reference class `PointSelector`
(/root/reference/point_selector.py:13-207).  Parity is PINNED: `tests/test_oracle_golden.py` checks
every function below against the golden vectors in `tests/golden/*.npz`, which were produced by
running the reference itself (`tests/golden/make_golden.py`).

Two posterior routes are given:
  * `posterior_literal`  - follows the reference's arithmetic (explicit `inv`), but keeps only the
    diagonal of the M x M predictive covariance the reference forms in full
    (point_selector.py:78,91,98).  O(M N^2) work, O(M N) memory per chunk.
  * `posterior_chol`     - Cholesky + triangular solve, chunked over candidates and BLAS-threaded.
    Same mathematics; this is the route that scales to BASELINE.json's candidate counts and the one
    `bench.py` times as the CPU baseline ("port").
Expected Improvement is NOT in the reference (docs/README.md:363-365 lists it as future work); the
closed form here is this build's own definition - parity for EI is pinned
only by this restatement.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
from scipy.special import ndtr

JITTER_KERNEL = 1e-4   # point_selector.py:193  (added inside kernel_rbf when shapes coincide)
JITTER_ASSEMBLY = 1e-6  # point_selector.py:78-79 (added when the covariance blocks are assembled)
# diagonal of cov_pred exactly as the reference's float arithmetic leaves it: (exp(0)+1e-4)+1e-6
PRIOR_VAR = (1.0 + JITTER_KERNEL) + JITTER_ASSEMBLY


def kernel_rbf(x1: np.ndarray, x2: np.ndarray, ls: np.ndarray) -> np.ndarray:
    """ARD squared-exponential Gram matrix, point_selector.py:166-195.

    Operation order as in the reference: subtract, square, divide by ls**2, sum over features,
    times -0.5, exp.  Jitter rule: 1e-4 on the diagonal iff x1.shape == x2.shape (:173,191-193) -
    keyed on shape equality, so K(X, X*) gets it too when N == M.
    """
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    jitter = x1.shape == x2.shape
    ls = np.asarray(ls, dtype=np.float64).reshape(-1)  # tuned 1-D kernel_params arrive as (1, 1)
    acc = np.zeros((x1.shape[0], x2.shape[0]))
    d = x1.shape[1]
    if d <= 8 and x1.shape[0] * x2.shape[0] * d <= 1 << 24:
        # small case: the reference's own broadcast, including NumPy's reduction order over axis 2
        dist = (x1[:, None, :] - x2[None, :, :]) ** 2
        acc = np.sum(dist / ls ** 2, axis=2)
    else:
        for k in range(d):  # same values up to the summation order over features
            acc += (x1[:, k, None] - x2[None, :, k]) ** 2 / ls[k] ** 2
    rbf = np.exp(-0.5 * acc)
    if jitter:
        return rbf + JITTER_KERNEL * np.eye(x1.shape[0])
    return rbf


def nlml_grid(X: np.ndarray, y: np.ndarray, length_scales) -> np.ndarray:
    """float32 grid of -log marginal likelihood, point_selector.py:111-138 / :150-156.

    K carries the 1e-4 jitter only (no 1e-6); explicit inv and det as in the reference, so `det`
    underflow (-> -inf cells) and negative `det` (-> NaN) are reproduced.
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(X)
    if len(length_scales) == 2:
        a1, a2 = length_scales[0], length_scales[1]
        out = np.zeros((len(a1), len(a2)), dtype=np.float32)
        cells = [((i, j), np.array([a1[i], a2[j]])) for i in range(len(a1)) for j in range(len(a2))]
    else:
        out = np.zeros(len(length_scales), dtype=np.float32)
        cells = [((i,), np.array([length_scales[i]])) for i in range(len(length_scales))]
    with np.errstate(all="ignore"):
        for ij, kp in cells:
            rbf = kernel_rbf(X, X, kp)
            inv = np.linalg.inv(rbf)
            det = np.linalg.det(rbf)
            out[ij] = 0.5 * (y.T @ inv @ y + np.log(det) + n * np.log(2 * np.pi))
    return out


def nlml_cells(X: np.ndarray, y: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """eval_log_marginal (point_selector.py:111-120) for an explicit list of length-scale vectors [G x d], any d:
    the same arithmetic as `nlml_grid` per cell (1e-4 jitter only, explicit inv and det, float32 result).  The
    reference's tune_kernel only builds such cells for d = 1 and d = 2; for d > 2 this is its formula applied to the
    cells the build searches."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.float64).reshape(-1, X.shape[1])
    n = len(X)
    out = np.zeros(len(cells), dtype=np.float32)
    with np.errstate(all="ignore"):
        for g, kp in enumerate(cells):
            rbf = kernel_rbf(X, X, kp)
            inv = np.linalg.inv(rbf)
            det = np.linalg.det(rbf)
            out[g] = 0.5 * (y.T @ inv @ y + np.log(det) + n * np.log(2 * np.pi))
    return out


def nlml_cells_stable(X: np.ndarray, y: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """The same quantity from a Cholesky factorisation (log det = 2 sum log L_ii, y^T K^-1 y = |L^-1 y|^2), with the
    reference's det underflow applied afterwards (log(exp(logdet)): NumPy's det IS sign * exp(logdet)).  Equal to
    `nlml_cells` wherever LAPACK's LU determinant is accurate; the comparison value for large N, where the explicit
    inverse of the reference loses digits first."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.float64).reshape(-1, X.shape[1])
    n = len(X)
    out = np.zeros(len(cells))
    with np.errstate(all="ignore"):
        for g, kp in enumerate(cells):
            try:
                L = np.linalg.cholesky(kernel_rbf(X, X, kp))
            except np.linalg.LinAlgError:
                out[g] = np.nan
                continue
            z = sla.solve_triangular(L, y, lower=True, check_finite=False)
            logdet = 2.0 * np.sum(np.log(np.diag(L)))
            out[g] = 0.5 * (z @ z + np.log(np.exp(logdet)) + n * np.log(2 * np.pi))
    return out


def nlml_cells_logdet(X: np.ndarray, y: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """The likelihood of `nlml_cells` WITHOUT the reference's determinant (a documented departure, not a restatement of
    point_selector.py:118): 0.5 (|L^-1 y|^2 + 2 sum log L_ii + N log 2 pi) from a Cholesky factor of K = k(X,X) + 1e-4 I,
    fp64, finite at any N; NaN where K is not positive definite.  The checker of the build's likelihood="logdet" mode.
    Mathematically equal to the reference's value wherever det(K) neither under- nor overflows."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.float64).reshape(-1, X.shape[1])
    n = len(X)
    out = np.zeros(len(cells))
    for g, kp in enumerate(cells):
        try:
            L = np.linalg.cholesky(kernel_rbf(X, X, kp))
        except np.linalg.LinAlgError:
            out[g] = np.nan
            continue
        z = sla.solve_triangular(L, y, lower=True, check_finite=False)
        out[g] = 0.5 * (z @ z + 2.0 * np.sum(np.log(np.diag(L))) + n * np.log(2 * np.pi))
    return out


def coordinate_search(X, y, axes, sweeps: int = 2, nlml=None):
    """Length-scale search for d > 2 (not in the reference, whose tune_kernel handles one or two axes only): start
    from the middle of every axis (the reference's own choice when it cannot tune, point_selector.py:63-73), then for
    each axis in turn evaluate the likelihood with that coordinate running over its grid and keep the FIRST minimum
    (np.argwhere(g == amin)[0], as :141,159), `sweeps` passes over the axes.  Returns (kernel_params, last grids)."""
    axes = [np.asarray(a, dtype=np.float64) for a in axes]
    ls = np.array([a[len(a) // 2] for a in axes])
    grids = [None] * len(axes)
    for _ in range(sweeps):
        for k, a in enumerate(axes):
            cells = np.tile(ls, (len(a), 1))
            cells[:, k] = a
            g = (nlml or nlml_cells)(X, y, cells)     # nlml=nlml_cells_logdet: the build's second likelihood mode
            grids[k] = g
            ls[k] = a[int(first_min_index(g)[0])]
    return ls, grids


def first_min_index(grid: np.ndarray) -> np.ndarray:
    """np.argwhere(g == np.amin(g))[0]  (point_selector.py:141,159): first row-major minimum;
    a NaN anywhere makes amin NaN, the comparison empty, and the [0] an IndexError."""
    return np.argwhere(grid == np.amin(grid))[0]


def select_length_scales(X: np.ndarray, y: np.ndarray, length_scales):
    """The branch at point_selector.py:60-73 plus tune_kernel's winner (:141-143, :159-161).
    Returns (kernel_params, nlogml_grid_or_None)."""
    X = np.asarray(X)
    if len(X[:, 0]) > 1:
        g = nlml_grid(X, y, length_scales)
        idx = first_min_index(g)
        if len(length_scales) == 2:
            return np.array([length_scales[0][idx[0]], length_scales[1][idx[1]]]), g
        # 1-D quirk (:161): indexing with the length-1 index ARRAY makes kernel_params shape (1, 1)
        return np.array([length_scales[idx]]), g
    if len(length_scales) == 2:
        a1, a2 = length_scales[0], length_scales[1]
        return np.array([a1[len(a1) // 2], a2[len(a2) // 2]]), None
    return np.array([length_scales[len(length_scales) // 2]]), None


def posterior_literal(X, y, Xs, ls, chunk: int = 8192):
    """mu and sigma at every candidate following point_selector.py:78-98 with explicit inv.

    sigma = sqrt(|diag(K** - K*x inv K*x^T)|): note the abs (:98) and that the reference's
    "cov_func" is a standard deviation.  Only the diagonal is formed.
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    cov_meas = kernel_rbf(X, X, ls) + JITTER_ASSEMBLY * np.eye(len(X))     # :79
    inv = np.linalg.inv(cov_meas)                                           # :89
    w = inv @ y
    M = len(Xs)
    mu = np.empty(M)
    sig = np.empty(M)
    same_shape = X.shape == Xs.shape
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        kmp = kernel_rbf(X, Xs[s:e], ls) if not same_shape else kernel_rbf(X, Xs, ls)[:, s:e]
        kmp = kmp.T                                                         # :81  (m, N)
        mu[s:e] = kmp @ w                                                   # :90
        t = inv @ kmp.T                                                     # :91
        var = PRIOR_VAR - np.einsum("mn,nm->m", kmp, t)
        sig[s:e] = np.sqrt(np.abs(var))                                     # :98
    return mu, sig


def factorise(X, y, ls):
    """K + 1.01e-4 I = L L^T; alpha = K^-1 y.  Raises numpy.linalg.LinAlgError when K is not
    positive definite (the reference's inv raises LinAlgError for an exactly singular K, :89)."""
    K = kernel_rbf(X, X, ls) + JITTER_ASSEMBLY * np.eye(len(X))
    L = np.linalg.cholesky(K)
    alpha = sla.cho_solve((L, True), np.asarray(y, dtype=np.float64))
    return K, L, alpha


def posterior_chol(X, y, Xs, ls, chunk: int = 16384, L=None, alpha=None, variance_dtype=np.float64):
    """Cholesky route: v = L^-1 k*, sigma^2 = c - |v|^2, mu = k* . alpha; chunked over candidates.
    variance_dtype=np.float32: the restatement of BASELINE config 4's arithmetic - fp64 factorisation,
    K* entries and mean; L and K* rounded to fp32 for the N^2-per-candidate triangular solve and |v|^2 - i.e. what the
    fp32 screen computes (csrc/posterior_f32.hip), on the CPU."""
    X = np.asarray(X, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    if L is None:
        _, L, alpha = factorise(X, y, ls)
    Lv = L if variance_dtype == np.float64 else np.asarray(L, dtype=variance_dtype)
    M = len(Xs)
    mu = np.empty(M)
    sig = np.empty(M)
    same_shape = X.shape == Xs.shape
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        kmp = kernel_rbf(X, Xs[s:e], ls) if not same_shape else kernel_rbf(X, Xs, ls)[:, s:e]
        mu[s:e] = kmp.T @ alpha
        v = sla.solve_triangular(Lv, kmp.astype(variance_dtype, copy=False), lower=True, check_finite=False)
        sig[s:e] = np.sqrt(np.abs(PRIOR_VAR - np.einsum("nm,nm->m", v, v).astype(np.float64)))
    return mu, sig


def lcb(mu, sigma, explore=4):
    """point_selector.py:204: acq = explore * sigma - mu (maximised)."""
    return explore * sigma - mu


def expected_improvement(mu, sigma, f_best, xi=0.0):
    """EI for MINIMISATION (this build's definition):
    imp = f_best - mu - xi; z = imp/sigma; EI = imp*Phi(z) + sigma*phi(z); sigma == 0 -> max(imp,0)."""
    mu = np.asarray(mu, dtype=np.float64)
    sigma = np.asarray(sigma, dtype=np.float64)
    imp = f_best - mu - xi
    with np.errstate(divide="ignore", invalid="ignore"):
        z = imp / sigma
        ei = imp * ndtr(z) + sigma * np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)
    return np.where(sigma > 0.0, ei, np.maximum(imp, 0.0))


def argmax_first(acq: np.ndarray) -> np.ndarray:
    """np.argwhere(acq == np.amax(acq))[0]  (point_selector.py:207): multi-index of the first
    row-major maximum; NaN anywhere -> IndexError."""
    return np.argwhere(acq == np.amax(acq))[0]


def select_next(X, y, Xs, feature_domain, length_scales=None, kernel_params=None, explore=4,
                route="literal", acquisition="lcb", xi=0.0):
    """update_surrogate() + lower_confidence_bound() in one call (point_selector.py:42-102,197-207).
    Returns dict(kernel_params, nlogml, mean_func, cov_func, acq_func_eval, index)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    nlogml = None
    if kernel_params is None:
        kernel_params, nlogml = select_length_scales(X, y, length_scales)
    post = posterior_literal if route == "literal" else posterior_chol
    mu, sig = post(X, y, Xs, kernel_params)
    fd = [int(v) for v in feature_domain]
    mu = mu.reshape(fd)
    sig = sig.reshape(fd)
    if acquisition == "lcb":
        acq = lcb(mu, sig, explore)
    else:
        acq = expected_improvement(mu, sig, float(np.min(y)), xi)
    return dict(kernel_params=np.asarray(kernel_params), nlogml=nlogml, mean_func=mu, cov_func=sig,
                acq_func_eval=acq, index=argmax_first(acq))


def qei_base_samples(n_samples: int = 512, q: int = 8, seed: int = 7) -> np.ndarray:
    """Fixed base samples of the Monte-Carlo qEI: default_rng(7).standard_normal((512, 8))."""
    return np.random.default_rng(seed).standard_normal((n_samples, q))


def qei_mc(X, y, Xs, ls, Z, f_best=None, xi=0.0, q: int = 8):
    """q-point Monte-Carlo Expected Improvement over consecutive batches of q candidates (this build's
    definition, not in the reference): joint posterior of a batch from the Cholesky route,
    Sigma_b = K_bb - V_b^T V_b with the reference's prior diagonal, L_b = chol(Sigma_b),
    qEI_b = mean_s max(0, max_j (f_best - xi - (mu_b + L_b z_s)_j)).  Returns the (M/q,) array."""
    X = np.asarray(X, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if f_best is None:
        f_best = float(np.min(y))
    _, L, alpha = factorise(X, y, ls)
    M = len(Xs)
    assert M % q == 0
    out = np.empty(M // q)
    for b in range(M // q):
        P = Xs[b * q:(b + 1) * q]
        ksx = kernel_rbf(X, P, ls)                      # (N, q); no jitter (shapes differ unless N == q)
        if X.shape == P.shape:
            ksx = ksx - JITTER_KERNEL * np.eye(len(X))  # the batch never triggers the reference's N == M quirk
        mu = ksx.T @ alpha
        V = sla.solve_triangular(L, ksx, lower=True, check_finite=False)
        lsq = np.asarray(ls, dtype=np.float64).reshape(-1)
        d2 = np.zeros((q, q))
        for k in range(P.shape[1]):
            d2 += (P[:, k, None] - P[None, :, k]) ** 2 / lsq[k] ** 2
        Kbb = np.exp(-0.5 * d2)
        Kbb[np.arange(q), np.arange(q)] = PRIOR_VAR
        Sig = Kbb - V.T @ V
        Lb = np.linalg.cholesky(Sig)
        f = mu[None, :] + Z @ Lb.T                      # (S, q)
        out[b] = np.mean(np.maximum(0.0, np.max(f_best - xi - f, axis=1)))
    return out
