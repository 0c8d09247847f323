"""Seeded synthetic inputs for the GP acquisition path (SURVEY.md §8(d) recipe).

Not part of the reference: the reference ships no observation data
(`measured_points/*.npy` is git-ignored, /root/reference/.gitignore:9), so tests and
bench.py build their own deterministic inputs here.  Everything is NumPy / CPU torch
(`SobolEngine`), so the same arrays are produced in the build container and on the GPU box.
"""
from __future__ import annotations

import numpy as np


def sobol_points(n_skip: int, n: int, d: int) -> np.ndarray:
    """`n` points of the unscrambled Sobol sequence in [0,1]^d after skipping `n_skip`.

    X is the first N points and X* the next M of the same engine, so X and X* never share a row.
    """
    import torch

    eng = torch.quasirandom.SobolEngine(d, scramble=False)
    if n_skip:
        eng.fast_forward(n_skip)
    return eng.draw(n, dtype=torch.float64).numpy().copy()


def ard_length_scales(d: int) -> np.ndarray:
    """Anisotropic ARD length scales: geomspace(0.2, 2.0, d)."""
    return np.geomspace(0.2, 2.0, d)


def rff_objective(X: np.ndarray, ls: np.ndarray, n_features: int = 256,
                  seed: int = 1234, noise: float = 1e-2, noise_seed: int = 4321) -> np.ndarray:
    """Random-Fourier-feature draw of a zero-mean ARD-SE GP at X, unit variance, plus noise."""
    rng = np.random.default_rng(seed)
    d = X.shape[1]
    omega = rng.standard_normal((n_features, d)) / ls[None, :]
    phase = rng.uniform(0.0, 2.0 * np.pi, n_features)
    w = rng.standard_normal(n_features)
    f = np.sqrt(2.0 / n_features) * (np.cos(X @ omega.T + phase[None, :]) @ w)
    f = f / max(np.std(f), 1e-12)
    return f + noise * np.random.default_rng(noise_seed).standard_normal(X.shape[0])


def make_problem(N: int, M: int, d: int):
    """(X, y, Xs, ls) for one synthetic BO step of BASELINE.json's configs 2-4."""
    ls = ard_length_scales(d)
    X = sobol_points(0, N, d)
    Xs = sobol_points(N, M, d)
    y = rff_objective(X, ls)
    return X, y, Xs, ls
