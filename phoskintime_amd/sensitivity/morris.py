"""Morris one-at-a-time screening: trajectory sampler and elementary-effects analyser (host side, numpy).

The reference delegates both to SALib 1.5.1 (``morris.sample`` / ``morris.analyze``, sensitivity/analysis.py:223,264), a
third-party package that is absent from this image and whose RNG stream cannot be regenerated.  Parity for this row is
therefore "same X in => same Y / EE / mu* / sigma out": ``analyze`` takes any sample matrix (SALib's included) and is
pinned by known-answer tests (tests/test_morris_cpu.py); ``sample`` implements the standard Morris (1991) / Campolongo
(2007) construction without SALib's optional trajectory optimisation."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def sample(problem: Dict, N: int, num_levels: int = 4, seed: Optional[int] = None) -> np.ndarray:
    """N trajectories of D + 1 points each -> [N * (D + 1), D], scaled to ``problem['bounds']``.

    Levels are the grid {0, 1/(p-1), ..., 1}; the jump is delta = p / (2 (p - 1)) (p = num_levels, even recommended);
    each trajectory starts at a random grid point from which every coordinate can move by +-delta inside [0, 1],
    and changes the coordinates one at a time in random order."""
    D = int(problem["num_vars"])
    p = int(num_levels)
    if p < 2:
        raise ValueError("num_levels must be >= 2")
    rng = np.random.default_rng(seed)
    delta = p / (2.0 * (p - 1))
    grid = np.arange(p) / (p - 1.0)
    X = np.empty((N * (D + 1), D))
    for r in range(N):
        # base point: a level from which +delta or -delta stays in [0, 1]
        sign = rng.choice([-1.0, 1.0], size=D)
        base = np.empty(D)
        for i in range(D):
            ok = grid[(grid + sign[i] * delta >= -1e-12) & (grid + sign[i] * delta <= 1 + 1e-12)]
            if ok.size == 0:                      # tiny p: flip the direction
                sign[i] = -sign[i]
                ok = grid[(grid + sign[i] * delta >= -1e-12) & (grid + sign[i] * delta <= 1 + 1e-12)]
            base[i] = rng.choice(ok)
        order = rng.permutation(D)
        x = base.copy()
        X[r * (D + 1)] = x
        for s, i in enumerate(order):
            x = x.copy()
            x[i] = x[i] + sign[i] * delta
            X[r * (D + 1) + s + 1] = x
    b = np.asarray(problem["bounds"], dtype=float)
    return b[:, 0] + np.clip(X, 0.0, 1.0) * (b[:, 1] - b[:, 0])


def elementary_effects(problem: Dict, X: np.ndarray, Y: np.ndarray, num_levels: int = 4) -> np.ndarray:
    """EE[r, i] of trajectory r for parameter i, with inputs rescaled to the unit cube (as SALib does)."""
    D = int(problem["num_vars"])
    X = np.asarray(X, float); Y = np.asarray(Y, float)
    if X.shape[0] % (D + 1) or X.shape[0] != Y.shape[0]:
        raise ValueError("X must hold N * (D + 1) rows and Y one value per row")
    N = X.shape[0] // (D + 1)
    b = np.asarray(problem["bounds"], float)
    width = np.where(b[:, 1] > b[:, 0], b[:, 1] - b[:, 0], 1.0)
    U = (X - b[:, 0]) / width
    U = U.reshape(N, D + 1, D); Yt = Y.reshape(N, D + 1)
    dU = np.diff(U, axis=1)                                     # [N, D, D]: step s changes exactly one coordinate
    dY = np.diff(Yt, axis=1)                                    # [N, D]
    which = np.argmax(np.abs(dU), axis=2)                       # coordinate moved at step s
    step = np.take_along_axis(dU, which[:, :, None], axis=2)[:, :, 0]
    ee = np.full((N, D), np.nan)
    rows = np.arange(N)[:, None]
    ee[rows, which] = dY / step
    return ee


def analyze(problem: Dict, X: np.ndarray, Y: np.ndarray, num_levels: int = 4, conf_level: float = 0.95,
            num_resamples: int = 100, scaled: bool = False, seed: Optional[int] = None) -> Dict:
    """mu, mu_star, sigma, mu_star_conf per parameter (the dict keys the reference reads from SALib's result,
    global_model/sensitivity.py:269-274).  ``scaled=True``: sigma-scaled elementary effects EE_i * std(x_i) / std(Y)
    (Sin & Gernaey 2009), the option the per-protein driver switches on (sensitivity/analysis.py:264)."""
    ee = elementary_effects(problem, X, Y, num_levels)
    if scaled:
        b = np.asarray(problem["bounds"], float)
        width = np.where(b[:, 1] > b[:, 0], b[:, 1] - b[:, 0], 1.0)
        sx = np.std((np.asarray(X, float) - b[:, 0]) / width, axis=0)
        sy = np.std(np.asarray(Y, float))
        ee = ee * (sx / sy if sy > 0 else 0.0)
    mu = np.nanmean(ee, axis=0)
    mu_star = np.nanmean(np.abs(ee), axis=0)
    sigma = np.nanstd(ee, axis=0, ddof=1) if ee.shape[0] > 1 else np.zeros(ee.shape[1])
    rng = np.random.default_rng(seed)
    N = ee.shape[0]
    if N > 1 and num_resamples > 0:
        idx = rng.integers(0, N, size=(num_resamples, N))
        res = np.nanmean(np.abs(ee)[idx], axis=1)              # [num_resamples, D]
        from scipy.stats import norm
        conf = norm.ppf(0.5 + conf_level / 2.0) * res.std(axis=0, ddof=1)
    else:
        conf = np.zeros(ee.shape[1])
    return {"names": list(problem.get("names", [f"x{i}" for i in range(ee.shape[1])])), "mu": mu, "mu_star": mu_star,
            "sigma": sigma, "mu_star_conf": conf, "elementary_effects": ee}
