"""Morris one-at-a-time screening: trajectory sampler and elementary-effects analyser (host side, numpy).

The reference delegates both to SALib 1.5.1 (``morris.sample`` / ``morris.analyze``, sensitivity/analysis.py:223,264), a
third-party package that is absent from this image and whose RNG stream cannot be regenerated.  Parity for this row is
therefore "same X in => same Y / EE / mu* / sigma out": ``analyze`` takes any sample matrix (SALib's included) and is
pinned by known-answer tests (tests/test_morris_cpu.py); ``sample`` implements the standard Morris (1991) / Campolongo
(2007) construction without SALib's optional trajectory optimisation."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np


@dataclass
class Draws:
    """The random part of a Morris design: N trajectories in D dimensions on the unit cube."""
    base: np.ndarray       # [N, D] start level of every coordinate (a grid value from which the jump stays inside [0, 1])
    sign: np.ndarray       # [N, D] +1 / -1 direction of the jump of coordinate i
    rank: np.ndarray       # [N, D] int32: coordinate i moves at step rank[i] + 1 (rank[r, :] is a permutation of 0..D-1)
    delta: float           # p / (2 (p - 1))


def draw(D: int, N: int, num_levels: int = 4, seed: Optional[int] = None) -> Draws:
    """All random choices of N trajectories at once (vectorised; KBs even for N * D ~ 10^5)."""
    p = int(num_levels)
    if p < 2:
        raise ValueError("num_levels must be >= 2")
    rng = np.random.default_rng(seed)
    delta = p / (2.0 * (p - 1))
    sign = rng.choice(np.array([-1.0, 1.0]), size=(N, D))
    # grid k / (p - 1): moving up needs k <= (p - 1)(1 - delta) = p/2 - 1, moving down needs k >= (p - 1) delta = p/2
    k_up_max = int(np.floor(p / 2.0 - 1.0 + 1e-9))
    k_dn_min = int(np.ceil(p / 2.0 - 1e-9))
    ku = rng.integers(0, k_up_max + 1, size=(N, D))
    kd = rng.integers(k_dn_min, p, size=(N, D))
    base = np.where(sign > 0, ku, kd) / (p - 1.0)
    order = rng.permuted(np.tile(np.arange(D, dtype=np.int32), (N, 1)), axis=1)       # order[r, s] = coordinate moved at step s + 1
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.tile(np.arange(D, dtype=np.int32), (N, 1)), axis=1)
    return Draws(base=base, sign=sign, rank=rank.astype(np.int32), delta=delta)


def build(draws: Draws, bounds) -> np.ndarray:
    """Sample matrix [N (D + 1), D] of a design: row s of trajectory r is base + sign * delta * [rank < s], scaled to ``bounds``."""
    N, D = draws.base.shape
    steps = np.arange(D + 1)[None, :, None]
    U = draws.base[:, None, :] + draws.sign[:, None, :] * draws.delta * (draws.rank[:, None, :] < steps)
    b = np.asarray(bounds, dtype=float)
    return (b[:, 0] + np.clip(U, 0.0, 1.0) * (b[:, 1] - b[:, 0])).reshape(N * (D + 1), D)


def sample(problem: Dict, N: int, num_levels: int = 4, seed: Optional[int] = None) -> np.ndarray:
    """N trajectories of D + 1 points each -> [N * (D + 1), D], scaled to ``problem['bounds']``.

    Levels are the grid {0, 1/(p-1), ..., 1}; the jump is delta = p / (2 (p - 1)) (p = num_levels, even recommended);
    each trajectory starts at a random grid point from which every coordinate can move by +-delta inside [0, 1],
    and changes the coordinates one at a time in random order."""
    return build(draw(int(problem["num_vars"]), N, num_levels, seed), problem["bounds"])


def sample_device(problem: Dict, N: int, num_levels: int = 4, seed: Optional[int] = None, device: Optional[int] = None):
    """The same design built in HBM (``pk_morris_build_batch``): only the draws cross PCIe.  Returns (X [N (D+1), D] GPU tensor,
    handle for ``elementary_effects_device``).  Bit-identical to ``sample`` with the same seed."""
    import torch
    from .. import batch
    ctx = batch.get_context(device)
    dev = torch.device("cuda", ctx.device)
    D = int(problem["num_vars"])
    d = draw(D, N, num_levels, seed)
    b = np.asarray(problem["bounds"], dtype=float)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    h = dict(draws=d, base=t(d.base, torch.float64), sign=t(d.sign, torch.float64), rank=t(d.rank, torch.int32), lb=t(b[:, 0], torch.float64),
             ub=t(b[:, 1], torch.float64), ctx=ctx, N=N, D=D)
    X = torch.empty((N * (D + 1), D), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.pk_morris_build_batch(ctx.handle, N, D, d.delta, h["base"].data_ptr(), h["sign"].data_ptr(), h["rank"].data_ptr(),
                                            h["lb"].data_ptr(), h["ub"].data_ptr(), X.data_ptr()))
    return X, h


def elementary_effects_device(h: dict, Y):
    """EE [N, D] (GPU tensor) of a design made by ``sample_device`` from its outputs Y [N (D + 1)] (GPU tensor)."""
    import torch
    ctx = h["ctx"]
    Y = Y.to(dtype=torch.float64).contiguous()
    if Y.numel() != h["N"] * (h["D"] + 1):
        raise ValueError("Y must hold one value per row of the design")
    EE = torch.empty((h["N"], h["D"]), dtype=torch.float64, device=Y.device)
    ctx.set_stream(torch.cuda.current_stream(Y.device).cuda_stream)
    ctx.check(ctx.lib.pk_morris_effects_batch(ctx.handle, h["N"], h["D"], h["draws"].delta, h["sign"].data_ptr(), h["rank"].data_ptr(),
                                              Y.data_ptr(), EE.data_ptr()))
    return EE


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


def analyze_effects(ee: np.ndarray, names=None, conf_level: float = 0.95, num_resamples: int = 100, seed: Optional[int] = None) -> Dict:
    """mu, mu_star, sigma, mu_star_conf from the elementary effects [N, D] (NaN = trajectory did not move that coordinate)."""
    ee = np.asarray(ee, float)
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
    return {"names": list(names if names is not None else [f"x{i}" for i in range(ee.shape[1])]), "mu": mu, "mu_star": mu_star,
            "sigma": sigma, "mu_star_conf": conf, "elementary_effects": ee}


def analyze(problem: Dict, X: np.ndarray, Y: np.ndarray, num_levels: int = 4, conf_level: float = 0.95,
            num_resamples: int = 100, scaled: bool = False, seed: Optional[int] = None) -> Dict:
    """mu, mu_star, sigma, mu_star_conf per parameter (the dict keys the reference reads from SALib's result,
    global_model/sensitivity.py:269-274) for ANY Morris sample matrix X (SALib's included).  ``scaled=True``: sigma-scaled elementary
    effects EE_i * std(x_i) / std(Y) (Sin & Gernaey 2009), the option the per-protein driver switches on (sensitivity/analysis.py:264)."""
    ee = elementary_effects(problem, X, Y, num_levels)
    if scaled:
        b = np.asarray(problem["bounds"], float)
        width = np.where(b[:, 1] > b[:, 0], b[:, 1] - b[:, 0], 1.0)
        sx = np.std((np.asarray(X, float) - b[:, 0]) / width, axis=0)
        sy = np.std(np.asarray(Y, float))
        ee = ee * (sx / sy if sy > 0 else 0.0)
    return analyze_effects(ee, problem.get("names"), conf_level, num_resamples, seed)
