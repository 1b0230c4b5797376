"""Per-gene parameter estimation with the reference's names and argument lists -- drop-in for ``paramest/normest.py``.

  worker_find_lambda(lam, gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data)      normest.py:22-115
  find_best_lambda(gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data, lambdas, ...) normest.py:118-165
  _curve_fit_multistart(gene, model_func, time_points, target_fit, base_p0, free_bounds, sigma, init_cond, num_psites, target,
                        n_starts, jitter_frac, maxfev, seed)                                                       normest.py:167-326
  normest(gene, pr_data, p_data, r_data, init_cond, num_psites, time_points, bounds, bootstraps, use_regularization) normest.py:328-563

What changes underneath: the reference runs one ``scipy.optimize.curve_fit`` per (lambda, weighting), per multistart point and per
bootstrap replicate -- a process pool of 10 for the lambda scan, the rest serial -- and every TRF iteration of every fit calls
``solve_ode`` 1 + P times.  Here each of the three phases is ONE lockstep batch of bounded Levenberg-Marquardt fits on the GPU
(``paramest.multistart.fit_rows_batch``: Jacobians from the forward-sensitivity kernel, one launch per iteration for ALL rows).
What stays: the start point (``np.random.seed(42)``, one uniform draw per parameter), the multistart list (same generator, draw for
draw), the model ``[flat(p) ; lambda / P * p**2]`` against ``[target ; 0]``, log-space fitting for randmod, the scoring of every fit
with ``score_fit`` on the un-regularised target, the bootstrap noise stream (global NumPy state, as the reference), the return tuple
``(est_params, model_fits, error_vals, regularization_term)`` and the confidence-interval file.  The optimiser is not SciPy's TRF:
iterates differ, the minima of well-posed problems agree (tests/test_gpu_callers.py compares with a reference-run fixture).
The seaborn bar chart of the intervals (normest.py:545-548) is not drawn."""
from __future__ import annotations

import logging
import os
from itertools import combinations
from typing import Tuple

import numpy as np

from .. import config, models
from ..models.weights import early_emphasis, get_weight_options, get_protein_weights
from .identifiability import confidence_intervals
from . import multistart as _ms

logger = logging.getLogger(__name__)


def get_param_names(num_psites: int) -> list:
    """config/constants.py:164-169 -> config/helpers/__init__.py:5-37, for the currently configured model."""
    names = ['A', 'B', 'C', 'D'] + [f'S{i}' for i in range(1, num_psites + 1)]
    if config.ODE_MODEL == 'randmod':
        for i in range(1, num_psites + 1):
            names += [f"D{''.join(map(str, c))}" for c in combinations(range(1, num_psites + 1), i)]
        return names
    return names + [f'D{i + 1}' for i in range(num_psites)]


def _phys(p):
    return np.exp(p) if config.ODE_MODEL == 'randmod' else p


def _solver_kw() -> dict:
    """Options every batched solve of a fit shares with the one-call drop-in ``models.solve_ode`` (which reads the same config)."""
    kw = dict(config.SOLVER_OPTS)
    if config.NORMALIZE_MODEL_OUTPUT:
        kw["normalize"] = True
    return kw


def _score_at(popt, init_cond, num_psites, time_points, target) -> float:
    """``score_fit(theta, target, solve_ode(theta)[1])`` -- how the reference ranks fits (normest.py:93-101, 293-304, 461-465)."""
    return float(_ms._scores(config.ODE_MODEL, np.asarray(popt, float)[None], init_cond, num_psites, time_points, np.asarray(target, float), _solver_kw())[0])


def _scan(gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data, lambdas):
    """All (lambda, weighting) fits of the scan in one lockstep batch -> (scores [L, W], weight keys)."""
    lambdas = np.atleast_1d(np.asarray(lambdas, dtype=float))
    p0 = np.asarray(p0, dtype=float)
    opts = get_weight_options(target, time_points, num_psites, use_regularization=True, reg_len=len(p0),
                              early_weights=early_emphasis(pr_data, p_data, time_points, num_psites), ms_gauss_weights=get_protein_weights(gene))
    _, _, scores = _ms.find_best_lambda_batch(config.ODE_MODEL, target, p0, time_points, tuple(np.asarray(b, float) for b in free_bounds), init_cond,
                                              num_psites, opts, lambdas=lambdas, **_solver_kw())
    return scores, list(opts)


def worker_find_lambda(lam: float, gene: str, target, p0, time_points, free_bounds, init_cond, num_psites: int, p_data, pr_data) -> Tuple[float, float, str]:
    """(lam, best score over the weightings, its key) for ONE lambda: the reference's pool worker.  ``find_best_lambda`` does not call
    this per lambda -- it puts all lambdas into one batch."""
    scores, keys = _scan(gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data, [lam])
    j = int(np.argmin(scores[0]))
    if not np.isfinite(scores[0, j]):
        logger.warning(f"[{gene}] All fits failed for lambda = {lam:.2f}")
        return lam, float("inf"), None
    return lam, float(scores[0, j]), keys[j]


def find_best_lambda(gene: str, target, p0, time_points, free_bounds, init_cond, num_psites: int, p_data, pr_data, lambdas=np.logspace(-2, 0, 10),
                     max_workers: int = 4, per_lambda_timeout: float = 1800.0) -> Tuple[float, str]:
    """(best lambda, its weighting).  ``max_workers`` / ``per_lambda_timeout`` are accepted and unused: there is no pool."""
    scores, keys = _scan(gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data, lambdas)
    best_w = np.argmin(scores, axis=1)
    per_lam = scores[np.arange(scores.shape[0]), best_w]
    if not np.isfinite(per_lam).any():
        return None, None
    i = int(np.argmin(per_lam))
    return float(np.asarray(lambdas, float)[i]), keys[int(best_w[i])]


def _ridge_of(model_func, time_points, base_p0, n_data: int, n_fit: int) -> float:
    """lambda of a ``model_func(tpts, *params) -> [flat ; lambda / P * params**2]`` callable: read from its ``lambda_reg`` attribute when
    the caller set one, else recovered from ONE evaluation at the base point (the ridge rows are lambda / P * p**2 by construction)."""
    P = len(base_p0)
    if n_fit == n_data:
        return 0.0
    if n_fit != n_data + P:
        raise ValueError(f"target_fit holds {n_fit} entries; expected {n_data} (plain) or {n_data + P} (with the ridge rows)")
    lam = getattr(model_func, "lambda_reg", None)
    if lam is not None:
        return float(lam)
    if model_func is None:
        raise ValueError("a regularised target needs model_func (or model_func.lambda_reg) to define lambda")
    p = np.asarray(base_p0, dtype=float)
    rows = np.asarray(model_func(time_points, *p), dtype=float)[n_data:]
    k = int(np.argmax(np.abs(p)))
    if p[k] == 0.0:
        raise ValueError("cannot recover lambda from model_func at an all-zero base point: set model_func.lambda_reg")
    return float(rows[k] * P / (p[k] * p[k]))


def _curve_fit_multistart(gene: str, model_func, time_points, target_fit, base_p0, free_bounds, sigma, init_cond, num_psites: int, target,
                          n_starts: int = 24, jitter_frac: float = 0.10, maxfev: int = 20000, seed: int = 42):
    """(popt_best, pcov_best, best_score) over ``n_starts`` start points, all fitted in one lockstep batch.  ``model_func`` cannot be
    batched as a black box: the fit is of the configured model's ``[flat ; ridge]`` (what every caller in the reference passes), with the
    ridge weight taken from ``model_func`` (see ``_ridge_of``).  Raises ValueError for non-finite bounds and RuntimeError when every
    start fails, as the reference."""
    lb, ub = (np.asarray(b, dtype=float) for b in free_bounds)
    target = np.asarray(target, dtype=float)
    lam = _ridge_of(model_func, time_points, base_p0, target.size, np.size(target_fit))
    res = _ms.curve_fit_multistart_batch(config.ODE_MODEL, init_cond, num_psites, time_points, target, base_p0, (lb, ub), sigma=sigma, lam=lam, gene=gene,
                                         n_starts=n_starts, jitter_frac=jitter_frac, seed=seed, absolute_sigma=not config.USE_CUSTOM_WEIGHTS, **_solver_kw())
    if not np.isfinite(res.score):
        raise RuntimeError(f"[{gene}] multistart curve_fit: all starts failed (n={res.p_all.shape[0]}).")
    logger.info(f"[{gene}]\t\tMultistart curve_fit: starts={res.p_all.shape[0]} best_score={res.score:6.2f}")
    return res.popt, res.pcov, res.score


def normest(gene, pr_data, p_data, r_data, init_cond, num_psites, time_points, bounds, bootstraps, use_regularization=None):
    """Estimate the parameters of one gene.  Returns ``(est_params, model_fits, error_vals, regularization_term)``: one-element lists
    holding the parameter vector (physical space), ``(sol [T, S], flat)`` at it and the mean squared error against the data, plus
    lambda / P * sum(theta**2)."""
    use_regularization = config.USE_REGULARIZATION if use_regularization is None else use_regularization
    model = config.ODE_MODEL
    pr_data, p_data, r_data = (np.asarray(a, dtype=float) for a in (pr_data, p_data, r_data))
    time_points = np.asarray(time_points, dtype=float)
    lb, ub = _ms.build_free_bounds(model, bounds, num_psites)
    free_bounds = (list(lb), list(ub))
    np.random.seed(42)                                                        # normest.py:386: the GLOBAL state, as the reference
    p0 = np.array([np.random.uniform(low=l, high=u) for l, u in zip(lb, ub)])
    P = len(p0)
    target = np.concatenate([r_data.flatten(), pr_data.flatten(), p_data.flatten()])
    target_fit = np.concatenate([target, np.zeros(P)]) if use_regularization else target

    lambda_reg, lambda_weight = find_best_lambda(gene, target, p0, time_points, free_bounds, init_cond, num_psites, p_data, pr_data, max_workers=10)
    if lambda_reg is None:
        raise RuntimeError(f"[{gene}] every fit of the lambda scan failed")
    logger.info(f"[{gene}]      Using λ = {lambda_reg / P * np.sum(np.square(p0)): .4f}")
    lam_fit = lambda_reg if use_regularization else 0.0

    def model_func(tpts, *params):
        _, flat = models.model_module_for(model).solve_ode(_phys(np.asarray(params, float)), init_cond, num_psites, np.atleast_1d(tpts))
        return np.concatenate([flat, lambda_reg / P * np.square(params)]) if use_regularization else flat
    model_func.lambda_reg = lam_fit

    weight_options = get_weight_options(target, time_points, num_psites, use_regularization, P,
                                        early_emphasis(pr_data, p_data, time_points, num_psites), get_protein_weights(gene))
    sigma = weight_options[lambda_weight]
    try:
        popt, pcov, _ = _curve_fit_multistart(gene=gene, model_func=model_func, time_points=time_points, target_fit=target_fit, base_p0=p0,
                                              free_bounds=free_bounds, sigma=sigma, init_cond=init_cond, num_psites=num_psites, target=target,
                                              n_starts=48, jitter_frac=0.10, maxfev=20000, seed=42)
    except Exception as e:                                                    # normest.py:453-456
        logger.warning(f"[{gene}] Final multistart fit failed for {lambda_weight}: {e}")
        popt, pcov = p0, None
    popt_best, pcov_best = popt, pcov
    logger.info(f"[{gene}]      Fit Score: {_score_at(popt, init_cond, num_psites, time_points, target):.2f}")
    ci_results = confidence_intervals(gene, _phys(popt_best), pcov_best, target_fit, model_func(time_points, *popt_best), alpha_val=config.ALPHA_CI)

    if bootstraps > 0:
        # normest.py:488-531: refits of target_fit * (1 + N(0, 0.05)) from popt_best, noise from the global NumPy state; one batch
        popt_best, pcov_best, _ = _ms.bootstrap_fit_batch(model, target_fit, popt_best, time_points, (lb, ub), init_cond, num_psites, sigma=sigma, lam=lam_fit,
                                                          bootstraps=int(bootstraps), noise=0.05, rng=np.random, absolute_sigma=not config.USE_CUSTOM_WEIGHTS,
                                                          **_solver_kw())
        ci_results = confidence_intervals(gene, _phys(popt_best), pcov_best, target_fit, model_func(time_points, *popt_best), alpha_val=config.ALPHA_CI)

    if config.OUT_DIR is not None and ci_results is not None:
        import pandas as pd
        os.makedirs(config.OUT_DIR, exist_ok=True)
        pd.DataFrame({'Parameter': get_param_names(num_psites), 'Estimate': ci_results['beta_hat'], 'Std_Error': ci_results['se_lin'],
                      'p_value': ci_results['pval'], 'Lower_95CI': ci_results['lwr_ci'], 'Upper_95CI': ci_results['upr_ci']}
                     ).to_csv(f"{config.OUT_DIR}/{gene}_confidence_intervals.csv", index=False)

    param_final = _phys(popt_best)
    sol, p_fit = models.model_module_for(model).solve_ode(param_final, init_cond, num_psites, time_points)
    error = np.sum(np.abs(p_fit.flatten() - target) ** 2) / target.size
    regularization_term = lambda_reg / len(param_final) * np.sum(np.square(param_final))
    return [param_final], [(sol, p_fit)], [error], regularization_term
