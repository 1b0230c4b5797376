"""Drop-in names of the reference's ``sensitivity/analysis.py`` plus the batched driver that replaces its process pool.

  compute_bound                        sensitivity/analysis.py:20-35
  define_sensitivity_problem_ds/_rand  sensitivity/analysis.py:38-87
  _compute_Y                           sensitivity/analysis.py:90-176   (single-array host version; the batched path computes it
                                                                          on the GPU, fused into the solve kernel)
  _perturb_solve                       sensitivity/analysis.py:178-195  (the pool worker: one parameter set -> (i, sol, flat, Y))
  _sensitivity_analysis                sensitivity/analysis.py:197-331  (reference argument list; returns (Si, best_trajectories))
  sensitivity_analysis_batch           the numerical core of _sensitivity_analysis: sample -> N*(D+1) solves + Y in
                                       ONE launch -> analyze -> RMSE ranking against data -> best K curves (no plotting)
"""
from __future__ import annotations

import logging
import math
from typing import Dict, Optional, Sequence

import numpy as np

from .. import config
from . import morris

logger = logging.getLogger(__name__)


def compute_bound(value, perturbation=None):
    perturbation = config.PERTURBATIONS_VALUE if perturbation is None else perturbation
    if abs(value) < 1e-6:
        return [0.0, 0.1]
    lb = value * (1 - perturbation)
    ub = value * (1 + perturbation)
    return [max(0.0, lb), ub]


def _rand_param_names(num_psites: int):
    """config/helpers/__init__.py:5-22 (get_param_names_rand): D12 = joint dephosphorylation of sites 1 and 2; subsets by size, then
    lexicographic -- the order of the D block of randmod's parameter vector."""
    from itertools import combinations
    names = ['A', 'B', 'C', 'D'] + [f'S{i}' for i in range(1, num_psites + 1)]
    for i in range(1, num_psites + 1):
        for combo in combinations(range(1, num_psites + 1), i):
            names.append(f"D{''.join(map(str, combo))}")
    return names


def define_sensitivity_problem_rand(num_psites, values):
    num_vars = 4 + num_psites + (1 << num_psites) - 1
    assert len(values) == num_vars, "Length mismatch with values"
    return {'num_vars': num_vars, 'names': _rand_param_names(num_psites), 'bounds': [compute_bound(v) for v in values]}


def define_sensitivity_problem_ds(num_psites, values):
    num_vars = 4 + 2 * num_psites
    names = ['A', 'B', 'C', 'D'] + [f'S{i + 1}' for i in range(num_psites)] + [f'D{i + 1}' for i in range(num_psites)]
    assert len(values) == num_vars, "Length mismatch with values"
    return {'num_vars': num_vars, 'names': names, 'bounds': [compute_bound(v) for v in values]}


def _compute_Y(solution: np.ndarray, num_psites: int, metric: Optional[str] = None) -> float:
    """Scalar Morris output of ONE solution array (reference signature; Y_METRIC from config unless given)."""
    metric = config.Y_METRIC if metric is None else metric
    sub = np.asarray(solution, dtype=float)[:, :2 + num_psites]
    n_t = sub.shape[0]
    length = 2 * n_t + n_t * num_psites
    total = float(sub.sum())
    if metric == 'total_signal':
        return total
    if metric == 'mean_activity':
        return total / length
    if metric == 'variance':
        return float(((sub - total / length) ** 2).sum() / length)
    if metric == 'dynamics':
        return float((np.diff(sub, axis=0) ** 2).sum())
    if metric == 'l2_norm':
        return math.sqrt(float((sub ** 2).sum()))
    raise ValueError("Unknown Y_METRIC")


def sensitivity_analysis_batch(popt: Sequence[float], time_points, num_psites: int, init_cond, model: Optional[str] = None,
                               N: Optional[int] = None, num_levels: Optional[int] = None, param_values: Optional[np.ndarray] = None,
                               pr_data=None, p_data=None, rna_data=None, y_metric: Optional[str] = None, seed: Optional[int] = None,
                               conf_level: float = 0.99, keep_solutions: bool = True, **solver_kw) -> Dict:
    """Morris screening around a fitted parameter vector, all solves in one batch on the GPU.

    ``param_values`` may be supplied (e.g. from ``SALib.sample.morris.sample`` with ``local_optimization=True`` as the
    reference does); otherwise the built-in sampler is used.  Returns a dict with ``Si`` (mu, mu_star, sigma,
    mu_star_conf, names), ``param_values``, ``Y``, ``status`` and -- if data are given -- ``rmse`` and ``best_idx``
    (the K = ceil(10 N / num_levels) closest simulations, sensitivity/analysis.py:289-294)."""
    from .. import batch
    model = config.ODE_MODEL if model is None else model
    N = config.NUM_TRAJECTORIES if N is None else int(N)
    num_levels = config.PARAMETER_SPACE if num_levels is None else int(num_levels)
    y_metric = config.Y_METRIC if y_metric is None else y_metric
    popt = np.asarray(popt, dtype=float)
    problem = (define_sensitivity_problem_rand if model == 'randmod' else define_sensitivity_problem_ds)(num_psites, list(popt))
    if param_values is None:
        # design built in HBM from the draws, outputs reduced to elementary effects on the GPU: only draws (KB) and EE [N, D] cross PCIe
        Xd, h = morris.sample_device(problem, N=N, num_levels=num_levels, seed=seed)
        res = batch.solve_ode_batch(model, Xd, init_cond, num_psites, time_points, want_sol=keep_solutions, want_flat=False,
                                    metric=y_metric, **solver_kw)
        import torch
        Yd = torch.nan_to_num(res.metric, nan=0.0, posinf=0.0, neginf=0.0)        # sensitivity/analysis.py:261
        ee = morris.elementary_effects_device(h, Yd)
        b = np.asarray(problem["bounds"], float)
        U = (Xd - h["lb"]) / torch.where(h["ub"] > h["lb"], h["ub"] - h["lb"], torch.ones_like(h["lb"]))
        sy = float(Yd.std(unbiased=False))
        scale = (U.std(dim=0, unbiased=False) / sy) if sy > 0 else torch.zeros(U.shape[1], dtype=U.dtype, device=U.device)
        Si = morris.analyze_effects((ee * scale).cpu().numpy(), problem.get("names"), conf_level=conf_level, seed=seed)
        param_values = Xd.cpu().numpy()
        Y = Yd.cpu().numpy()
    else:
        param_values = np.ascontiguousarray(param_values, dtype=float)
        res = batch.solve_ode_batch(model, param_values, init_cond, num_psites, time_points, want_sol=keep_solutions, want_flat=False,
                                    metric=y_metric, **solver_kw)
        Y = res.metric.cpu().numpy()
        Y = np.nan_to_num(Y, nan=0.0, posinf=0.0, neginf=0.0)             # sensitivity/analysis.py:261
        Si = morris.analyze(problem, param_values, Y, num_levels=num_levels, conf_level=conf_level, scaled=True, seed=seed)
    out = {"problem": problem, "Si": Si, "param_values": param_values, "Y": Y, "status": res.status.cpu().numpy()}
    if keep_solutions:
        sol = res.sol.cpu().numpy()
        out["solutions"] = sol
        if pr_data is not None and p_data is not None and rna_data is not None:
            n_rna = len(config.TIME_POINTS_RNA)
            protein_ref = np.asarray(pr_data, float).reshape(-1)
            psite_ref = np.asarray(p_data, float)
            rna_ref = np.asarray(rna_data, float).reshape(-1)
            rna_diff = np.abs(sol[:, -n_rna:, 0] - rna_ref[None, :]) / rna_ref.size
            psite_diff = np.abs(sol[:, :, 2:2 + num_psites] - psite_ref.T[None, :, :]) / psite_ref.size
            protein_diff = np.abs(sol[:, :, 1] - protein_ref[None, :]) / protein_ref.size
            rmse = np.sqrt(((rna_diff ** 2).mean(axis=1) + (psite_diff ** 2).mean(axis=(1, 2)) + (protein_diff ** 2).mean(axis=1)) / 2.0)
            K = int(np.ceil(N * 10 / num_levels))
            out["rmse"] = rmse
            out["best_idx"] = np.argsort(rmse)[:K]
    return out


def _perturb_solve(i_X_tuple):
    """``(i, X, init_cond, num_psites, time_points) -> (i, solution [T, S], flat, Y)`` for one parameter set: the reference's pool worker.
    ``_sensitivity_analysis`` does not call it per sample -- all samples are one launch."""
    from ..models import model_module_for
    i, X, init_cond, num_psites, time_points = i_X_tuple
    solution, flat = model_module_for(config.ODE_MODEL).solve_ode(tuple(X), init_cond, num_psites, time_points)
    return i, solution, flat, _compute_Y(solution, num_psites)


def _sensitivity_analysis(pr_data, p_data, rna_data, popt, time_points, num_psites, psite_labels, state_labels, init_cond, gene,
                          param_values: Optional[np.ndarray] = None, seed: Optional[int] = None):
    """Morris screening around ``popt`` with the reference's argument list -> ``(Si, best_trajectories)``.

    ``Si``: dict with ``names``, ``mu``, ``mu_star``, ``sigma``, ``mu_star_conf`` (the keys read from SALib's result; sigma-scaled
    effects, 99 % bootstrap interval as sensitivity/analysis.py:264).  ``best_trajectories``: the K = ceil(10 N / levels) simulations
    closest to the data, each ``{"params", "solution", "rmse"}``, ascending in RMSE (:286-297).  N = ``config.NUM_TRAJECTORIES``,
    levels = ``config.PARAMETER_SPACE``; all N (D + 1) solves and their Y are ONE launch (the reference: a process pool of
    ``os.cpu_count()`` workers).  ``psite_labels`` / ``state_labels`` only label the reference's plots and are unused; no plot is drawn.
    ``param_values`` (keyword, not in the reference): a ready sample matrix, e.g. SALib's, instead of the built-in design."""
    out = sensitivity_analysis_batch(popt, time_points, num_psites, init_cond, param_values=param_values, pr_data=pr_data, p_data=p_data,
                                     rna_data=rna_data, seed=seed, normalize=config.NORMALIZE_MODEL_OUTPUT, **config.SOLVER_OPTS)
    logger.info(f"[{gene}]      Sensitivity Analysis completed")
    best = [{"params": out["param_values"][i], "solution": out["solutions"][i], "rmse": out["rmse"][i]} for i in out["best_idx"]]
    return out["Si"], best
