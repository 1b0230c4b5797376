"""Network Morris screening (reference: global_model/sensitivity.py), batched and shardable.

  compute_bounds / _reconstruct_params / _compute_scalar_metric     sensitivity.py:41-140  (host helpers, same semantics)
  run_sensitivity_batch                                             the numerical core of run_sensitivity_analysis (:171-277): sample ->
        candidates -> ONE simulate launch (rtol 1e-5 / atol 1e-7 as simulate_and_measure) -> fold-change observables -> scalar metric per
        candidate -> (multi-GPU: one all-gather of Y) -> elementary-effects analysis.  The reference pickles the whole System per task."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ..sensitivity import morris
from ..distributed import interleaved_rows, all_gather_interleaved_with_status, shared_seed
from . import config
from .engine import NetworkEngine

_ORDER = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i", "tf_scale")


def compute_bounds(params_dict, perturbation=config.SENSITIVITY_PERTURBATION):
    bounds, names = [], []
    for key, value in params_dict.items():
        if isinstance(value, np.ndarray):
            for i, v in enumerate(value):
                lb, ub = v * (1 - perturbation), v * (1 + perturbation)
                if abs(v) < 1e-6:
                    lb, ub = 0.0, 0.01
                bounds.append([max(0.0, lb), ub]); names.append(f"{key}_{i}")
        else:
            v = float(value)
            lb, ub = v * (1 - perturbation), v * (1 + perturbation)
            if abs(v) < 1e-6:
                lb, ub = 0.0, 0.01
            bounds.append([max(0.0, lb), ub]); names.append(key)
    return {"num_vars": len(names), "names": names, "bounds": bounds}


def _reconstruct_params(param_vector, names_map, original_shapes):
    p_out, curr = {}, 0
    for key, shape in original_shapes.items():
        if shape == ():
            p_out[key] = param_vector[curr]; curr += 1
        else:
            size = int(np.prod(shape))
            p_out[key] = np.array(param_vector[curr: curr + size]); curr += size
    return p_out


def _compute_scalar_metric(df_prot, df_rna, df_phos, metric="total_signal"):
    parts = [d["pred_fc"].values for d in (df_prot, df_rna, df_phos) if d is not None]
    combined = np.concatenate(parts) if parts else np.array([])
    if len(combined) == 0:
        return 0.0
    if metric == "mean":
        return np.mean(combined)
    if metric == "variance":
        return np.var(combined)
    if metric == "l2_norm":
        return np.linalg.norm(combined)
    return np.sum(combined)


def scalar_metric_batch(pred: torch.Tensor, metric: str = "total_signal") -> torch.Tensor:
    """_compute_scalar_metric over the rows of pred [B, n_obs] (GPU).  The row sums run as a fixed pairwise tree over the columns
    (elementwise adds of column halves): a row's value does not depend on how many rows share the launch -- a library reduction picks its
    split by the whole shape, and a rank that owns half the rows would then round differently from the one-process run."""
    def rowsum(a: torch.Tensor) -> torch.Tensor:
        n = a.shape[1]
        if n == 0:
            return a.new_zeros(a.shape[0])
        m = 1 << (n - 1).bit_length()
        if m != n:
            a = torch.cat([a, a.new_zeros(a.shape[0], m - n)], dim=1)
        while m > 1:
            m >>= 1
            a = a[:, :m] + a[:, m:]
        return a[:, 0]
    n = max(pred.shape[1], 1)
    if metric == "mean":
        return rowsum(pred) / n
    if metric == "variance":
        mu = rowsum(pred) / n
        return rowsum((pred - mu[:, None]) ** 2) / n
    if metric == "l2_norm":
        return torch.sqrt(rowsum(pred * pred))
    return rowsum(pred)


def run_sensitivity_batch(eng: NetworkEngine, fitted_params: Dict, times_p, times_r, times_ph, perturbation: float = config.SENSITIVITY_PERTURBATION,
                          trajectories: int = config.SENSITIVITY_TRAJECTORIES, num_levels: int = config.SENSITIVITY_LEVELS,
                          metric: str = config.SENSITIVITY_METRIC, seed: Optional[int] = None,
                          param_values: Optional[np.ndarray] = None, y0=None, conf_level: float = 0.95, rtol: Optional[float] = None, atol: Optional[float] = None,
                          vary=None, return_pred: bool = False):
    """Returns dict(Si, problem, param_values, Y, status).  ``fitted_params`` maps the eight parameter groups (System.update order)
    to arrays / a scalar; every entry is varied, as in the reference (sensitivity.py:196-215) -- unless ``vary`` names a subset: indices
    into the flat parameter vector (or parameter names ``"A_i_3"``, ``"tf_scale"``); the design then has len(vary) dimensions
    (trajectories x (len(vary) + 1) simulations: BASELINE config 4 is 128 x (200 + 1)) and every other entry stays at its fitted value.
    ``param_values`` (a ready sample matrix) has one column per varied entry.  ``return_pred``: also return the fold-change observables
    ``pred`` [rows of this rank, n_obs] (GPU tensor) and their ``layout``."""
    params = {k: (np.asarray(fitted_params[k], float) if k != "tf_scale" else float(fitted_params[k])) for k in _ORDER}
    from .simulate import measure_tolerances                  # the settings of simulate_and_measure, which the reference's workers call
    tol = measure_tolerances(eng)
    rtol = tol["rtol"] if rtol is None else rtol
    atol = tol["atol"] if atol is None else atol
    problem = compute_bounds(params, perturbation)
    center = None
    if vary is not None:
        names = problem["names"]
        pos = {nm: i for i, nm in enumerate(names)}
        vidx = np.array([pos[v] if isinstance(v, str) else int(v) for v in vary], dtype=np.int64)
        if vidx.size == 0 or np.unique(vidx).size != vidx.size or vidx.min() < 0 or vidx.max() >= len(names):
            raise ValueError("vary must hold distinct indices (or names) of the flat parameter vector")
        center = eng.pack_params(*(params[k] for k in _ORDER))
        problem = {"num_vars": int(vidx.size), "names": [names[i] for i in vidx], "bounds": [problem["bounds"][i] for i in vidx]}
    design = None
    seed = shared_seed(seed)              # N > 1 ranks: every rank must build the SAME design (seed=None would give each its own)
    if param_values is None:
        # the design is built in HBM from the draws (pk_morris_build_batch): no N (D + 1) x D matrix crosses PCIe on the way in
        Xd, design = morris.sample_device(problem, N=trajectories, num_levels=num_levels, seed=seed, device=eng.ctx.device)
    else:
        Xd = torch.as_tensor(np.ascontiguousarray(param_values, dtype=np.float64), device=torch.device("cuda", eng.ctx.device))
    if Xd.shape[1] != problem["num_vars"]:
        raise ValueError(f"the sample matrix has {Xd.shape[1]} columns, the problem {problem['num_vars']} variables")
    Xv = Xd                                                             # the design in its own (varied) coordinates
    if center is not None:                                              # scatter the varied columns into copies of the fitted vector (in HBM)
        Xd = torch.as_tensor(center, device=Xv.device).repeat(Xv.shape[0], 1)
        Xd[:, torch.as_tensor(vidx, device=Xv.device)] = Xv
    if Xd.shape[1] != eng.n_var:                                        # flat vector order == candidate row order (both System.update order)
        raise ValueError(f"parameter vector has {Xd.shape[1]} entries, the network expects {eng.n_var}")
    total = Xd.shape[0]
    times = np.unique(np.concatenate([times_p, times_r, times_ph]).astype(np.float64))
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    # rows are dealt round-robin over the ranks: a Morris design walks through parameter space trajectory by trajectory, so contiguous
    # blocks would hand each rank its own region (and its own step counts); interleaved, every rank integrates the same mix
    rows = interleaved_rows(total, rank, world).to(Xd.device)
    lists, ld = eng.make_index_lists(times, times_p, times_r, times_ph)
    n_obs = ld["p_prot"].size + ld["p_rna"].size + ld["p_pho"].size
    pred = None
    mean_steps = None
    try:
        if rows.numel() > 0:
            Y, status, nst = eng.simulate_batch(Xd if world == 1 else Xd[rows], times, y0=y0, rtol=rtol, atol=atol, max_steps=5000 * times.size)
            mean_steps = nst.double().mean(dim=0)
            pred = eng.observables_batch(lists, Y, n_obs, eps=1e-12)
            yloc = scalar_metric_batch(pred, metric)
            yloc = torch.where(status != 0, torch.zeros_like(yloc), yloc)          # failed simulations contribute Y = 0
        else:
            dev = torch.device("cuda", eng.ctx.device)
            yloc = torch.empty(0, dtype=torch.float64, device=dev); status = torch.empty(0, dtype=torch.int32, device=dev)
        Yall, stat = all_gather_interleaved_with_status(yloc, status, total) if world > 1 else (yloc, status)      # the ONE collective
        Yall = torch.nan_to_num(Yall, nan=0.0, posinf=0.0, neginf=0.0)
        if design is not None:
            ee = morris.elementary_effects_device(design, Yall).cpu().numpy()      # EE [N, D] on the GPU: 8 N D bytes come back
        torch.cuda.current_stream().synchronize()
    finally:
        eng.free_loss(lists)
    Yh = Yall.cpu().numpy()
    X = Xv.cpu().numpy()
    if design is not None:
        Si = morris.analyze_effects(ee, problem.get("names"), conf_level=conf_level, seed=seed)
    else:
        Si = morris.analyze(problem, X, Yh, num_levels=num_levels, conf_level=conf_level, seed=seed)
    out = {"Si": Si, "problem": problem, "param_values": X, "Y": Yh, "status": stat.cpu().numpy(),
           "mean_steps": (None if mean_steps is None else [float(v) for v in mean_steps.cpu()])}      # accepted, rejected: this rank's rows
    if return_pred:
        out.update(pred=pred, layout=ld, rows=rows.cpu().numpy(), times=times)
    return out


def _worker_simulation(task_args):
    """``(idx, param_vector, names_map, original_shapes, sys, idx_sys, times_p, times_r, times_ph, metric) -> (idx, y, dfp, dfr, dfph)``:
    the reference's pool worker (sensitivity.py:143-168), one simulation.  ``run_sensitivity_analysis`` does not use it: all samples are
    one launch and no ``System`` is pickled."""
    from .simulate import simulate_and_measure
    (i, param_vector, names_map, original_shapes, sys, idx_sys, times_p, times_r, times_ph, metric) = task_args
    sys.update(**_reconstruct_params(param_vector, names_map, original_shapes))
    dfp, dfr, dfph = simulate_and_measure(sys, idx_sys, times_p, times_r, times_ph)
    return i, _compute_scalar_metric(dfp, dfr, dfph, metric), dfp, dfr, dfph


def run_sensitivity_analysis(sys, idx, fitted_params, output_dir, metric="total_signal", seed: Optional[int] = None,
                             param_values: Optional[np.ndarray] = None):
    """Morris screening of the network around ``fitted_params`` with the reference's argument list (sensitivity.py:171-297) -> the
    DataFrame ``Parameter, mu_star, sigma, mu_star_conf`` sorted by influence, also written to ``<output_dir>/sensitivity_indices.csv``.

    N = ``config.SENSITIVITY_TRAJECTORIES`` trajectories on ``config.SENSITIVITY_LEVELS`` levels within +-``config.SENSITIVITY_PERTURBATION``
    of every entry; the N (D + 1) simulations are ONE launch on the engine of ``sys`` (the reference pickles the whole System into every
    task of a process pool); under an initialised ``torch.distributed`` group the rows are sharded over the ranks with one all-gather of Y.
    ``<output_dir>/sensitivity_trajectories.csv`` lists the ``config.SENSITIVITY_TOP_CURVES`` samples of largest Y (id, y_val and the
    sampled parameter values; the reference stores whole DataFrames in that table's cells).  The two plots (:290-296) are not drawn.
    ``seed`` defaults to ``config.SEED``; ``param_values`` (keyword, not in the reference) takes a ready sample matrix such as SALib's."""
    import os
    import pandas as pd
    from .simulate import engine_for
    eng = engine_for(sys)
    params = {}
    for k in _ORDER:
        v = fitted_params[k]
        params[k] = float(v) if k == "tf_scale" else np.asarray(v, dtype=float)
    res = run_sensitivity_batch(eng, params, config.TIME_POINTS_PROTEIN, config.TIME_POINTS_RNA, config.TIME_POINTS_PHOSPHO, metric=metric,
                                perturbation=config.SENSITIVITY_PERTURBATION, trajectories=config.SENSITIVITY_TRAJECTORIES,
                                num_levels=config.SENSITIVITY_LEVELS, seed=config.SEED if seed is None else seed, param_values=param_values, y0=np.asarray(sys.y0(), dtype=np.float64))
    Si = res["Si"]
    df_sens = pd.DataFrame({"Parameter": res["problem"]["names"], "mu_star": Si["mu_star"], "sigma": Si["sigma"], "mu_star_conf": Si["mu_star_conf"]})
    df_sens = df_sens.sort_values("mu_star", ascending=False)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        df_sens.to_csv(os.path.join(output_dir, "sensitivity_indices.csv"), index=False)
        top = np.argsort(-res["Y"], kind="stable")[:config.SENSITIVITY_TOP_CURVES]
        traj = pd.DataFrame(res["param_values"][top], columns=res["problem"]["names"])
        traj.insert(0, "y_val", res["Y"][top]); traj.insert(0, "id", top)
        traj.to_csv(os.path.join(output_dir, "sensitivity_trajectories.csv"), index=False)
    return df_sens
