"""Discrete Frechet distance between time courses, batched over candidates and series on the GPU (``pk_frechet_batch``).

  frechet_distance(true_coords, pred_coords)   drop-in of frechet/distance.py:9-56 for one pair of [n, 2] / [m, 2] curves
  frechet_batch(obs_curves, pred_t, pred_idx, pred)   B candidates x n_series series in one launch: the inner double loop of the Pareto
                                                      pick (global_model/runner.py:780-841) without DataFrames
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .. import batch

MAX_POINTS = 32


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def frechet_batch(obs_curves: Sequence[np.ndarray], pred_t: Sequence[np.ndarray], pred_idx: Sequence[np.ndarray], pred,
                  device: Optional[int] = None) -> torch.Tensor:
    """out [B, n_series]: series s compares ``obs_curves[s]`` ([n_s, 2] rows (time, value), sorted by time) with the predicted points
    ``(pred_t[s][k], pred[b, pred_idx[s][k]])``.  ``pred`` is [B, n_obs] (numpy or GPU tensor)."""
    ctx = batch.get_context(device)
    dev = torch.device("cuda", ctx.device)
    n_series = len(obs_curves)
    if not (len(pred_t) == len(pred_idx) == n_series):
        raise ValueError("obs_curves, pred_t and pred_idx must list the same series")
    pr = (pred if isinstance(pred, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(pred, dtype=np.float64))).to(dev, torch.float64).contiguous()
    if pr.dim() == 1:
        pr = pr.unsqueeze(0)
    B, n_obs = pr.shape
    out = torch.zeros((B, n_series), dtype=torch.float64, device=dev)
    if B == 0 or n_series == 0:
        return out
    optr = np.zeros(n_series + 1, np.int32); pptr = np.zeros(n_series + 1, np.int32)
    for s in range(n_series):
        o = np.asarray(obs_curves[s], float).reshape(-1, 2)
        if len(pred_t[s]) != len(pred_idx[s]):
            raise ValueError("pred_t[s] and pred_idx[s] differ in length")
        optr[s + 1] = optr[s] + o.shape[0]; pptr[s + 1] = pptr[s] + len(pred_t[s])
    lens = np.concatenate([np.diff(optr), np.diff(pptr)])
    if lens.max(initial=0) > MAX_POINTS:
        raise ValueError(f"curves of at most {MAX_POINTS} points")
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate([np.asarray(x, dt).reshape(-1) for x in xs]) if len(xs) else np.zeros(0, dt))
    oc = np.concatenate([np.asarray(c, float).reshape(-1, 2) for c in obs_curves]) if n_series else np.zeros((0, 2))
    pidx = cat(pred_idx, np.int32)
    if pidx.size and (pidx.min() < 0 or pidx.max() >= n_obs):
        raise ValueError("pred_idx out of range")
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    d = [t(optr, torch.int32), t(oc[:, 0], torch.float64), t(oc[:, 1], torch.float64), t(pptr, torch.int32), t(cat(pred_t, float), torch.float64),
         t(pidx, torch.int32)]
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.pk_frechet_batch(ctx.handle, B, n_series, _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), _ptr(d[4]), _ptr(d[5]), _ptr(pr), n_obs,
                                       int(lens.max(initial=0)), _ptr(out)))
    out._keepalive = (d, pr)  # type: ignore[attr-defined]
    return out


def frechet_distance(true_coords, pred_coords) -> float:
    """One pair of curves (reference signature: two C-contiguous [n, 2] float arrays -> float)."""
    p = np.asarray(pred_coords, float).reshape(-1, 2)
    return float(frechet_batch([np.asarray(true_coords, float)], [p[:, 0]], [np.arange(p.shape[0])], p[:, 1][None, :])[0, 0])
