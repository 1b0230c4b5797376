"""Pareto pick by Frechet distance (global_model/runner.py:780-841) for a whole population at once.

The reference loops over the Pareto set: ``sys.update`` -> ``simulate_and_measure`` (three DataFrames) -> per protein / site
``frechet_distance(obs[['time','fc']], pred[['time','pred_fc']])`` -> weighted sum -> argmin.  Here: one simulate launch for all
candidates, one observables launch, one Frechet launch; observed data come as arrays."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from ..frechet import frechet_batch
from .engine import NetworkEngine


def frechet_pick_batch(eng: NetworkEngine, X, times_p, times_r, times_ph, obs_prot: Dict[int, np.ndarray], obs_rna: Dict[int, np.ndarray],
                       obs_pho: Dict[Tuple[int, int], np.ndarray], lambdas=(1.0, 1.0, 1.0), raw: bool = False, rtol: float = 1e-5,
                       atol: float = 1e-7, y0=None):
    """X [B, n_var] candidates (physical, or raw with ``raw=True``).  ``obs_prot[i]`` / ``obs_rna[i]`` are [n, 2] arrays (time, fc) of
    protein i, ``obs_pho[(i, j)]`` of site j of protein i.  A series enters only if both curves have more than one point, as in the
    reference.  Returns dict(best, scores [B], per_series [B, n_series], series (labels), status [B])."""
    times = np.unique(np.concatenate([times_p, times_r, times_ph]).astype(np.float64))
    lists, ld = eng.make_index_lists(times, times_p, times_r, times_ph)
    n_p, n_r = ld["p_prot"].size, ld["p_rna"].size
    n_obs = n_p + n_r + ld["p_pho"].size
    sel = lambda tp: times[np.isin(times, np.asarray(tp, float))]
    tp_, tr_, tph_ = sel(times_p), sel(times_r), sel(times_ph)
    ns = eng._keep[2]
    site_flat = {}
    k = 0
    for i in range(eng.N):
        for j in range(int(ns[i])):
            site_flat[(i, j)] = k; k += 1
    curves, ptimes, pidx, labels, weight = [], [], [], [], []
    def add(obs, tt, first, lab, w):
        o = np.asarray(obs, float).reshape(-1, 2)
        o = o[np.argsort(o[:, 0], kind="stable")]
        if o.shape[0] > 1 and tt.size > 1:
            curves.append(o); ptimes.append(tt); pidx.append(first + np.arange(tt.size)); labels.append(lab); weight.append(w)
    for i, o in obs_prot.items():
        add(o, tp_, int(i) * tp_.size, ("prot", int(i)), lambdas[0])
    for i, o in obs_rna.items():
        add(o, tr_, n_p + int(i) * tr_.size, ("rna", int(i)), lambdas[1])
    for key, o in obs_pho.items():
        add(o, tph_, n_p + n_r + site_flat[(int(key[0]), int(key[1]))] * tph_.size, ("phospho", (int(key[0]), int(key[1]))), lambdas[2])
    try:
        Y, status, _ = eng.simulate_batch(X, times, y0=y0, raw=raw, rtol=rtol, atol=atol, max_steps=5000 * times.size)
        pred = eng.observables_batch(lists, Y, n_obs, eps=1e-12)
        per = frechet_batch(curves, ptimes, pidx, pred)
    finally:
        eng.free_loss(lists)
    per_h = per.cpu().numpy()
    scores = per_h @ np.asarray(weight, float) if labels else np.zeros(per_h.shape[0])
    st = status.cpu().numpy()
    scores = np.where(st != 0, np.inf, scores)                     # a failed simulation can never be picked
    return dict(best=int(np.argmin(scores)), scores=scores, per_series=per_h, series=labels, status=st)
