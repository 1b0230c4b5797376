"""Drop-in names of the reference's ``global_model/simulate.py`` on the MI355X engine.

``simulate_odeint(sys, t_eval, rtol, atol, mxstep)`` and ``simulate_and_measure(sys, idx, t_p, t_r, t_pho)`` accept the reference's
``System`` / ``Index`` objects (duck-typed: only array attributes and ``idx.proteins`` / ``idx.sites`` are read) and see ``sys.update``
mutations because parameters are packed from the object at call time.  The batched forms (``NetworkEngine.simulate_batch``,
``measure_batch``) are what the optimiser / Morris drivers use."""
from __future__ import annotations

import weakref

import numpy as np
import pandas as pd
import torch

from . import config
from .engine import NetworkEngine

# id(sys) -> (weakref to sys, {model: engine}).  The weak reference is what makes the id trustworthy: an entry is used only while its
# referent is alive AND identical to the caller's object, and a finalizer drops the entry when the System dies, so a recycled id can
# never hand out a stale topology.  Eviction only forgets the engines: a caller may still hold one (``eng = engine_for(build_system())``),
# and the HBM of an engine is freed when ITS last reference dies (NetworkEngine.__del__).
_engines: dict = {}


def _evict(key: int) -> None:
    _engines.pop(key, None)


def engine_for(sys, model=None) -> NetworkEngine:
    """One ``NetworkEngine`` per live reference ``System`` and kinetic model (the topology is static; parameters travel per call).
    Engines are evicted and closed when their ``System`` is garbage-collected."""
    model = config.MODEL if model is None else model
    try:
        weakref.ref(sys)
        weak = True
    except TypeError:
        weak = False
    if not weak:
        # duck-typed stand-ins that cannot be weakly referenced (e.g. types.SimpleNamespace): the engines live ON the object, so their
        # lifetime is the object's and no id is ever involved
        store = getattr(sys, "__dict__", None)
        if store is None:
            raise TypeError("engine_for needs a System object that is weak-referenceable or has a __dict__")
        cache = store.setdefault("_pk_engines", {})
        if model not in cache:
            cache[model] = NetworkEngine.from_system(sys, model)
        return cache[model]
    key = id(sys)
    ent = _engines.get(key)
    if ent is not None and ent[0]() is not sys:                # dead referent whose finalizer has not run yet, or a recycled id
        _evict(key)
        ent = None
    if ent is None:
        ent = _engines[key] = (weakref.ref(sys), {})
        weakref.finalize(sys, _evict, key)
    eng = ent[1].get(model)
    if eng is None:
        eng = ent[1][model] = NetworkEngine.from_system(sys, model)
    return eng


def candidate_of(sys, eng: NetworkEngine) -> np.ndarray:
    return eng.pack_params(sys.c_k, sys.A_i, sys.B_i, sys.C_i, sys.D_i, sys.Dp_i, sys.E_i, sys.tf_scale)


def simulate_odeint(sys, t_eval, rtol, atol, mxstep):
    """Y [T, S] (C-contiguous float64) for the system's current parameters: reference simulate.py:34-80."""
    eng = engine_for(sys)
    y0 = np.asarray(sys.y0(), dtype=np.float64)
    Y, status, _ = eng.simulate_batch(candidate_of(sys, eng)[None, :], np.asarray(t_eval, dtype=np.float64), y0=y0, rtol=rtol, atol=atol,
                                      max_steps=max(int(mxstep), 1) * max(len(np.atleast_1d(t_eval)), 1))
    return np.ascontiguousarray(Y[0].cpu().numpy())


def measure_batch(eng: NetworkEngine, Y: torch.Tensor, times, t_points_p, t_points_r, t_points_pho):
    """Fold-change observables of B trajectories: (pred [B, n_obs] GPU tensor, layout dict with the (protein, site, time) of each column)."""
    lists, ld = eng.make_index_lists(times, t_points_p, t_points_r, t_points_pho)
    n_obs = ld["p_prot"].size + ld["p_rna"].size + ld["p_pho"].size
    try:
        pred = eng.observables_batch(lists, Y, n_obs, eps=1e-12)
        torch.cuda.current_stream().synchronize()
    finally:
        eng.free_loss(lists)
    return pred, ld


def measure_tolerances(eng: NetworkEngine) -> dict:
    """Solver settings behind the reference's hard-wired ``simulate_odeint(sys, times, rtol=1e-5, atol=1e-7, mxstep=5000)`` of
    simulate_and_measure (simulate.py:110).  LSODA at those settings is 0.5-4 of ITS band widths from the truth (pins fixtures); the
    order-3 Rosenbrock-W method matches that at the same numbers, the order-4 additive method needs one decade less (and still takes
    half the steps)."""
    return dict(rtol=1e-6, atol=1e-8) if eng.ark_eligible() else dict(rtol=1e-5, atol=1e-7)


def simulate_and_measure(sys, idx, t_points_p, t_points_r, t_points_pho):
    """(df_prot, df_rna, df_phos) with columns protein, [psite,] time, pred_fc: reference simulate.py:83-202 (one solve at
    rtol 1e-5 / atol 1e-7 on the union grid, FC against t = 0 (protein, phospho) and t = 4 (RNA), rows filtered by exact time)."""
    eng = engine_for(sys)
    times = np.unique(np.concatenate([t_points_p, t_points_r, t_points_pho]).astype(np.float64))
    y0 = np.asarray(sys.y0(), dtype=np.float64)
    Y, _, _ = eng.simulate_batch(candidate_of(sys, eng)[None, :], times, y0=y0, max_steps=5000 * times.size, **measure_tolerances(eng))
    pred, ld = measure_batch(eng, Y, times, t_points_p, t_points_r, t_points_pho)
    v = pred[0].cpu().numpy()
    n_p, n_r = ld["p_prot"].size, ld["p_rna"].size
    prots = np.asarray(idx.proteins, dtype=object)
    df_p = pd.DataFrame({"protein": prots[ld["p_prot"]], "time": times[ld["t_prot"]], "pred_fc": v[:n_p]})
    df_r = pd.DataFrame({"protein": prots[ld["p_rna"]], "time": times[ld["t_rna"]], "pred_fc": v[n_p:n_p + n_r]})
    psite = np.asarray([idx.sites[i][j] for i, j in zip(ld["p_pho"], ld["s_pho"])], dtype=object)
    df_pho = pd.DataFrame({"protein": prots[ld["p_pho"]], "psite": psite, "time": times[ld["t_pho"]], "pred_fc": v[n_p + n_r:]})
    if df_pho.empty:
        df_pho = pd.DataFrame(columns=["protein", "psite", "time", "pred_fc"])
    return df_p, df_r, df_pho
