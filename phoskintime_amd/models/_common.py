"""Shared plumbing of the three drop-in model modules: one-replica calls through the C ABI's host-pointer entry
points (pk_*_host), numpy in / numpy out like the reference."""
from __future__ import annotations

import ctypes as C
import numpy as np

from .. import _capi, config
from ..batch import get_context, flat_len, n_params, n_states


def _as_f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def pack_params(A, B, Cc, D, S_rates, D_rates) -> np.ndarray:
    return np.concatenate(([A, B, Cc, D], np.asarray(S_rates, float).ravel(), np.asarray(D_rates, float).ravel()))


def rhs_host(model: int, theta: np.ndarray, y, num_psites: int) -> np.ndarray:
    ctx = get_context()
    th = _as_f64(theta).reshape(1, -1)
    yy = _as_f64(y).reshape(1, -1)
    S, P = n_states(model, num_psites), n_params(model, num_psites)
    if th.shape[1] != P or yy.shape[1] != S:
        raise ValueError(f"expected {P} parameters and {S} states, got {th.shape[1]} and {yy.shape[1]}")
    out = np.empty_like(yy)
    ctx.check(ctx.lib.pk_rhs_protein_batch_host(ctx.handle, model, int(num_psites), 1, th.ctypes.data, yy.ctypes.data,
                                                out.ctypes.data))
    return out[0]


def solve_host(model: int, params, init_cond, num_psites: int, t, normalize=None, **opt_kw):
    """(sol[T, S], flat) for one parameter vector -- the reference's ``solve_ode`` contract."""
    ctx = get_context()
    n = int(num_psites)
    th = _as_f64(params).reshape(1, -1)           # tuple / list / ndarray accepted (sensitivity/analysis.py:192 passes a tuple)
    y0 = _as_f64(init_cond).reshape(-1)
    tt = _as_f64(np.atleast_1d(t)).reshape(-1)    # normest.py:55 passes np.atleast_1d(tpts)
    S, P = n_states(model, n), n_params(model, n)
    if th.shape[1] != P:
        raise ValueError(f"params must hold {P} values for {n} sites, got {th.shape[1]}")
    if y0.shape[0] != S:
        raise ValueError(f"init_cond must hold {S} values, got {y0.shape[0]}")
    T = tt.shape[0]
    F = flat_len(model, n, T)
    sol = np.empty((1, T, S)); flat = np.empty((1, F))
    status = np.zeros(1, dtype=np.int32)
    kw = dict(config.SOLVER_OPTS); kw.update(opt_kw)
    norm = config.NORMALIZE_MODEL_OUTPUT if normalize is None else bool(normalize)
    opts = _capi.default_opts(clip_nonneg=1, normalize=int(norm), **kw)
    ctx.check(ctx.lib.pk_solve_protein_batch_host(ctx.handle, model, n, 1, th.ctypes.data, y0.ctypes.data, 0, tt.ctypes.data, T,
                                                  C.byref(opts), sol.ctypes.data, flat.ctypes.data, None, 0,
                                                  status.ctypes.data, None))
    # like odeint, solver trouble does not raise: flagged rows are NaN and callers test np.isfinite
    return sol[0], flat[0]


def solve_jac_host(model: int, params, init_cond, num_psites: int, t, normalize=None, **opt_kw):
    """(flat [F], dflat [F, P]) for one parameter vector: the model output the reference's fits compare with data, and its parameter
    Jacobian from the same integration (forward sensitivities, csrc/pk_sens.hpp) -- the ``jac=`` callable scipy.optimize.curve_fit
    accepts, where the reference lets curve_fit difference ``solve_ode`` 1 + P times (paramest/normest.py:167-326).
    Raises PhoskinError (PK_ERR_UNSUPPORTED) for sizes without a sensitivity kernel (``batch.sens_available``)."""
    ctx = get_context()
    n = int(num_psites)
    th = _as_f64(params).reshape(1, -1)
    y0 = _as_f64(init_cond).reshape(-1)
    tt = _as_f64(np.atleast_1d(t)).reshape(-1)
    S, P = n_states(model, n), n_params(model, n)
    if th.shape[1] != P:
        raise ValueError(f"params must hold {P} values for {n} sites, got {th.shape[1]}")
    if y0.shape[0] != S:
        raise ValueError(f"init_cond must hold {S} values, got {y0.shape[0]}")
    T = tt.shape[0]
    F = flat_len(model, n, T)
    flat = np.empty((1, F)); dflat = np.empty((1, F, P))
    status = np.zeros(1, dtype=np.int32)
    kw = {k: v for k, v in dict(config.SOLVER_OPTS, **opt_kw).items() if k in ("rtol", "atol", "h0", "max_steps")}
    norm = config.NORMALIZE_MODEL_OUTPUT if normalize is None else bool(normalize)
    opts = _capi.default_opts(clip_nonneg=1, normalize=int(norm), **kw)
    ctx.check(ctx.lib.pk_solve_protein_sens_batch_host(ctx.handle, model, n, 1, th.ctypes.data, y0.ctypes.data, 0, tt.ctypes.data, T, C.byref(opts),
                                                       flat.ctypes.data, dflat.ctypes.data, status.ctypes.data, None))
    return flat[0], dflat[0]
