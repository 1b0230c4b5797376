"""Drop-in for the reference's ``global_model/lossfn.py``: ``LOSS_FN`` with its positional argument list, on the GPU.

  sq, huber, pseudo_huber, log_cosh, cauchy_loss, poisson_scaled_mse, geman_mcclure, charbonnier     lossfn.py:28-110  (host, vectorised)
  loss_function_noncomb(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna,
                        p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map, prot_base_idx, rna_base_idx, pho_base_idx)   lossfn.py:114-247
  loss_function_comb(... same list ...)                                                               lossfn.py:250-382
  LOSS_FN                                                                                              lossfn.py:386

``Y`` [T, S] gives the reference's 3-tuple of floats; ``Y`` [B, T, S] gives three arrays [B] (one launch for all B trajectories).  The
point loss is ``config.LOSS_MODE`` read at call time (the reference freezes it at import).  ``optproblem.GlobalODEBatch`` does not come
through here: it keeps the loss tables in HBM and never brings ``Y`` to the host (``NetworkEngine.objective_batch``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _capi
from ..batch import get_context
from . import config

EPS = 1e-9


def sq(diff):
    return diff * diff


def huber(diff, delta=1.0):
    a = np.abs(diff)
    return np.where(a <= delta, 0.5 * diff * diff, delta * (a - 0.5 * delta))


def pseudo_huber(diff, delta=1.0):
    x = diff / delta
    return (delta * delta) * (np.sqrt(1.0 + x * x) - 1.0)


def charbonnier(diff, eps=1e-3):
    return np.sqrt(diff * diff + eps * eps) - eps


def log_cosh(diff):
    s = np.abs(diff)
    with np.errstate(over="ignore"):
        return np.where(s > 20.0, s - 0.69314718056, np.log(np.cosh(np.minimum(s, 20.0))))


def cauchy_loss(diff, c=1.0):
    return np.log(1.0 + (diff / c) ** 2)


def poisson_scaled_mse(diff, pred_val, eps=1e-6):
    return (diff * diff) / (np.abs(pred_val) + eps)


def geman_mcclure(diff, delta=1.0):
    x2 = diff * diff
    return x2 / (x2 + delta * delta)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _loss(comb, Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
          prot_base_idx, rna_base_idx, pho_base_idx, loss_mode=None):
    ctx = get_context()
    Yh = _f64(Y.detach().cpu().numpy() if hasattr(Y, "detach") else Y)
    single = Yh.ndim == 2
    if single:
        Yh = Yh[None]
    if Yh.ndim != 3:
        raise ValueError("Y must be [T, S] or [B, T, S]")
    B, T, S = Yh.shape
    ints = [_i32(a) for a in (p_prot, t_prot, p_rna, t_rna, p_pho, s_pho, t_pho)]
    flts = [_f64(a) for a in (obs_prot, w_prot, obs_rna, w_rna, obs_pho, w_pho)]
    (pp, tp, pr, tr, ph, sh, th), (op, wp, orr, wr, oph, wph) = ints, flts
    if not (pp.size == tp.size == op.size == wp.size and pr.size == tr.size == orr.size == wr.size and ph.size == sh.size == th.size == oph.size == wph.size):
        raise ValueError("the index / observation / weight arrays of a modality must have equal lengths")
    pm = _i32(prot_map).reshape(-1, 2)
    d = _capi.LossData(pp.size, pr.size, ph.size, pp.ctypes.data, tp.ctypes.data, op.ctypes.data, wp.ctypes.data, pr.ctypes.data, tr.ctypes.data,
                       orr.ctypes.data, wr.ctypes.data, ph.ctypes.data, sh.ctypes.data, th.ctypes.data, oph.ctypes.data, wph.ctypes.data,
                       int(prot_base_idx), int(rna_base_idx), int(pho_base_idx))
    out = np.empty((B, 3))
    mode = config.LOSS_MODE if loss_mode is None else int(loss_mode)
    ctx.check(ctx.lib.pk_loss_fn_batch_host(ctx.handle, int(comb), mode, B, Yh.ctypes.data, T, S, C.byref(d), pm.ctypes.data, pm.shape[0], out.ctypes.data))
    if single:
        return float(out[0, 0]), float(out[0, 1]), float(out[0, 2])
    return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy()


def loss_function_noncomb(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
                          prot_base_idx, rna_base_idx, pho_base_idx):
    """(loss_protein, loss_rna, loss_phospho) for the linear state layouts (topologies 0 / 1 / 4): weighted point losses of the fold changes
    total protein / RNA / phospho site against their baselines (floors 1e-9)."""
    return _loss(False, Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
                 prot_base_idx, rna_base_idx, pho_base_idx)


def loss_function_comb(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
                       prot_base_idx, rna_base_idx, pho_base_idx):
    """The same for the combinatorial topology: ``prot_map[:, 1]`` holds 2^n state counts, a site's signal sums the states with its bit."""
    return _loss(True, Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
                 prot_base_idx, rna_base_idx, pho_base_idx)


def LOSS_FN(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
            prot_base_idx, rna_base_idx, pho_base_idx):
    """Dispatch on ``config.MODEL`` at call time (lossfn.py:386 binds at import)."""
    f = loss_function_comb if config.MODEL == 2 else loss_function_noncomb
    return f(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
             prot_base_idx, rna_base_idx, pho_base_idx)
