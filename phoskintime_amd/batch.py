"""Batched siblings of the reference's per-protein call surface, running on one MI355X.

``solve_ode_batch`` evaluates ``models.solve_ode`` (reference models/{distmod,succmod,randmod}.py ``solve_ode``) for a whole
[B, P] matrix of parameter vectors in one kernel launch; ``rhs_batch`` / ``jacobian_batch`` do the same for
``ode_core`` / ``ode_system`` and the (analytic) Jacobian.  PyTorch is used only for HBM allocations, the current HIP
stream and (in ``phoskintime_amd.distributed``) the RCCL all-gather; every kernel is reached through the C ABI
(include/phoskin.h) with raw device pointers.
"""
from __future__ import annotations

import ctypes as C
import functools
import threading
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _capi
from ._capi import (DIST, SUCC, RAND, MODEL_IDS, METRICS, Context, PhoskinError, default_opts)

ArrayLike = Union[np.ndarray, torch.Tensor, Sequence[float]]

_tls = threading.local()


def get_context(device: Optional[int] = None) -> Context:
    """The calling THREAD's context for a GPU (created on first use, destroyed with the thread).  include/phoskin.h asks for one context
    per thread: a context carries the stream of the last ``pk_set_stream`` and the staging buffers of the ``_host`` entry points, and
    ctypes releases the GIL during calls, so two threads must never share one (the C side additionally serialises ``_host`` calls per
    context).  Fails loudly when no GPU / library is present."""
    if not torch.cuda.is_available():
        # still go through load() first so that a missing .so is reported as such
        _capi.load()
        raise PhoskinError("phoskintime_amd needs a HIP GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    dev = torch.cuda.current_device() if device is None else int(device)
    contexts = _tls.__dict__.setdefault("contexts", {})
    ctx = contexts.get(dev)
    if ctx is None:
        ctx = contexts[dev] = Context(dev)
    return ctx


def model_id(model: Union[int, str]) -> int:
    if isinstance(model, str):
        return MODEL_IDS[model]
    return int(model)


# shapes are pure functions of small integers and sit on the one-theta-per-call path (models.solve_ode): memoised
@functools.lru_cache(maxsize=4096)
def _n_states(mid: int, n: int) -> int:
    return _shape_or_raise(_capi.load().pk_protein_n_states(mid, n))


@functools.lru_cache(maxsize=4096)
def _n_params(mid: int, n: int) -> int:
    return _shape_or_raise(_capi.load().pk_protein_n_params(mid, n))


@functools.lru_cache(maxsize=4096)
def _flat_len(mid: int, n: int, T: int) -> int:
    return _shape_or_raise(_capi.load().pk_protein_flat_len(mid, n, T))


def n_states(model, n_sites: int) -> int:
    return _n_states(model_id(model), int(n_sites))


def n_params(model, n_sites: int) -> int:
    return _n_params(model_id(model), int(n_sites))


def flat_len(model, n_sites: int, T: int) -> int:
    return _flat_len(model_id(model), int(n_sites), int(T))


def _shape_or_raise(v: int) -> int:
    if v < 0:
        raise ValueError("invalid model / n_sites / T")
    return v


def _dev_f64(x: ArrayLike, device: torch.device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.float64)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), device=device)
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(None)


@dataclass
class BatchResult:
    """Outputs of ``solve_ode_batch`` (tensors live on the GPU; ``None`` where not requested)."""
    sol: Optional[torch.Tensor]       # [B, T, S]  clipped / normalised like the reference's return value
    flat: Optional[torch.Tensor]      # [B, F]     [R(t5..), P(t0..), sites site-major]
    metric: Optional[torch.Tensor]    # [B]        _compute_Y
    status: torch.Tensor              # [B] int32  0 ok | 1 non-finite | 2 max steps | 4 step underflow
    n_steps: torch.Tensor             # [B, 2] int32 accepted, rejected


def solve_ode_batch(model, theta: ArrayLike, init_cond: ArrayLike, num_psites: int, t: ArrayLike, *,
                    want_sol: bool = True, want_flat: bool = True, metric: Optional[str] = None,
                    method: Union[str, int, None] = None, linsolve: Union[str, int, None] = None,
                    rtol: Optional[float] = None, atol: Optional[float] = None, h0: Optional[float] = None,
                    rk4_h: Optional[float] = None, max_steps: Optional[int] = None,
                    clip_nonneg: bool = True, normalize: bool = False, stage_form: int = 0, kernel: Union[str, int, None] = None,
                    device: Optional[int] = None, out: Optional[BatchResult] = None) -> BatchResult:
    """Integrate B replicas of one per-protein model.  ``theta`` is [B, P]; ``init_cond`` [S] (shared) or [B, S].

    Asynchronous on torch's current stream.  Semantics per replica are those of the reference's ``solve_ode``
    (odeint -> clip >= 0 -> optional / y0 -> flat); the integrator is the engine's own (include/phoskin.h).
    ``kernel`` = "auto" | "group" | "tpr": small systems switch kernel family by batch size under "auto", so a sharded run that must
    reproduce the single-GPU bits pins one family on every rank."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    mid = model_id(model)
    n = int(num_psites)
    S, P = n_states(mid, n), n_params(mid, n)
    th = _dev_f64(theta, dev)
    if th.dim() == 1:
        th = th.unsqueeze(0)
    if th.dim() != 2 or th.shape[1] != P:
        raise ValueError(f"theta must be [B, {P}] for model {mid} with {n} sites, got {tuple(th.shape)}")
    B = th.shape[0]
    y0 = _dev_f64(init_cond, dev)
    if y0.shape == (S,):
        batched = 0
    elif y0.shape == (B, S):
        batched = 1
    else:
        raise ValueError(f"init_cond must be [{S}] or [{B}, {S}], got {tuple(y0.shape)}")
    tt = _dev_f64(np.atleast_1d(t) if not isinstance(t, torch.Tensor) else t, dev).reshape(-1)
    T = tt.numel()
    if T < 1:
        raise ValueError("t must hold at least one time point")
    F = flat_len(mid, n, T)
    opts = default_opts(method=method, linsolve=linsolve, rtol=rtol, atol=atol, h0=h0, rk4_h=rk4_h, max_steps=max_steps,
                        clip_nonneg=int(bool(clip_nonneg)), normalize=int(bool(normalize)), stage_form=int(stage_form), kernel=kernel)
    if out is None:
        out = BatchResult(
            sol=torch.empty((B, T, S), dtype=torch.float64, device=dev) if want_sol else None,
            flat=torch.empty((B, F), dtype=torch.float64, device=dev) if want_flat else None,
            metric=torch.empty((B,), dtype=torch.float64, device=dev) if metric is not None else None,
            status=torch.zeros((B,), dtype=torch.int32, device=dev),
            n_steps=torch.zeros((B, 2), dtype=torch.int32, device=dev))
    if B == 0:
        return out
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rc = ctx.lib.pk_solve_protein_batch(ctx.handle, mid, n, B, _ptr(th), _ptr(y0), batched, _ptr(tt), T, C.byref(opts),
                                        _ptr(out.sol), _ptr(out.flat), _ptr(out.metric),
                                        METRICS[metric] if metric is not None else 0, _ptr(out.status), _ptr(out.n_steps))
    ctx.check(rc)
    # keep inputs alive until the kernel has consumed them (they may be temporaries made above)
    out._keepalive = (th, y0, tt)  # type: ignore[attr-defined]
    return out


def sens_available(model, num_psites: int) -> bool:
    """True where ``solve_ode_sens_batch`` has a kernel (distmod / succmod n <= 62, randmod n <= 7).  Pure host arithmetic."""
    return bool(_capi.load().pk_protein_sens_available(model_id(model), int(num_psites)))


@dataclass
class SensResult:
    flat: torch.Tensor                # [B, F]
    dflat: torch.Tensor               # [B, F, P]  d flat / d theta
    status: torch.Tensor              # [B] int32
    n_steps: torch.Tensor             # [B, 2] int32


def solve_ode_sens_batch(model, theta: ArrayLike, init_cond: ArrayLike, num_psites: int, t: ArrayLike, *,
                         rtol: Optional[float] = None, atol: Optional[float] = None, h0: Optional[float] = None, max_steps: Optional[int] = None,
                         clip_nonneg: bool = True, normalize: bool = False, device: Optional[int] = None) -> SensResult:
    """``flat`` of the reference's ``solve_ode`` AND its Jacobian d flat / d theta for B parameter vectors from one launch (forward
    sensitivities, csrc/pk_sens.hpp) -- what scipy's curve_fit obtains from 1 + P calls of ``solve_ode`` (paramest/normest.py:167-326).
    Raises ``PhoskinError`` (PK_ERR_UNSUPPORTED) for sizes without a kernel: see ``sens_available``."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    mid = model_id(model)
    n = int(num_psites)
    S, P = n_states(mid, n), n_params(mid, n)
    th = _dev_f64(theta, dev)
    if th.dim() == 1:
        th = th.unsqueeze(0)
    if th.dim() != 2 or th.shape[1] != P:
        raise ValueError(f"theta must be [B, {P}] for model {mid} with {n} sites, got {tuple(th.shape)}")
    B = th.shape[0]
    y0 = _dev_f64(init_cond, dev)
    if y0.shape == (S,):
        batched = 0
    elif y0.shape == (B, S):
        batched = 1
    else:
        raise ValueError(f"init_cond must be [{S}] or [{B}, {S}], got {tuple(y0.shape)}")
    tt = _dev_f64(np.atleast_1d(t) if not isinstance(t, torch.Tensor) else t, dev).reshape(-1)
    T = tt.numel()
    if T < 1:
        raise ValueError("t must hold at least one time point")
    F = flat_len(mid, n, T)
    opts = default_opts(rtol=rtol, atol=atol, h0=h0, max_steps=max_steps, clip_nonneg=int(bool(clip_nonneg)), normalize=int(bool(normalize)))
    out = SensResult(flat=torch.empty((B, F), dtype=torch.float64, device=dev), dflat=torch.empty((B, F, P), dtype=torch.float64, device=dev),
                     status=torch.zeros((B,), dtype=torch.int32, device=dev), n_steps=torch.zeros((B, 2), dtype=torch.int32, device=dev))
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.pk_solve_protein_sens_batch(ctx.handle, mid, n, B, _ptr(th), _ptr(y0), batched, _ptr(tt), T, C.byref(opts),
                                                  _ptr(out.flat), _ptr(out.dflat), _ptr(out.status), _ptr(out.n_steps)))
    out._keepalive = (th, y0, tt)  # type: ignore[attr-defined]
    return out


def rhs_batch(model, theta: ArrayLike, y: ArrayLike, num_psites: int, device: Optional[int] = None) -> torch.Tensor:
    """dy/dt for B (theta, y) pairs: reference ``ode_core`` / ``ode_system`` batched.  Returns [B, S] on the GPU."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    mid, n = model_id(model), int(num_psites)
    S, P = n_states(mid, n), n_params(mid, n)
    th = _dev_f64(theta, dev).reshape(-1, P)
    yy = _dev_f64(y, dev).reshape(-1, S)
    if th.shape[0] != yy.shape[0]:
        raise ValueError("theta and y must have the same batch size")
    out = torch.empty_like(yy)
    if th.shape[0]:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.check(ctx.lib.pk_rhs_protein_batch(ctx.handle, mid, n, th.shape[0], _ptr(th), _ptr(yy), _ptr(out)))
        out._keepalive = (th, yy)  # type: ignore[attr-defined]
    return out


def jacobian_batch(model, theta: ArrayLike, num_psites: int, device: Optional[int] = None) -> torch.Tensor:
    """Analytic Jacobian, row-major [B, S, S] on the GPU."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    mid, n = model_id(model), int(num_psites)
    S, P = n_states(mid, n), n_params(mid, n)
    th = _dev_f64(theta, dev).reshape(-1, P)
    out = torch.empty((th.shape[0], S, S), dtype=torch.float64, device=dev)
    if th.shape[0]:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.check(ctx.lib.pk_jacobian_protein_batch(ctx.handle, mid, n, th.shape[0], _ptr(th), _ptr(out)))
        out._keepalive = (th,)  # type: ignore[attr-defined]
    return out


def steady_state_batch(model, theta: ArrayLike, num_psites: int, device: Optional[int] = None):
    """Steady states y* [B, S] of dy/dt = J(theta) y + b(theta) for B parameter vectors, and status [B] (1 = singular J, NaN row).
    ``steady.initial_condition(n)`` of the reference (steady/init*.py) is the special case theta = ones."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    mid, n = model_id(model), int(num_psites)
    S, P = n_states(mid, n), n_params(mid, n)
    th = _dev_f64(theta, dev).reshape(-1, P)
    out = torch.empty((th.shape[0], S), dtype=torch.float64, device=dev)
    status = torch.zeros((th.shape[0],), dtype=torch.int32, device=dev)
    if th.shape[0]:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.check(ctx.lib.pk_steady_state_protein_batch(ctx.handle, mid, n, th.shape[0], _ptr(th), _ptr(out), _ptr(status)))
        out._keepalive = (th,)  # type: ignore[attr-defined]
    return out, status


def score_fit_batch(theta: ArrayLike, target: ArrayLike, prediction: ArrayLike, alpha: float = 1.0, beta: float = 1.0, gamma: float = 1.0,
                    delta: float = 1.0, mu: float = 1.0, device: Optional[int] = None) -> torch.Tensor:
    """``config.config.score_fit`` (reference config/config.py:176-226) for B candidates: theta [B, P], target [N], prediction [B, N]."""
    ctx = get_context(device)
    dev = torch.device("cuda", ctx.device)
    th = _dev_f64(theta, dev); pr = _dev_f64(prediction, dev); tg = _dev_f64(target, dev).reshape(-1)
    if th.dim() == 1:
        th = th.unsqueeze(0)
    if pr.dim() == 1:
        pr = pr.unsqueeze(0)
    B, P = th.shape
    N = tg.numel()
    if pr.shape != (B, N):
        raise ValueError(f"prediction must be [{B}, {N}]")
    out = torch.empty((B,), dtype=torch.float64, device=dev)
    w = (C.c_double * 5)(alpha, beta, gamma, delta, mu)
    if B:
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.check(ctx.lib.pk_score_fit_batch(ctx.handle, B, _ptr(th), P, _ptr(tg), _ptr(pr), N, C.cast(w, C.c_void_p), _ptr(out)))
        out._keepalive = (th, pr, tg)  # type: ignore[attr-defined]
    return out
