"""Batched multistart least squares: the numerical core of ``paramest.normest._curve_fit_multistart`` (normest.py:167-326).

The reference runs 24-48 independent ``scipy.optimize.curve_fit`` (TRF, ``x_scale='jac'``) calls; each TRF iteration costs 1 + P
``solve_ode`` calls for a 2-point finite-difference Jacobian, one at a time.  Here ALL starts advance in lockstep: every
iteration is ONE launch of ``n_active * (1 + P)`` replicas (``solve_ode_batch`` returning the ``flat`` observable vectors), the
small dense algebra (P <= 64) stays on the host.

* start points: same construction and the same NumPy RNG stream as the reference (base, n/3 Gaussian jitters of 10 % of the
  range, stratified uniform for the rest; seed + hash(gene)), so a given (gene, seed) yields the reference's start list;
* model: ``[flat(p) ; lam / P * p**2]`` against ``[target ; 0]`` with ``sigma`` weights (normest.py:53-60, 403-423); randmod is
  fitted in log space (normest.py:54, 367-369);
* optimiser: bounded Levenberg-Marquardt (Marquardt scaling = the 'jac' scaling of the reference's call, projection on the box,
  gain-ratio damping).  It is not SciPy's TRF: iterates differ, minima of well-posed problems agree (tests compare the costs).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .. import batch


@dataclass
class FitResult:
    popt: np.ndarray            # best parameters (in the fitted space: log for randmod)
    pcov: Optional[np.ndarray]  # (J^T J)^-1 [* s^2] at the optimum
    score: float                # score_fit of the best start (config/config.py:176-226)
    cost: np.ndarray            # final 0.5 * ||r||^2 per start
    p_all: np.ndarray           # final parameters of every start
    n_iter: int
    n_solves: int


def multistart_candidates(gene: str, base_p0, lb, ub, n_starts: int = 24, jitter_frac: float = 0.10, seed: int = 42) -> np.ndarray:
    """Start list of normest.py:217-265, draw for draw."""
    lb = np.asarray(lb, float); ub = np.asarray(ub, float)
    if not (np.all(np.isfinite(lb)) and np.all(np.isfinite(ub))):
        raise ValueError("free_bounds must be finite for multistart sampling.")
    rng = np.random.default_rng(int(seed + (sum(ord(c) for c in str(gene)) % 1000003)))
    base = np.clip(np.asarray(base_p0, float).copy(), lb, ub)
    out = [base]
    span = ub - lb
    span[span <= 0] = 1.0
    for _ in range(max(0, n_starts // 3)):
        out.append(np.clip(base + (jitter_frac * span) * rng.normal(0.0, 1.0, size=base.shape[0]), lb, ub))
    remaining = max(0, n_starts - len(out))
    if remaining > 0:
        d = base.shape[0]
        U = np.empty((remaining, d))
        for j in range(d):
            u = (np.arange(remaining) + rng.random(remaining)) / float(remaining)
            rng.shuffle(u)
            U[:, j] = u
        out.extend(list(lb + U * (ub - lb)))
    return np.stack(out)


def curve_fit_multistart_batch(model: str, init_cond, num_psites: int, time_points, target, base_p0, bounds: Tuple, sigma=None,
                               lam: float = 0.0, gene: str = "", n_starts: int = 24, jitter_frac: float = 0.10, seed: int = 42,
                               max_iter: int = 100, ftol: float = 1e-10, xtol: float = 1e-10, absolute_sigma: bool = True,
                               **solver_kw) -> FitResult:
    """Fit ``flat(p)`` to ``target`` (+ ridge term ``lam``) from ``n_starts`` start points at once.  ``bounds = (lb, ub)`` in the fitted
    space; ``sigma`` covers the data block and, when ``lam > 0``, the P regularisation rows as well (as in the reference)."""
    log_space = (model == "randmod")
    lb, ub = (np.asarray(b, float) for b in bounds)
    P0 = multistart_candidates(gene, base_p0, lb, ub, n_starts, jitter_frac, seed)
    n_s, P = P0.shape
    target = np.asarray(target, float)
    Nd = target.size
    use_reg = lam > 0.0
    Nr = Nd + (P if use_reg else 0)
    sig = np.ones(Nr) if sigma is None else np.asarray(sigma, float)
    if sig.size != Nr:
        raise ValueError(f"sigma must hold {Nr} entries")
    tfull = np.concatenate([target, np.zeros(P)]) if use_reg else target
    n_solves = 0

    def residuals(Pm):
        """Pm [m, P] -> r [m, Nr]  (one launch)."""
        nonlocal n_solves
        theta = np.exp(Pm) if log_space else Pm
        flat = batch.solve_ode_batch(model, theta, init_cond, num_psites, time_points, want_sol=False, want_flat=True, **solver_kw).flat.cpu().numpy()
        n_solves += Pm.shape[0]
        f = np.concatenate([flat, (lam / P) * Pm ** 2], axis=1) if use_reg else flat
        r = (f - tfull[None, :]) / sig[None, :]
        return np.where(np.isfinite(r), r, 1e6)              # failed solves are very bad, not fatal

    p = P0.copy()
    r = residuals(p)
    cost = 0.5 * np.sum(r * r, axis=1)
    mu = np.full(n_s, 1e-3)
    active = np.ones(n_s, bool)
    J = np.zeros((n_s, Nr, P))
    it = 0
    for it in range(1, max_iter + 1):
        idx = np.where(active)[0]
        if idx.size == 0:
            break
        # forward-difference Jacobian (SciPy's '2-point' rule: h = sqrt(eps) * max(1, |p|), flipped at the upper bound)
        h = np.sqrt(np.finfo(float).eps) * np.maximum(1.0, np.abs(p[idx]))
        h = np.where(p[idx] + h > ub[None, :], -h, h)
        Pp = np.repeat(p[idx], P, axis=0)
        rows = np.arange(idx.size * P); cols = np.tile(np.arange(P), idx.size)
        Pp[rows, cols] += h.reshape(-1)
        rp = residuals(Pp).reshape(idx.size, P, Nr)
        J[idx] = np.transpose((rp - r[idx][:, None, :]) / h[:, :, None], (0, 2, 1))
        # Levenberg-Marquardt trial steps, still in lockstep: every pending start proposes one step, ONE launch evaluates them all;
        # the rejected ones raise their damping and go again
        G = {}; AA = {}; DD = {}; FR = {}
        pending = []
        for s_ in idx:
            g = J[s_].T @ r[s_]
            A = J[s_].T @ J[s_]
            free = ~(((p[s_] <= lb) & (g > 0)) | ((p[s_] >= ub) & (g < 0)))
            if not free.any() or np.linalg.norm(g[free]) < 1e-14 * max(1.0, cost[s_]):
                active[s_] = False
                continue
            G[s_], AA[s_], DD[s_], FR[s_] = g, A, np.maximum(np.sqrt(np.diag(A)), 1e-12), free      # Marquardt scaling ('jac')
            pending.append(s_)
        for _ in range(12):
            if not pending:
                break
            trial = []
            for s_ in pending:
                free = FR[s_]
                Af = AA[s_][np.ix_(free, free)] + mu[s_] * np.diag(DD[s_][free] ** 2)
                step = np.zeros(P)
                try:
                    step[free] = -np.linalg.solve(Af, G[s_][free])
                except np.linalg.LinAlgError:
                    pass
                trial.append(np.clip(p[s_] + step, lb, ub))
            trial = np.stack(trial)
            rn_all = residuals(trial)
            still = []
            for k_, s_ in enumerate(pending):
                pn, rn = trial[k_], rn_all[k_]
                cn = 0.5 * rn @ rn
                dp = pn - p[s_]
                pred = -(G[s_] @ dp + 0.5 * dp @ AA[s_] @ dp)
                rho = (cost[s_] - cn) / pred if pred > 0 else -1.0
                if cn < cost[s_] and rho > 1e-4:
                    dx = np.linalg.norm(dp); dc = cost[s_] - cn
                    p[s_], r[s_], cost[s_] = pn, rn, cn
                    mu[s_] = max(mu[s_] * (1.0 / 3.0 if rho > 0.75 else 1.0), 1e-12)
                    if dc <= ftol * max(cn, 1e-300) or dx <= xtol * (xtol + np.linalg.norm(pn)):
                        active[s_] = False
                else:
                    mu[s_] *= 4.0
                    still.append(s_)
            pending = still
        for s_ in pending:                 # no acceptable step within the damping budget: this start has converged / stalled
            active[s_] = False
    # score every start like the reference (solve at popt, score_fit against the un-regularised target) and keep the best
    theta = np.exp(p) if log_space else p
    flat = batch.solve_ode_batch(model, theta, init_cond, num_psites, time_points, want_sol=False, want_flat=True, **solver_kw).flat
    scores = batch.score_fit_batch(theta, target, flat).cpu().numpy()
    scores = np.where(np.isfinite(scores), scores, np.inf)
    best = int(np.argmin(scores))
    Jb = J[best]
    try:
        pcov = np.linalg.inv(Jb.T @ Jb)
        if not absolute_sigma and Nr > P:
            pcov = pcov * (2.0 * cost[best] / (Nr - P))
    except np.linalg.LinAlgError:
        pcov = None
    return FitResult(popt=p[best].copy(), pcov=pcov, score=float(scores[best]), cost=cost, p_all=p, n_iter=it, n_solves=n_solves)
