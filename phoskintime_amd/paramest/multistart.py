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


@dataclass
class RowsFit:
    p: np.ndarray               # [R, P] final parameters (fitted space)
    cost: np.ndarray            # [R] 0.5 * ||r||^2
    r: np.ndarray               # [R, Nr] final weighted residuals
    J: np.ndarray               # [R, Nr, P] last Jacobian of the weighted residuals
    n_iter: int
    n_solves: int


def fit_rows_batch(model: str, num_psites: int, time_points, P0, init_cond, target, sigma=None, lam=0.0, bounds=None,
                   max_iter: int = 100, ftol: float = 1e-10, xtol: float = 1e-10, **solver_kw) -> RowsFit:
    """R independent bounded least-squares problems in lockstep: row k fits ``[flat(p) ; lam_k / P * p**2]`` to ``[target_k ; 0]`` with
    weights ``sigma_k`` from the start point ``P0[k]``.  Rows may be the starts of one multistart fit, the (lambda, weight) grid of
    ``find_best_lambda``, bootstrap replicates, different proteins of the same size -- or any mix.

    P0 [R, P]; init_cond [S] or [R, S]; target [Nd] or [R, Nd]; sigma None, [Nr] or [R, Nr] with Nr = Nd (+ P when any lam > 0);
    lam scalar or [R]; bounds (lb, ub), each [P] or [R, P].

    Every iteration is ONE launch for the forward-difference Jacobians (n_active * P replicas) plus one launch per damping round
    for the trial points; the P x P algebra of all rows is batched numpy on the host."""
    log_space = (model == "randmod")
    P0 = np.atleast_2d(np.asarray(P0, float))
    R, P = P0.shape
    target = np.asarray(target, float)
    tgt = np.broadcast_to(target, (R, target.shape[-1])) if target.ndim == 1 else target
    Nd = tgt.shape[1]
    lam = np.broadcast_to(np.asarray(lam, float), (R,)).copy()
    use_reg = bool(np.any(lam > 0.0))
    Nr = Nd + (P if use_reg else 0)
    if sigma is None:
        sig = np.ones((R, Nr))
    else:
        sig = np.asarray(sigma, float)
        sig = np.broadcast_to(sig, (R, sig.shape[-1])) if sig.ndim == 1 else sig
        if sig.shape[1] != Nr:
            raise ValueError(f"sigma must hold {Nr} entries")
    tfull = np.concatenate([tgt, np.zeros((R, P))], axis=1) if use_reg else tgt
    lb, ub = (np.broadcast_to(np.asarray(b, float), (R, P)) for b in bounds)
    y0 = np.asarray(init_cond, float)
    y0_rows = y0.ndim == 2
    n_solves = 0

    def residuals(Pm, rows):
        """Pm [m, P] for the problems `rows` [m] -> weighted residuals [m, Nr]  (one launch)."""
        nonlocal n_solves
        theta = np.exp(Pm) if log_space else Pm
        flat = batch.solve_ode_batch(model, theta, y0[rows] if y0_rows else y0, num_psites, time_points, want_sol=False, want_flat=True,
                                     **solver_kw).flat.cpu().numpy()
        n_solves += Pm.shape[0]
        f = np.concatenate([flat, (lam[rows, None] / P) * Pm ** 2], axis=1) if use_reg else flat
        rr = (f - tfull[rows]) / sig[rows]
        return np.where(np.isfinite(rr), rr, 1e6)            # failed solves are very bad, not fatal

    p = np.clip(P0, lb, ub)
    allr = np.arange(R)
    r = residuals(p, allr)
    cost = 0.5 * np.sum(r * r, axis=1)
    mu = np.full(R, 1e-3)
    active = np.ones(R, bool)
    J = np.zeros((R, Nr, P))
    it = 0
    for it in range(1, max_iter + 1):
        idx = np.where(active)[0]
        if idx.size == 0:
            break
        # forward-difference Jacobian (SciPy's '2-point' rule: h = sqrt(eps) * max(1, |p|), flipped at the upper bound)
        h = np.sqrt(np.finfo(float).eps) * np.maximum(1.0, np.abs(p[idx]))
        h = np.where(p[idx] + h > ub[idx], -h, h)
        Pp = np.repeat(p[idx], P, axis=0)
        Pp[np.arange(idx.size * P), np.tile(np.arange(P), idx.size)] += h.reshape(-1)
        rp = residuals(Pp, np.repeat(idx, P)).reshape(idx.size, P, Nr)
        J[idx] = np.transpose((rp - r[idx][:, None, :]) / h[:, :, None], (0, 2, 1))
        Ja = J[idx]
        g = np.einsum("knp,kn->kp", Ja, r[idx])
        A = np.einsum("knp,knq->kpq", Ja, Ja)
        free = ~(((p[idx] <= lb[idx]) & (g > 0)) | ((p[idx] >= ub[idx]) & (g < 0)))
        gfree = np.where(free, g, 0.0)
        done = (~free.any(axis=1)) | (np.linalg.norm(gfree, axis=1) < 1e-14 * np.maximum(1.0, cost[idx]))
        active[idx[done]] = False
        DD = np.maximum(np.sqrt(np.einsum("kpp->kp", A)), 1e-12)                       # Marquardt scaling (the reference's x_scale='jac')
        pend = np.where(~done)[0]                                                      # positions inside idx
        # Levenberg-Marquardt trial steps, still in lockstep: every pending row proposes one step, ONE launch evaluates them all;
        # the rejected ones raise their damping and go again
        for _ in range(12):
            if pend.size == 0:
                break
            rows = idx[pend]
            fr = free[pend]
            Af = A[pend] * (fr[:, :, None] & fr[:, None, :])
            dg = np.where(fr, mu[rows, None] * DD[pend] ** 2, 1.0)                     # fixed variables: identity row, zero step
            Af[:, np.arange(P), np.arange(P)] += dg
            rhs = -gfree[pend]
            try:
                step = np.linalg.solve(Af, rhs[:, :, None])[:, :, 0]
            except np.linalg.LinAlgError:
                step = np.zeros_like(rhs)
                for k_ in range(pend.size):
                    try:
                        step[k_] = np.linalg.solve(Af[k_], rhs[k_])
                    except np.linalg.LinAlgError:
                        pass
            step = np.where(np.isfinite(step), step, 0.0)
            trial = np.clip(p[rows] + step, lb[rows], ub[rows])
            rn = residuals(trial, rows)
            cn = 0.5 * np.sum(rn * rn, axis=1)
            dp = trial - p[rows]
            pred = -(np.einsum("kp,kp->k", g[pend], dp) + 0.5 * np.einsum("kp,kpq,kq->k", dp, A[pend], dp))
            rho = np.where(pred > 0, (cost[rows] - cn) / np.where(pred > 0, pred, 1.0), -1.0)
            ok = (cn < cost[rows]) & (rho > 1e-4)
            dx = np.linalg.norm(dp, axis=1); dc = cost[rows] - cn
            conv = ok & ((dc <= ftol * np.maximum(cn, 1e-300)) | (dx <= xtol * (xtol + np.linalg.norm(trial, axis=1))))
            acc = rows[ok]
            p[acc], r[acc], cost[acc] = trial[ok], rn[ok], cn[ok]
            mu[acc] = np.maximum(mu[acc] * np.where(rho[ok] > 0.75, 1.0 / 3.0, 1.0), 1e-12)
            active[rows[conv]] = False
            mu[rows[~ok]] *= 4.0
            pend = pend[~ok]
        active[idx[pend]] = False          # no acceptable step within the damping budget: converged / stalled
    return RowsFit(p=p, cost=cost, r=r, J=J, n_iter=it, n_solves=n_solves)


def _pcov(Jb, cost_b, absolute_sigma):
    Nr, P = Jb.shape
    try:
        pcov = np.linalg.inv(Jb.T @ Jb)
        if not absolute_sigma and Nr > P:
            pcov = pcov * (2.0 * cost_b / (Nr - P))
        return pcov
    except np.linalg.LinAlgError:
        return None


def _scores(model, theta_space_p, init_cond, num_psites, time_points, target, solver_kw):
    """score_fit of every row against the un-regularised target, as the reference scores a fit (normest.py:93-100, 293-300)."""
    theta = np.exp(theta_space_p) if model == "randmod" else theta_space_p
    flat = batch.solve_ode_batch(model, theta, init_cond, num_psites, time_points, want_sol=False, want_flat=True, **solver_kw).flat
    sc = batch.score_fit_batch(theta, target, flat).cpu().numpy()
    return np.where(np.isfinite(sc), sc, np.inf)


def curve_fit_multistart_batch(model: str, init_cond, num_psites: int, time_points, target, base_p0, bounds: Tuple, sigma=None,
                               lam: float = 0.0, gene: str = "", n_starts: int = 24, jitter_frac: float = 0.10, seed: int = 42,
                               max_iter: int = 100, ftol: float = 1e-10, xtol: float = 1e-10, absolute_sigma: bool = True,
                               **solver_kw) -> FitResult:
    """Fit ``flat(p)`` to ``target`` (+ ridge term ``lam``) from ``n_starts`` start points at once.  ``bounds = (lb, ub)`` in the fitted
    space; ``sigma`` covers the data block and, when ``lam > 0``, the P regularisation rows as well (as in the reference)."""
    lb, ub = (np.asarray(b, float) for b in bounds)
    P0 = multistart_candidates(gene, base_p0, lb, ub, n_starts, jitter_frac, seed)
    fit = fit_rows_batch(model, num_psites, time_points, P0, init_cond, target, sigma=sigma, lam=lam, bounds=(lb, ub), max_iter=max_iter,
                         ftol=ftol, xtol=xtol, **solver_kw)
    # score every start like the reference (solve at popt, score_fit against the un-regularised target) and keep the best
    scores = _scores(model, fit.p, init_cond, num_psites, time_points, target, solver_kw)
    best = int(np.argmin(scores))
    return FitResult(popt=fit.p[best].copy(), pcov=_pcov(fit.J[best], fit.cost[best], absolute_sigma), score=float(scores[best]), cost=fit.cost,
                     p_all=fit.p, n_iter=fit.n_iter, n_solves=fit.n_solves)


def find_best_lambda_batch(model: str, target, p0, time_points, free_bounds: Tuple, init_cond, num_psites: int, weight_options: dict,
                           lambdas=None, max_iter: int = 100, **solver_kw):
    """``paramest.normest.find_best_lambda`` + ``worker_find_lambda`` (normest.py:36-166): for every lambda in ``lambdas`` (default
    ``np.logspace(-2, 0, 10)``) and every weighting ``sigma`` in ``weight_options`` (``models.weights.get_weight_options`` with
    ``use_regularization=True``: Nd + P entries each) one fit from ``p0``; the reference runs these 10 x W ``curve_fit`` calls in a
    process pool, here they are the rows of ONE lockstep batch.  Each fit is scored with ``score_fit`` against the un-regularised
    target; returns ``(best_lambda, best_weight_key, scores)`` with ``scores[i, j]`` for ``lambdas[i]``, ``list(weight_options)[j]``."""
    lambdas = np.logspace(-2, 0, 10) if lambdas is None else np.asarray(lambdas, float)
    keys = list(weight_options)
    if not keys:
        raise ValueError("weight_options is empty")
    p0 = np.asarray(p0, float)
    L, W, P = lambdas.size, len(keys), p0.size
    sig = np.stack([np.asarray(weight_options[k], float) for k in keys])               # [W, Nd + P]
    fit = fit_rows_batch(model, num_psites, time_points, np.tile(p0, (L * W, 1)), init_cond, target, sigma=np.tile(sig, (L, 1)),
                         lam=np.repeat(lambdas, W), bounds=free_bounds, max_iter=max_iter, **solver_kw)
    scores = _scores(model, fit.p, init_cond, num_psites, time_points, target, solver_kw).reshape(L, W)
    # the reference keeps the first strict improvement while scanning weights inside a lambda, then lambdas: same tie-breaking
    best_per_lam = np.argmin(scores, axis=1)
    best_l = int(np.argmin(scores[np.arange(L), best_per_lam]))
    return float(lambdas[best_l]), keys[int(best_per_lam[best_l])], scores


def bootstrap_fit_batch(model: str, target_fit, popt, time_points, free_bounds: Tuple, init_cond, num_psites: int, sigma=None,
                        lam: float = 0.0, bootstraps: int = 10, noise: float = 0.05, rng=None, absolute_sigma: bool = True,
                        max_iter: int = 100, **solver_kw):
    """The bootstrap loop of ``normest`` (normest.py:488-523): ``bootstraps`` refits from ``popt`` of ``target_fit * (1 + N(0, noise))``
    (noise on the regularisation zeros is a no-op, as in the reference), all replicates in one lockstep batch.
    Returns (mean of the replicate estimates, mean of their covariances or None, all estimates [bootstraps, P])."""
    rng = np.random if rng is None else rng                                             # the reference draws from the global NumPy state
    tf = np.asarray(target_fit, float)
    popt = np.asarray(popt, float)
    P = popt.size
    noisy = np.stack([tf * (1 + rng.normal(0, noise, size=tf.shape)) for _ in range(bootstraps)])
    Nd = tf.size - (P if lam > 0.0 else 0)
    fit = fit_rows_batch(model, num_psites, time_points, np.tile(popt, (bootstraps, 1)), init_cond, noisy[:, :Nd], sigma=sigma, lam=lam,
                         bounds=free_bounds, max_iter=max_iter, **solver_kw)
    covs = [c for c in (_pcov(fit.J[k], fit.cost[k], absolute_sigma) for k in range(bootstraps)) if c is not None]
    return fit.p.mean(axis=0), (np.mean(covs, axis=0) if covs else None), fit.p


def build_free_bounds(model: str, bounds: dict, num_psites: int, eps: float = 1e-8):
    """(lb, ub) in the fitted space from the config bound dict {"A","B","C","D","S(i)","D(i)"} -- normest.py:350-385: randmod has one
    D(i) entry per non-empty site subset (2^n - 1) and is fitted in log space with the lower bounds floored at eps."""
    n = int(num_psites)
    nd = (1 << n) - 1 if model == "randmod" else n
    lo = [bounds[k][0] for k in "ABCD"] + [bounds["S(i)"][0]] * n + [bounds["D(i)"][0]] * nd
    hi = [bounds[k][1] for k in "ABCD"] + [bounds["S(i)"][1]] * n + [bounds["D(i)"][1]] * nd
    if model == "randmod":
        return np.log(np.maximum(lo, eps)), np.log(np.asarray(hi, float))
    return np.asarray(lo, float), np.asarray(hi, float)


def normest_core(model: str, gene: str, target, init_cond, num_psites: int, time_points, bounds: dict, weight_options: dict,
                 bootstraps: int = 0, use_regularization: bool = True, n_starts: int = 48, lambdas=None, seed: int = 42,
                 absolute_sigma: bool = True, **solver_kw):
    """The numerical pipeline of ``paramest.normest.normest`` (normest.py:328-600) without its file / plot side effects:
      1. p0 ~ U(lb, ub) drawn from ``np.random.seed(42)`` one parameter at a time (normest.py:389-392);
      2. lambda / weighting scan from p0 (find_best_lambda, one lockstep batch);
      3. multistart fit (n_starts = 48) with the winning lambda and weighting;
      4. optional bootstrap refits (their mean replaces the estimate, normest.py:509);
      5. final solve at the estimate.
    ``weight_options`` is what ``models.weights.get_weight_options(..., use_regularization, reg_len=P, ...)`` returns (host-side data
    preparation of the reference, used as is).  Returns a dict: param_final (physical space), popt, pcov, lambda_reg, weight_key,
    score, sol [T, S], fit (flat), error (mean squared error as normest.py:596), regularization_term (normest.py:598)."""
    lb, ub = build_free_bounds(model, bounds, num_psites)
    rs = np.random.RandomState(seed)
    p0 = np.array([rs.uniform(low=l, high=u) for l, u in zip(lb, ub)])
    target = np.asarray(target, float)
    P = p0.size
    if use_regularization:
        lam, wkey, _ = find_best_lambda_batch(model, target, p0, time_points, (lb, ub), init_cond, num_psites, weight_options, lambdas=lambdas,
                                              **solver_kw)
    else:
        # without the ridge rows the reference still scans the weightings at its regularised model_func; here: lambda = 0 rows
        keys = list(weight_options)
        sig = np.stack([np.asarray(weight_options[k], float) for k in keys])
        fit = fit_rows_batch(model, num_psites, time_points, np.tile(p0, (len(keys), 1)), init_cond, target, sigma=sig, lam=0.0, bounds=(lb, ub),
                             **solver_kw)
        lam, wkey = 0.0, keys[int(np.argmin(_scores(model, fit.p, init_cond, num_psites, time_points, target, solver_kw)))]
    sigma = np.asarray(weight_options[wkey], float)
    res = curve_fit_multistart_batch(model, init_cond, num_psites, time_points, target, p0, (lb, ub), sigma=sigma, lam=lam, gene=gene,
                                     n_starts=n_starts, seed=seed, absolute_sigma=absolute_sigma, **solver_kw)
    popt, pcov = res.popt, res.pcov
    if bootstraps > 0:
        tf = np.concatenate([target, np.zeros(P)]) if lam > 0.0 else target
        popt, pcov, _ = bootstrap_fit_batch(model, tf, popt, time_points, (lb, ub), init_cond, num_psites, sigma=sigma, lam=lam,
                                            bootstraps=bootstraps, rng=rs, absolute_sigma=absolute_sigma, **solver_kw)
    theta = np.exp(popt) if model == "randmod" else popt
    out = batch.solve_ode_batch(model, theta[None], init_cond, num_psites, time_points, want_sol=True, want_flat=True, **solver_kw)
    flat = out.flat[0].cpu().numpy()
    return dict(param_final=theta, popt=popt, pcov=pcov, lambda_reg=lam, weight_key=wkey, score=res.score, sol=out.sol[0].cpu().numpy(), fit=flat,
                error=float(np.sum(np.abs(flat - target) ** 2) / target.size), regularization_term=float(lam / P * np.sum(np.square(theta))))
