"""Batched multistart least squares: the numerical core of ``paramest.normest._curve_fit_multistart`` (normest.py:167-326).

The reference runs 24-48 independent ``scipy.optimize.curve_fit`` (TRF, ``x_scale='jac'``) calls; each TRF iteration costs 1 + P
``solve_ode`` calls for a 2-point finite-difference Jacobian, one at a time.  Here ALL starts advance in lockstep: every
iteration is ONE launch of ``n_active * P`` perturbed replicas; residuals, Jacobian columns and the normal equations J^T J / J^T r
are formed on the GPU from the ``flat`` vectors the kernel wrote, and only P x P + P doubles per row return to the host, where the small
bounded-LM algebra (P <= 64) runs batched.

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
    JTJ: np.ndarray             # [R, P, P] J^T J of the weighted residuals at the last Jacobian evaluation (what pcov needs)
    n_iter: int
    n_solves: int
    n_launches: int = 0         # solve launches (Jacobian batches + trial batches)


def fit_rows_batch(model: str, num_psites: int, time_points, P0, init_cond, target, sigma=None, lam=0.0, bounds=None,
                   max_iter: int = 100, ftol: float = 1e-10, xtol: float = 1e-10, device_algebra="auto", jacobian="auto", trial_levels="auto",
                   force_reg: Optional[bool] = None, lm_algebra="auto", **solver_kw) -> RowsFit:
    """R independent bounded least-squares problems in lockstep: row k fits ``[flat(p) ; lam_k / P * p**2]`` to ``[target_k ; 0]`` with
    weights ``sigma_k`` from the start point ``P0[k]``.  Rows may be the starts of one multistart fit, the (lambda, weight) grid of
    ``find_best_lambda``, bootstrap replicates, different proteins of the same size -- or any mix.

    P0 [R, P]; init_cond [S] or [R, S]; target [Nd] or [R, Nd]; sigma None, [Nr] or [R, Nr] with Nr = Nd (+ P when any lam > 0);
    lam scalar or [R]; bounds (lb, ub), each [P] or [R, P].

    ``jacobian="sens"``: the Jacobian of every active row comes from ONE launch of the forward-sensitivity kernel (csrc/pk_sens.hpp:
    n_active integrations, each carrying its P tangents; exact derivative of the discrete solution) -- ``"fd"``: SciPy's '2-point'
    forward differences, one launch of n_active * P perturbed replicas on the throughput kernels, which is what the reference's
    curve_fit does call by call.  ``"auto"`` (default) takes "sens" where a kernel exists (``batch.sens_available``: distmod / succmod
    n <= 62, randmod n <= 7) and the solver options are the default method's, else "fd".  Either way one more launch per damping round
    evaluates the trial points, ``trial_levels`` consecutive damping values (mu, 4 mu, 16 mu) per row and launch ("auto": three while at
    most 256 rows are pending, else one); the first acceptable one is taken, as a one-try-per-launch loop (``trial_levels=1``) would.

    ``device_algebra=True``: residuals, Jacobians and the normal equations J^T J, J^T r are formed on the GPU (torch ops on the
    `flat` tensors the kernel wrote); per iteration and row only P x P + P doubles come back and a P-vector of trial parameters goes up.
    ``False``: round 1's path -- every ``flat`` vector (n_active * P x Nd doubles per iteration) crosses PCIe and numpy does the algebra.
    ``"auto"`` (default) picks by the size of that transfer (> 2 MB: device).  Measured on MI355X (bench.py `lm_fit`): at 1 MB per Jacobian
    (48 starts, P = 20) the two paths tie (145-220 ms host, 183-186 ms device per 60-iteration fit: a dozen small torch launches and their
    synchronisations cost what the copy costs); at 109 MB (480 rows, P = 64: the lambda scan of a 30-site protein) the device path
    takes 0.39 s against 3.4 s.

    ``lm_algebra``: where the P x P damped normal equations of the trial steps are solved.  "host" (rounds 1-2): J^T J comes back (P x P per
    row and iteration) and batched numpy does the masking, the solves and the predicted reductions -- measured with cProfile on the 480-row
    lambda scan at P = 64 (tools/gpu_fit_profile2.py): 390 of the fit's 394 ms, against 10 ms of Jacobian kernels.  "device": J^T J stays in
    HBM; masking, the damped solves of all damping levels (``torch.linalg.solve_ex``: the vendor's batched LU), the projection on the box
    and the predicted reductions are device ops, and only vectors (gradient, diagonal, trial points' costs: O(P) per row) cross PCIe.
    "auto": device in every iteration whose ACTIVE rows x P^2 >= 2^19 (and the residual algebra is on the device), else host -- small
    problems, and the late iterations of big ones, are launch-bound: a dozen tiny device ops per round cost more than numpy on
    48 x 12 x 12 numbers (measured: 480 rows at P = 20, 60 iterations: 351 ms all-device, 261 ms all-host)."""
    import torch
    log_space = (model == "randmod")
    P0 = np.atleast_2d(np.asarray(P0, float))
    R, P = P0.shape
    target = np.asarray(target, float)
    tgt = np.broadcast_to(target, (R, target.shape[-1])) if target.ndim == 1 else target
    Nd = tgt.shape[1]
    lam = np.broadcast_to(np.asarray(lam, float), (R,)).copy()
    # ridge rows are part of the residual layout: a shard of a larger problem must use the layout of the WHOLE problem (force_reg),
    # or a rank whose rows all have lam == 0 would build Nd-wide residuals next to ranks building Nd + P
    use_reg = bool(np.any(lam > 0.0)) if force_reg is None else bool(force_reg)
    Nr = Nd + (P if use_reg else 0)
    if sigma is None:
        sig = np.ones((R, Nr))
    else:
        sig = np.asarray(sigma, float)
        sig = np.broadcast_to(sig, (R, sig.shape[-1])) if sig.ndim == 1 else sig
        if sig.shape[1] != Nr:
            raise ValueError(f"sigma must hold {Nr} entries")
    tfull = np.concatenate([tgt, np.zeros((R, P))], axis=1) if use_reg else np.array(tgt, dtype=float, copy=True)
    lb, ub = (np.broadcast_to(np.asarray(b, float), (R, P)) for b in bounds)
    y0 = np.asarray(init_cond, float)
    y0_rows = y0.ndim == 2
    n_solves = 0
    n_launches = 0
    dev = torch.device("cuda", batch.get_context().device)
    if device_algebra == "auto":
        device_algebra = R * P * Nd * 8 > (2 << 20)                   # bytes of `flat` one Jacobian evaluation would move over PCIe
    if lm_algebra not in ("auto", "host", "device"):
        raise ValueError("lm_algebra must be 'auto', 'host' or 'device'")
    # decided per iteration from the rows still active ("auto"): late iterations of a big fit are small, launch-bound problems again
    lm_dev_ok = (lm_algebra == "device") or (lm_algebra == "auto" and bool(device_algebra) and R * P * P >= (1 << 19))
    lm_dev = lm_dev_ok
    # per-fit constants go to HBM once (a host array handed to a launch is uploaded by that launch: 35 us each, several per iteration)
    t_d = torch.as_tensor(tfull, device=dev); isig_d = torch.as_tensor(1.0 / sig, device=dev); lam_d = torch.as_tensor(lam / P, device=dev)
    y0_d = torch.as_tensor(np.array(y0, dtype=float, copy=True), device=dev)
    tp_d = torch.as_tensor(np.ascontiguousarray(np.atleast_1d(np.asarray(time_points, dtype=float))), device=dev)

    def residuals_dev(Pm_d, rows_d):
        """Pm_d [m, P] (GPU) for the problems rows_d [m] (GPU index tensor) -> weighted residuals [m, Nr] on the GPU (one launch)."""
        nonlocal n_solves, n_launches
        theta = torch.exp(Pm_d) if log_space else Pm_d
        flat = batch.solve_ode_batch(model, theta, y0_d[rows_d] if y0_rows else y0_d, num_psites, tp_d, want_sol=False, want_flat=True, **solver_kw).flat
        n_solves += Pm_d.shape[0]; n_launches += 1
        f = torch.cat([flat, lam_d[rows_d, None] * Pm_d * Pm_d], dim=1) if use_reg else flat
        rr = (f - t_d[rows_d]) * isig_d[rows_d]
        return torch.where(torch.isfinite(rr), rr, torch.full_like(rr, 1e6))            # failed solves are very bad, not fatal

    # the sensitivity kernel integrates with the default method only and takes the tolerance / clipping options of the plain solve;
    # any other explicit solver option (another method, a linear solver, the stage form) keeps the differenced Jacobian
    sens_opts = ("rtol", "atol", "h0", "max_steps", "clip_nonneg", "normalize")
    sens_kw = {k: v for k, v in solver_kw.items() if k in sens_opts}

    def _default_for_sens(k, v):
        return v is None or k in sens_opts or k == "kernel" or (k == "method" and v in ("lrp12", 5))

    sens_ok = batch.sens_available(model, num_psites) and all(_default_for_sens(k, v) for k, v in solver_kw.items())
    if jacobian == "auto":
        jacobian = "sens" if sens_ok else "fd"
    if jacobian not in ("sens", "fd"):
        raise ValueError("jacobian must be 'auto', 'sens' or 'fd'")
    if jacobian == "sens" and not sens_ok:
        raise ValueError("jacobian='sens': no sensitivity kernel for this model size / these solver options (batch.sens_available)")

    def jacobian_sens(Pm, rows):
        """d (weighted residuals) / d p for the problems `rows` at Pm [m, P] from one sensitivity launch -> [m, Nr, P] (GPU tensor)."""
        nonlocal n_solves, n_launches
        Pm_d = torch.as_tensor(Pm, device=dev)
        theta = torch.exp(Pm_d) if log_space else Pm_d
        rows_d = torch.as_tensor(rows, device=dev)
        res = batch.solve_ode_sens_batch(model, theta, y0_d[rows_d] if y0_rows else y0_d, num_psites, tp_d, **sens_kw)
        n_solves += Pm_d.shape[0]; n_launches += 1
        D = res.dflat * theta[:, None, :] if log_space else res.dflat                   # chain rule of theta = exp(p)
        if use_reg:
            D = torch.cat([D, torch.diag_embed(2.0 * lam_d[rows_d, None] * Pm_d)], dim=1)
        D = D * isig_d[rows_d][:, :, None]
        return torch.where(torch.isfinite(D), D, torch.zeros_like(D))                   # a failed solve contributes no direction

    def residuals(Pm, rows):
        """Host path: Pm [m, P] for the problems `rows` [m] -> weighted residuals [m, Nr]  (one launch, flat over PCIe)."""
        nonlocal n_solves, n_launches
        theta = np.exp(Pm) if log_space else Pm
        flat = batch.solve_ode_batch(model, theta, y0_d[torch.as_tensor(rows, device=dev)] if y0_rows else y0_d, num_psites, tp_d, want_sol=False, want_flat=True,
                                     **solver_kw).flat.cpu().numpy()
        n_solves += Pm.shape[0]; n_launches += 1
        f = np.concatenate([flat, (lam[rows, None] / P) * Pm ** 2], axis=1) if use_reg else flat
        rr = (f - tfull[rows]) / sig[rows]
        return np.where(np.isfinite(rr), rr, 1e6)

    p = np.clip(P0, lb, ub)
    allr = np.arange(R)
    if device_algebra:
        r_d = residuals_dev(torch.as_tensor(p, device=dev), torch.as_tensor(allr, device=dev))
        cost = (0.5 * (r_d * r_d).sum(dim=1)).cpu().numpy()
    else:
        r = residuals(p, allr)
        cost = 0.5 * np.sum(r * r, axis=1)
    mu = np.full(R, 1e-3)
    active = np.ones(R, bool)
    JTJ = np.zeros((R, P, P))
    JTJ_d = torch.zeros((R, P, P), dtype=torch.float64, device=dev) if lm_dev_ok else None
    it = 0
    for it in range(1, max_iter + 1):
        idx = np.where(active)[0]
        if idx.size == 0:
            break
        lm_dev = lm_dev_ok and (lm_algebra == "device" or idx.size * P * P >= (1 << 19))
        # forward-difference Jacobian (SciPy's '2-point' rule: h = sqrt(eps) * max(1, |p|), flipped at the upper bound)
        h = np.sqrt(np.finfo(float).eps) * np.maximum(1.0, np.abs(p[idx]))
        h = np.where(p[idx] + h > ub[idx], -h, h)
        if jacobian == "fd":
            Pp = np.repeat(p[idx], P, axis=0)
            Pp[np.arange(idx.size * P), np.tile(np.arange(P), idx.size)] += h.reshape(-1)
        if jacobian == "sens":
            Jd = jacobian_sens(p[idx], idx)                                                                          # [k, Nr, P]
            if device_algebra:
                r_at = r_d[torch.as_tensor(idx, device=dev)]
            else:
                r_at = torch.as_tensor(r[idx], device=dev)
            # J^T [J | r] in one product, one copy back: [k, P, P + 1]  (lm_algebra = device: only the gradient and the diagonal come back)
            AG_d = torch.bmm(Jd.transpose(1, 2), torch.cat([Jd, r_at[:, :, None]], dim=2))
            if lm_dev_ok:
                A_d, g_d = AG_d[:, :, :P].contiguous(), AG_d[:, :, P].contiguous()
            if not lm_dev:
                AG = AG_d.cpu().numpy()
                A, g = np.ascontiguousarray(AG[:, :, :P]), np.ascontiguousarray(AG[:, :, P])
        elif device_algebra:
            idx_d = torch.as_tensor(idx, device=dev)
            rp = residuals_dev(torch.as_tensor(Pp, device=dev), idx_d.repeat_interleave(P)).reshape(idx.size, P, Nr)
            Jd = ((rp - r_d[idx_d][:, None, :]) / torch.as_tensor(h, device=dev)[:, :, None]).transpose(1, 2)        # [k, Nr, P]
            A_d = torch.bmm(Jd.transpose(1, 2), Jd)                                                                 # J^T J : [k, P, P]
            g_d = torch.bmm(Jd.transpose(1, 2), r_d[idx_d][:, :, None])[:, :, 0]                                    # J^T r : [k, P]
            if not lm_dev:
                A, g = A_d.cpu().numpy(), g_d.cpu().numpy()
        else:
            rp = residuals(Pp, np.repeat(idx, P)).reshape(idx.size, P, Nr)
            Ja = np.transpose((rp - r[idx][:, None, :]) / h[:, :, None], (0, 2, 1))
            g = np.einsum("knp,kn->kp", Ja, r[idx])
            A = np.einsum("knp,knq->kpq", Ja, Ja)
        if lm_dev_ok:
            if jacobian != "sens" and not device_algebra:                              # host residual algebra with device LM algebra: upload once
                A_d, g_d = torch.as_tensor(A, device=dev), torch.as_tensor(g, device=dev)
            JTJ_d[torch.as_tensor(idx, device=dev)] = A_d                              # what pcov needs stays in HBM until the end
        if lm_dev:
            g = g_d.cpu().numpy()
            diagA = torch.diagonal(A_d, dim1=1, dim2=2).cpu().numpy()
            p_idx_d, lb_idx_d, ub_idx_d = (torch.as_tensor(np.ascontiguousarray(x[idx]), device=dev) for x in (p, lb, ub))
        else:
            JTJ[idx] = A
            diagA = np.einsum("kpp->kp", A)
        free = ~(((p[idx] <= lb[idx]) & (g > 0)) | ((p[idx] >= ub[idx]) & (g < 0)))
        gfree = np.where(free, g, 0.0)
        done = (~free.any(axis=1)) | (np.linalg.norm(gfree, axis=1) < 1e-14 * np.maximum(1.0, cost[idx]))
        active[idx[done]] = False
        DD = np.maximum(np.sqrt(diagA), 1e-12)                                         # Marquardt scaling (the reference's x_scale='jac')
        pend = np.where(~done)[0]                                                      # positions inside idx
        # Levenberg-Marquardt trial steps, still in lockstep.  The sequential rule -- try mu; if the step is rejected, try 4 mu, 16 mu, ...
        # (up to 12 tries) -- is kept, but `trial_levels` consecutive damping values of every pending row are evaluated in ONE launch
        # and the first acceptable one in that order is taken: the same accepted steps as one try per launch, a third of the launches
        # (these batches are small: the fits are bound by launch latency and synchronisation, not by the solves)
        tries = 0
        while tries < 12:
            if pend.size == 0:
                break
            # "auto": small rounds are latency-bound (three levels per launch); beyond ~256 pending rows the wasted solves cost more than
            # the launches saved (measured: 480 rows of distmod n = 8, 65 ms with one level against 74 ms with three)
            K = (3 if pend.size <= 256 else 1) if trial_levels == "auto" else int(trial_levels)
            K = min(K, 12 - tries)
            tries += K
            rows = idx[pend]
            m = pend.size
            fr = free[pend]
            if lm_dev:
                # the damped normal equations of all K levels, the projection on the box and the predicted reductions on the device
                pend_d = torch.as_tensor(pend, device=dev)
                fr_d = torch.as_tensor(fr, device=dev)
                Ap_d = A_d[pend_d]
                Af0_d = Ap_d * (fr_d[:, :, None] & fr_d[:, None, :])
                rhs_d = torch.as_tensor(-gfree[pend], device=dev)
                dg0_d = torch.as_tensor(np.where(fr, mu[rows, None] * DD[pend] ** 2, 0.0), device=dev)
                one_fixed = (~fr_d).to(torch.float64)                                  # fixed variables: identity row, zero step
                pr_d, lbr_d, ubr_d = p_idx_d[pend_d], lb_idx_d[pend_d], ub_idx_d[pend_d]
                tl = []
                for lv in range(K):
                    Af = Af0_d.clone()
                    Af.diagonal(dim1=1, dim2=2).add_(dg0_d * (4.0 ** lv) + one_fixed)
                    step_d, _info = torch.linalg.solve_ex(Af, rhs_d[:, :, None])
                    step_d = step_d[:, :, 0]
                    step_d = torch.where(torch.isfinite(step_d), step_d, torch.zeros_like(step_d))
                    tl.append(torch.minimum(torch.maximum(pr_d + step_d, lbr_d), ubr_d))
                trials_d = torch.stack(tl)                                             # [K, m, P]
                dp_d = trials_d - pr_d[None]
                Adp = torch.matmul(Ap_d[None], dp_d[..., None])[..., 0]
                pred_dv = -((g_d[pend_d][None] * dp_d).sum(dim=2) + 0.5 * (dp_d * Adp).sum(dim=2))
                rows_d = torch.as_tensor(rows, device=dev)
                if device_algebra:
                    rn_d = residuals_dev(trials_d.reshape(K * m, P), rows_d.repeat(K)).reshape(K, m, Nr)
                    cn = (0.5 * (rn_d * rn_d).sum(dim=2)).cpu().numpy()
                trials = trials_d.cpu().numpy()
                if not device_algebra:
                    rn = residuals(trials.reshape(K * m, P), np.tile(rows, K)).reshape(K, m, Nr)
                    cn = 0.5 * np.sum(rn * rn, axis=2)
                dp = trials - p[rows][None]
                pred = pred_dv.cpu().numpy()
            trials = trials if lm_dev else np.empty((K, m, P))
            if not lm_dev:
                Af0 = A[pend] * (fr[:, :, None] & fr[:, None, :])
                rhs = -gfree[pend]
            for lv in range(0 if lm_dev else K):
                Af = Af0.copy()
                dg = np.where(fr, (mu[rows, None] * 4.0 ** lv) * DD[pend] ** 2, 1.0)   # fixed variables: identity row, zero step
                Af[:, np.arange(P), np.arange(P)] += dg
                try:
                    step = np.linalg.solve(Af, rhs[:, :, None])[:, :, 0]
                except np.linalg.LinAlgError:
                    step = np.zeros_like(rhs)
                    for k_ in range(m):
                        try:
                            step[k_] = np.linalg.solve(Af[k_], rhs[k_])
                        except np.linalg.LinAlgError:
                            pass
                step = np.where(np.isfinite(step), step, 0.0)
                trials[lv] = np.clip(p[rows] + step, lb[rows], ub[rows])
            if not lm_dev:
                if device_algebra:
                    rows_d = torch.as_tensor(rows, device=dev)
                    rn_d = residuals_dev(torch.as_tensor(trials.reshape(K * m, P), device=dev), rows_d.repeat(K)).reshape(K, m, Nr)
                    cn = (0.5 * (rn_d * rn_d).sum(dim=2)).cpu().numpy()                # [K, m]
                else:
                    rn = residuals(trials.reshape(K * m, P), np.tile(rows, K)).reshape(K, m, Nr)
                    cn = 0.5 * np.sum(rn * rn, axis=2)
                dp = trials - p[rows][None]
                Ap = A[pend]
                pred = -(np.einsum("kp,lkp->lk", g[pend], dp) + 0.5 * np.einsum("lkp,lkp->lk", dp, np.matmul(Ap[None], dp[..., None])[..., 0]))
            rho = np.where(pred > 0, (cost[rows][None] - cn) / np.where(pred > 0, pred, 1.0), -1.0)
            okl = (cn < cost[rows][None]) & (rho > 1e-4)                                # [K, m]
            ok = okl.any(axis=0)
            lvl = np.argmax(okl, axis=0)                                               # first acceptable damping level of each row
            sel = (lvl, np.arange(m))
            trial, cns, rhos, dps = trials[sel], cn[sel], rho[sel], dp[sel]
            dx = np.linalg.norm(dps, axis=1); dc = cost[rows] - cns
            conv = ok & ((dc <= ftol * np.maximum(cns, 1e-300)) | (dx <= xtol * (xtol + np.linalg.norm(trial, axis=1))))
            acc = rows[ok]
            p[acc], cost[acc] = trial[ok], cns[ok]
            if device_algebra:
                if acc.size:
                    ok_d = torch.as_tensor(ok, device=dev)
                    r_d[rows_d[ok_d]] = rn_d[torch.as_tensor(lvl, device=dev), torch.arange(m, device=dev)][ok_d]
            else:
                r[acc] = rn[sel][ok]
            mu[acc] = np.maximum(mu[acc] * 4.0 ** lvl[ok] * np.where(rhos[ok] > 0.75, 1.0 / 3.0, 1.0), 1e-12)
            active[rows[conv]] = False
            mu[rows[~ok]] *= 4.0 ** K
            pend = pend[~ok]
        active[idx[pend]] = False          # no acceptable step within the damping budget: converged / stalled
    r_out = r_d.cpu().numpy() if device_algebra else r
    if lm_dev_ok:
        JTJ = JTJ_d.cpu().numpy()
    return RowsFit(p=p, cost=cost, r=r_out, JTJ=JTJ, n_iter=it, n_solves=n_solves, n_launches=n_launches)


def fit_rows_sharded(model: str, num_psites: int, time_points, P0, init_cond, target, sigma=None, lam=0.0, bounds=None, **kw) -> RowsFit:
    """``fit_rows_batch`` with the R problems dealt round-robin over the ranks of an initialised ``torch.distributed`` group (one process
    per GPU): every rank fits its rows, then ONE all-gather of the per-row results [p | cost | J^T J] (P + 1 + P^2 doubles per row) gives
    every rank the complete ``RowsFit``.  Rows never interact, so the result equals the single-GPU fit row for row.  Without a process
    group (or at world size 1) it is ``fit_rows_batch``.  ``n_iter`` / ``n_solves`` / ``n_launches`` of the result are RANK-LOCAL counters
    (the work this rank did), not totals."""
    import torch
    from ..distributed import interleaved_rows, all_gather_interleaved, _world, _control_device
    rank, world = _world()
    P0 = np.atleast_2d(np.asarray(P0, float))
    R, P = P0.shape
    if world == 1:
        return fit_rows_batch(model, num_psites, time_points, P0, init_cond, target, sigma=sigma, lam=lam, bounds=bounds, **kw)
    # rows dealt round-robin (rank r: rows r, r + W, ...): rows of one lambda / one weighting sit next to each other and converge alike
    mine = interleaved_rows(R, rank, world).numpy()
    kw = dict(kw, force_reg=bool(np.any(np.asarray(lam, float) > 0.0)))            # the residual layout of the whole problem on every rank
    rows = lambda a, nd: (np.asarray(a, float)[mine] if np.asarray(a).ndim == nd else a)       # per-row arguments are sliced, shared ones passed on
    dev = _control_device()                                        # HBM for RCCL, host memory for the gloo tests
    if mine.size:
        fit = fit_rows_batch(model, num_psites, time_points, P0[mine], rows(init_cond, 2), rows(target, 2), sigma=(None if sigma is None else rows(sigma, 2)),
                             lam=rows(lam, 1), bounds=tuple(rows(b, 2) for b in bounds), **kw)
        packed = np.concatenate([fit.p, fit.cost[:, None], fit.JTJ.reshape(mine.size, P * P), fit.r], axis=1)
        meta = (fit.n_iter, fit.n_solves, fit.n_launches)
    else:
        packed = np.zeros((0, 0)); meta = (0, 0, 0)
    # every rank must offer the same row width: the residual length is known from the shapes alone
    Nd = np.asarray(target).shape[-1]
    Nr = Nd + (P if np.any(np.asarray(lam, float) > 0.0) else 0)
    if packed.shape[0] == 0:
        packed = np.zeros((0, P + 1 + P * P + Nr))
    full = all_gather_interleaved(torch.as_tensor(packed, device=dev), R).cpu().numpy()
    return RowsFit(p=full[:, :P], cost=full[:, P], JTJ=full[:, P + 1:P + 1 + P * P].reshape(R, P, P), r=full[:, P + 1 + P * P:], n_iter=meta[0], n_solves=meta[1],
                   n_launches=meta[2])


def _pcov(JTJ_b, cost_b, absolute_sigma, Nr):
    """(J^T J)^-1 [* s^2] as scipy.optimize.curve_fit reports it."""
    P = JTJ_b.shape[0]
    try:
        pcov = np.linalg.inv(JTJ_b)
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
    return FitResult(popt=fit.p[best].copy(), pcov=_pcov(fit.JTJ[best], fit.cost[best], absolute_sigma, fit.r.shape[1]), score=float(scores[best]), cost=fit.cost,
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
    covs = [c for c in (_pcov(fit.JTJ[k], fit.cost[k], absolute_sigma, fit.r.shape[1]) for k in range(bootstraps)) if c is not None]
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
