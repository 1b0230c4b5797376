"""CPU restatement (numpy + SciPy) of the reference's per-protein kinetic models.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference
file:line it follows.  The integrator arithmetic of the reference lives in a third-party
dependency, ``scipy.integrate.odeint`` (ODEPACK LSODA; reference pins scipy 1.15.2 in
poetry.lock:1838, this image has 1.15.3) -- the oracle calls the very same routine, with
the same defaults, on a numpy restatement of the reference's right-hand sides.

Model ids used everywhere in this repo: 0 = distmod, 1 = succmod, 2 = randmod.
State layout (all models): y = [R, P, X_1 .. X_m]; m = n_sites (dist/succ) or 2**n - 1 (rand).
Parameter layout: theta = [A, B, C, D, S_1..S_n, D_1..D_m]   (reference unpack_params).
"""
from __future__ import annotations

import math
import numpy as np
from scipy.integrate import odeint
from scipy.linalg import expm

DIST, SUCC, RAND = 0, 1, 2
MODEL_NAMES = {DIST: "distmod", SUCC: "succmod", RAND: "randmod"}
MODEL_IDS = {v: k for k, v in MODEL_NAMES.items()}

#: reference time grid, config/constants.py:56-62
TIME_POINTS = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0,
                        120.0, 240.0, 480.0, 960.0])


def n_states(model: int, n_sites: int) -> int:
    return 2 + (n_sites if model != RAND else (1 << n_sites) - 1)


def n_params(model: int, n_sites: int) -> int:
    return 4 + n_sites + (n_sites if model != RAND else (1 << n_sites) - 1)


def unpack_params(model: int, params, n_sites: int):
    """distmod.py:68-91, succmod.py:94-112, randmod.py:88-119."""
    p = np.asarray(params, dtype=float)
    m = n_sites if model != RAND else (1 << n_sites) - 1
    return p[0], p[1], p[2], p[3], p[4:4 + n_sites].copy(), p[4 + n_sites:4 + n_sites + m].copy()


# --------------------------------------------------------------------------- RHS
def rhs_dist(y, t, A, B, C, D, S, Dr):
    """models/distmod.py:7-65 (ode_core)."""
    n = S.shape[0]
    R, P = y[0], y[1]
    dy = np.empty_like(y)
    dy[0] = A - B * R
    sum_S = 0.0
    for i in range(n):
        sum_S += S[i]
    sum_P = 0.0
    for i in range(n):
        sum_P += y[2 + i]
    dy[1] = C * R - (D + sum_S) * P + sum_P
    for i in range(n):
        dy[2 + i] = S[i] * P - (1.0 + Dr[i]) * y[2 + i]
    return dy


def rhs_succ(y, t, A, B, C, D, S, Dr):
    """models/succmod.py:9-90 (ode_core), incl. the n == 1 branch (:59-63)."""
    n = S.shape[0]
    R, P = y[0], y[1]
    dy = np.empty_like(y)
    dy[0] = A - B * R
    dP = C * R - D * P
    if n > 0:
        dP -= S[0] * P
        dP += y[2]
    dy[1] = dP
    for i in range(n):
        if n == 1:
            dy[2] = S[0] * P - (1 + Dr[0]) * y[2]
        elif i == 0:
            dy[2] = S[0] * P - (1 + S[1] + Dr[0]) * y[2] + y[3]
        elif i < n - 1:
            dy[2 + i] = S[i] * y[1 + i] - (1 + S[i + 1] + Dr[i]) * y[2 + i] + y[3 + i]
        else:
            dy[2 + i] = S[i] * y[1 + i] - (1 + Dr[i]) * y[2 + i]
    return dy


def rand_tables(n: int):
    """models/randmod.py:9-85 (_precompute_indices); int64 tables, 1-based state masks."""
    m = (1 << n) - 1
    mono_idx = np.array([(1 << j) - 1 for j in range(n)], dtype=np.int64)
    forward = -np.ones((m, n), dtype=np.int64)
    drop = -np.ones((m, n), dtype=np.int64)
    fcounts = np.zeros(m, dtype=np.int64)
    dcounts = np.zeros(m, dtype=np.int64)
    for state in range(1, m + 1):
        fi = di = 0
        for j in range(n):
            if not (state & (1 << j)):
                forward[state - 1, fi] = state | (1 << j)
                fcounts[state - 1] += 1
                fi += 1
            else:
                drop[state - 1, di] = state & ~(1 << j)
                dcounts[state - 1] += 1
                di += 1
    return mono_idx, forward, drop, fcounts, dcounts


def rhs_rand(y, t, A, B, C, D, n, S, Ddeg, tables=None):
    """models/randmod.py:122-247 (ode_system), verbatim control flow.

    Quirk kept on purpose (randmod.py:201): the rate of the forward transition
    state -> tgt uses S[j] with j = log2(lowest set bit of the TARGET mask)."""
    mono_idx, forward, drop, fcounts, dcounts = tables if tables is not None else rand_tables(n)
    m = (1 << n) - 1
    R, P = y[0], y[1]
    dR = A - B * R
    dP = C * R - D * P
    dX = np.zeros(m)
    for k in range(n):
        rate = S[k] * P
        dX[mono_idx[k]] += rate
        dP -= rate
    for state in range(1, m + 1):
        xi = y[1 + state]
        base = state - 1
        for k in range(fcounts[base]):
            tgt = forward[base, k] - 1
            j = int(math.log2((tgt + 1) & -(tgt + 1)))
            rate = S[j] * xi
            dX[tgt] += rate
            dX[base] -= rate
        for k in range(dcounts[base]):
            lower = drop[base, k]
            rate = xi
            if lower == 0:
                dP += rate
            else:
                dX[lower - 1] += rate
            dX[base] -= rate
        dX[base] -= Ddeg[base] * xi
    out = np.empty(2 + m)
    out[0] = dR
    out[1] = dP
    out[2:] = dX
    return out


def rhs(model: int, y, t, params, n_sites: int):
    A, B, C, D, S, Dr = unpack_params(model, params, n_sites)
    y = np.asarray(y, dtype=float)
    if model == DIST:
        return rhs_dist(y, t, A, B, C, D, S, Dr)
    if model == SUCC:
        return rhs_succ(y, t, A, B, C, D, S, Dr)
    return rhs_rand(y, t, A, B, C, D, n_sites, S, Dr)


# ---------------------------------------------------------------- LTI form M y + b
def lti_matrix(model: int, params, n_sites: int):
    """Analytic Jacobian M (constant in y, t) and forcing b with dy/dt = M y + b.

    Derived from the three RHS above; the reference never
    forms it (LSODA finite-differences).  Built column-by-column from the RHS itself so it
    cannot drift from it: M[:, j] = rhs(e_j) - rhs(0), b = rhs(0) (exact: the RHS is affine)."""
    S_ = n_states(model, n_sites)
    b = rhs(model, np.zeros(S_), 0.0, params, n_sites)
    M = np.empty((S_, S_))
    for j in range(S_):
        e = np.zeros(S_)
        e[j] = 1.0
        M[:, j] = rhs(model, e, 0.0, params, n_sites) - b
    return M, b


def steady_state(model: int, params, n_sites: int):
    """y* with M y* + b = 0.  steady/initdist.py:9-50, initsucc.py:9-55 and initrand.py:9-77 pose exactly these equations (every rate
    fixed to 1) as an SLSQP feasibility problem; for the affine models the solution is this linear solve."""
    M, b = lti_matrix(model, params, n_sites)
    return np.linalg.solve(M, -b)


def initial_condition(kind: str, n_sites: int):
    """The list steady.<kind>.initial_condition(n) returns: unit rates; initsucc uses the DISTRIBUTIVE equations (initsucc.py:39-41);
    initrand orders the phospho states by subset size then lexicographically (initrand.py:24-28) instead of by bit mask."""
    from itertools import combinations
    model = {"initdist": 0, "initsucc": 0, "initrand": 2}[kind]
    y = steady_state(model, np.ones(n_params(model, n_sites)), n_sites)
    if kind != "initrand":
        return y
    out = [y[0], y[1]]
    for k in range(1, n_sites + 1):
        for comb in combinations(range(1, n_sites + 1), k):
            out.append(y[1 + sum(1 << (s - 1) for s in comb)])
    return np.array(out)


def jacobian_analytic(model: int, params, n_sites: int):
    """Closed-form Jacobian, written independently of `lti_matrix` (row-major J[i, j] = d f_i / d y_j)."""
    A, B, C, D, S, Dr = unpack_params(model, params, n_sites)
    n = n_sites
    S_ = n_states(model, n)
    J = np.zeros((S_, S_))
    J[0, 0] = -B
    J[1, 0] = C
    if model == DIST:
        J[1, 1] = -(D + S.sum())
        for i in range(n):
            J[1, 2 + i] = 1.0
            J[2 + i, 1] = S[i]
            J[2 + i, 2 + i] = -(1.0 + Dr[i])
    elif model == SUCC:
        J[1, 1] = -D
        if n > 0:
            J[1, 1] -= S[0]
            J[1, 2] = 1.0
        for i in range(n):
            J[2 + i, 1 + i] = S[i]
            J[2 + i, 2 + i] = -(1.0 + Dr[i] + (S[i + 1] if i < n - 1 else 0.0))
            if i < n - 1:
                J[2 + i, 3 + i] = 1.0
    else:
        J[1, 1] = -(D + S.sum())
        m = (1 << n) - 1
        for mask in range(1, m + 1):
            row = 1 + mask
            lsb = (mask & -mask).bit_length() - 1
            out_rate = 0.0
            for j in range(n):
                bit = 1 << j
                if mask & bit:
                    src = mask ^ bit
                    # inflow into `mask` from src (or from P): coefficient S[lsb(mask)] (quirk)
                    J[row, 1 + src if src else 1] += S[lsb]
                    # this state dephosphorylates bit j at unit rate
                    lower = mask ^ bit
                    J[1 + lower if lower else 1, row] += 1.0
                    out_rate += 1.0
                else:
                    tgt = mask | bit
                    out_rate += S[((tgt & -tgt).bit_length() - 1)]
            J[row, row] -= out_rate + Dr[mask - 1]
    return J


# ----------------------------------------------------------------------- solve
def _odeint(model, params, y0, n_sites, t, **kw):
    A, B, C, D, S, Dr = unpack_params(model, params, n_sites)
    if model == DIST:
        return odeint(rhs_dist, y0, t, args=(A, B, C, D, S, Dr), **kw)
    if model == SUCC:
        return odeint(rhs_succ, y0, t, args=(A, B, C, D, S, Dr), **kw)
    tabs = rand_tables(n_sites)
    return odeint(rhs_rand, y0, t, args=(A, B, C, D, n_sites, S, Dr, tabs), **kw)


def flatten_observables(model: int, sol: np.ndarray, n_sites: int) -> np.ndarray:
    """distmod.py:125-134, succmod.py:143-152, randmod.py:284-305: [R(t5..), P(t0..), sites site-major].
    randmod keeps only the first n_sites phospho columns (randmod.py:298-299)."""
    R_f = sol[5:, 0]
    P_f = sol[:, 1]
    if model == RAND:
        X_f = sol[:, 2:2 + n_sites].T
    else:
        X_f = sol[:, 2:].T
    return np.concatenate((R_f.ravel(), P_f.ravel(), X_f.ravel()))


def solve_ode(model: int, params, init_cond, n_sites: int, t, normalize: bool = False, **odeint_kw):
    """models/{distmod.py:93-134, succmod.py:114-152, randmod.py:249-305}: odeint at SciPy defaults
    (rtol = atol = 1.49012e-8, no Dfun) -> clip >= 0 -> optional / y0 -> flat."""
    t = np.atleast_1d(np.asarray(t, dtype=float))
    y0 = np.asarray(init_cond, dtype=float)
    sol = np.clip(np.asarray(_odeint(model, params, y0, n_sites, t, **odeint_kw)), 0, None)
    if normalize:
        sol *= (1.0 / y0)[None, :]
    return sol, flatten_observables(model, sol, n_sites)


def solve_tight(model: int, params, init_cond, n_sites: int, t):
    """Same SciPy odeint, same RHS, rtol = atol = 1e-13, mxstep = 500000 (SURVEY.md section 7 'y_ref_tight').
    NOT clipped."""
    t = np.atleast_1d(np.asarray(t, dtype=float))
    return np.asarray(_odeint(model, params, np.asarray(init_cond, float), n_sites, t,
                              rtol=1e-13, atol=1e-13, mxstep=500000))


def solve_exact_lti(model: int, params, init_cond, n_sites: int, t):
    """Third, integrator-free truth for the (linear time-invariant) per-protein models:
    z = [y; 1], dz/dt = [[M, b], [0, 0]] z, stepped interval by interval with scipy.linalg.expm."""
    t = np.atleast_1d(np.asarray(t, dtype=float))
    M, b = lti_matrix(model, params, n_sites)
    S_ = M.shape[0]
    Aug = np.zeros((S_ + 1, S_ + 1))
    Aug[:S_, :S_] = M
    Aug[:S_, S_] = b
    z = np.concatenate((np.asarray(init_cond, float), [1.0]))
    out = np.empty((t.size, S_))
    out[0] = z[:S_]
    for k in range(1, t.size):
        z = expm(Aug * (t[k] - t[k - 1])) @ z
        out[k] = z[:S_]
    return out


# --------------------------------------------------------------------- reductions
def compute_Y(solution: np.ndarray, n_sites: int, metric: str = "total_signal") -> float:
    """sensitivity/analysis.py:90-176 (_compute_Y) with Y_METRIC as an argument -- loop for loop, in the reference's summation order
    (pinned bit for bit by tests/golden/pins_protein.npz)."""
    sol = np.asarray(solution, dtype=float)
    n_t = sol.shape[0]
    length = 2 * n_t + n_t * n_sites
    sum_m = 0.0; sum_p = 0.0; sum_s = 0.0
    for t in range(n_t):
        sum_m += sol[t, 0]
        sum_p += sol[t, 1]
        for s in range(n_sites):
            sum_s += sol[t, 2 + s]
    if metric == "total_signal":
        return float(sum_m + sum_p + sum_s)
    if metric == "mean_activity":
        return float((sum_m + sum_p + sum_s) / length)
    if metric == "variance":
        mean = (sum_m + sum_p + sum_s) / length
        acc = 0.0
        for t in range(n_t):
            acc += (sol[t, 0] - mean) ** 2
            acc += (sol[t, 1] - mean) ** 2
            for s in range(n_sites):
                acc += (sol[t, 2 + s] - mean) ** 2
        return float(acc / length)
    if metric == "dynamics":
        acc = 0.0
        for col in range(2 + n_sites):                   # mRNA chain, protein chain, then the site chains in sequence
            prev = sol[0, col]
            for t in range(1, n_t):
                cur = sol[t, col]
                acc += (cur - prev) ** 2
                prev = cur
        return float(acc)
    if metric == "l2_norm":
        acc = 0.0
        for t in range(n_t):
            acc += sol[t, 0] ** 2
        for t in range(n_t):
            acc += sol[t, 1] ** 2
        for t in range(n_t):
            for s in range(n_sites):
                acc += sol[t, 2 + s] ** 2
        return float(math.sqrt(acc))
    raise ValueError("Unknown Y_METRIC")


METRICS = ("total_signal", "mean_activity", "variance", "dynamics", "l2_norm")


def score_fit(params, target, prediction, alpha=1.0, beta=1.0, gamma=1.0, delta=1.0, mu=1.0):
    """config/config.py:176-226."""
    params = np.asarray(params, float)
    target = np.asarray(target, float)
    prediction = np.asarray(prediction, float)
    residual = np.abs(target - prediction) / target.size
    mse = np.sum(residual ** 2)
    rmse = np.sqrt(np.mean(residual ** 2))
    mae = np.mean(residual)
    variance = np.var(residual)
    l2 = np.linalg.norm(params, ord=2) / len(params)
    return delta * mse + alpha * rmse + beta * mae + gamma * variance + mu * l2


def define_sensitivity_problem(model: int, n_sites: int, values, perturbation: float = 0.5):
    """sensitivity/analysis.py:38-87 (define_sensitivity_problem_ds / _rand): names A..D, S1..Sn, D1..Dm and compute_bound per value."""
    names = ["A", "B", "C", "D"] + [f"S{i + 1}" for i in range(n_sites)]
    if model == RAND:                    # config/helpers/__init__.py:5-22: one D per non-empty site subset, by size then lexicographic
        from itertools import combinations
        for i in range(1, n_sites + 1):
            for combo in combinations(range(1, n_sites + 1), i):
                names.append("D" + "".join(map(str, combo)))
    else:
        names += [f"D{i + 1}" for i in range(n_sites)]
    assert len(values) == len(names), "Length mismatch with values"
    return {"num_vars": len(names), "names": names, "bounds": [compute_bound(v, perturbation) for v in values]}


def multistart_start_list(gene: str, base_p0, lb, ub, n_starts: int = 24, jitter_frac: float = 0.10, seed: int = 42):
    """paramest/normest.py:217-265: the candidate start points of _curve_fit_multistart, draw for draw."""
    lb = np.asarray(lb, float); ub = np.asarray(ub, float)
    rng = np.random.default_rng(int(seed + (sum(ord(c) for c in str(gene)) % 1000003)))
    base = np.clip(np.asarray(base_p0, float).copy(), lb, ub)
    out = [base]
    span = ub - lb
    span[span <= 0] = 1.0
    for _ in range(max(0, n_starts // 3)):
        noise = rng.normal(0.0, 1.0, size=base.shape[0])
        out.append(np.clip(base + (jitter_frac * span) * noise, lb, ub))
    remaining = max(0, n_starts - len(out))
    if remaining > 0:
        U = np.empty((remaining, base.shape[0]))
        for j in range(base.shape[0]):
            u = (np.arange(remaining) + rng.random(remaining)) / float(remaining)
            rng.shuffle(u)
            U[:, j] = u
        out.extend(list(lb + U * (ub - lb)))
    return np.stack(out)


def compute_bound(value: float, perturbation: float = 0.5):
    """sensitivity/analysis.py:20-35."""
    if abs(value) < 1e-6:
        return [0.0, 0.1]
    return [max(0.0, value * (1 - perturbation)), value * (1 + perturbation)]


def band_error(y, y_ref, rtol=1e-6, atol=1e-8) -> float:
    """max |y - y_ref| / (atol + rtol |y_ref|)  -- the parity gate of BASELINE.json (pass <= 1)."""
    y = np.asarray(y); y_ref = np.asarray(y_ref)
    return float(np.max(np.abs(y - y_ref) / (atol + rtol * np.abs(y_ref))))


# --------------------------------------------------------------- CPU baseline helpers (bench.py cpu_baseline leg)
def rhs_dist_vec(y, t, A, B, C, D, S, Dr):
    """distmod.py:7-65 with the per-site loops as numpy vector ops (the reference runs them Numba-compiled; a pure-Python
    loop would overstate the CPU cost by ~10x).  Same arithmetic up to summation order."""
    dy = np.empty_like(y)
    P = y[1]
    dy[0] = A - B * y[0]
    dy[1] = C * y[0] - (D + S.sum()) * P + y[2:].sum()
    dy[2:] = S * P - (1.0 + Dr) * y[2:]
    return dy


def rhs_succ_vec(y, t, A, B, C, D, S, Dr):
    """succmod.py:9-90 vectorised (n >= 2)."""
    n = S.shape[0]
    dy = np.empty_like(y)
    dy[0] = A - B * y[0]
    dy[1] = C * y[0] - D * y[1] - S[0] * y[1] + y[2]
    x = y[2:]
    up = np.empty(n); up[:-1] = x[1:]; up[-1] = 0.0
    nxt = np.empty(n); nxt[:-1] = S[1:]; nxt[-1] = 0.0
    dy[2:] = S * y[1:1 + n] - (1.0 + nxt + Dr) * x + up
    return dy


def solve_ode_fast(model: int, params, init_cond, n_sites: int, t):
    """The reference call shape (odeint at SciPy defaults -> clip -> flat) with the vectorised RHS: the CPU baseline."""
    A, B, C, D, S, Dr = unpack_params(model, params, n_sites)
    y0 = np.asarray(init_cond, float)
    if model == DIST:
        sol = odeint(rhs_dist_vec, y0, t, args=(A, B, C, D, S, Dr))
    elif model == SUCC and n_sites >= 2:
        sol = odeint(rhs_succ_vec, y0, t, args=(A, B, C, D, S, Dr))
    else:
        sol = _odeint(model, params, y0, n_sites, t)
    sol = np.clip(np.asarray(sol), 0, None)
    return sol, flatten_observables(model, sol, n_sites)


def _baseline_worker(args):
    model, n_sites, thetas, y0, t = args
    import time as _t
    t0 = _t.perf_counter()
    acc = 0.0
    for th in thetas:
        sol, flat = solve_ode_fast(model, th, y0, n_sites, t)
        acc += float(sol[-1, 0])
    return len(thetas), _t.perf_counter() - t0, acc


def cpu_baseline(model: int, n_sites: int, thetas: np.ndarray, y0, t, workers: int):
    """Fan the sample out over `workers` processes, one chunk each (the reference's shape: one future per parameter vector
    in a ProcessPoolExecutor, sensitivity/analysis.py:241-243).  Returns (solves_per_second, wall_seconds)."""
    import time as _t
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    chunks = [c for c in np.array_split(thetas, workers) if len(c)]
    t0 = _t.perf_counter()
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("fork")) as ex:
        res = list(ex.map(_baseline_worker, [(model, n_sites, c, y0, t) for c in chunks]))
    wall = _t.perf_counter() - t0
    busy = max(r[1] for r in res)            # slowest worker's compute time: excludes pool start-up / import cost
    return sum(r[0] for r in res) / busy, wall
