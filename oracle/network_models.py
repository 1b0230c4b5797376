"""CPU restatement (numpy + SciPy) of the reference's global_model NETWORK right-hand sides and solver call.

TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  Pinned against tests/golden/network_m*.npz, which tools/make_golden_network.py
produced by running the reference's own classes (Index / System / rhs_odeint / fd_jacobian_odeint / simulate_odeint).

Kinetic topologies (global_model/config.py:59-61):  0 distributive, 1 sequential, 2 combinatorial, 4 saturating.
State layout (global_model/network.py:28-167): per protein i a block starting at offset_y[i]:
   models 0/1/4: [R_i, P_i, site_1 .. site_ns]          model 2: [R_i, state_0 (= unphosphorylated) .. state_{2^ns - 1}]
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
from scipy.integrate import odeint


@dataclass
class Network:
    """Static topology + inputs: everything of System.odeint_args() (network.py:443-526) that does not change per candidate."""
    model: int
    N: int
    n_K: int
    total_sites: int
    S: int
    offset_y: np.ndarray
    offset_s: np.ndarray
    n_sites: np.ndarray
    W_indptr: np.ndarray
    W_indices: np.ndarray
    W_data: np.ndarray
    TF_indptr: np.ndarray
    TF_indices: np.ndarray
    TF_data: np.ndarray
    tf_deg: np.ndarray
    driver_map: np.ndarray
    kin_grid: np.ndarray
    kin_Kmat: np.ndarray
    n_states: Optional[np.ndarray] = None

    @classmethod
    def from_npz(cls, g):
        kw = {k: (int(g[k]) if g[k].ndim == 0 else np.asarray(g[k])) for k in
              ("model", "N", "n_K", "total_sites", "S", "offset_y", "offset_s", "n_sites", "W_indptr", "W_indices", "W_data",
               "TF_indptr", "TF_indices", "TF_data", "tf_deg", "driver_map", "kin_grid", "kin_Kmat")}
        if "n_states" in g:
            kw["n_states"] = np.asarray(g["n_states"])
        return cls(**kw)


@dataclass
class Params:
    """One candidate's physical parameters (System.update, network.py:293-302)."""
    c_k: np.ndarray
    A_i: np.ndarray
    B_i: np.ndarray
    C_i: np.ndarray
    D_i: np.ndarray
    Dp_i: np.ndarray
    E_i: np.ndarray
    tf_scale: float

    @classmethod
    def from_npz(cls, g, k):
        return cls(*(np.asarray(g[n][k]) for n in ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")), float(g["tf_scale"][k]))


def time_bucket(t, grid):
    """global_model/utils.py:211-225 and jacspeedup.py:149-172 (kin_eval_step): column searchsorted(grid, t, 'right') - 1, clamped."""
    if t <= grid[0]:
        return 0
    if t >= grid[-1]:
        return grid.size - 1
    j = int(np.searchsorted(grid, t, side="right")) - 1
    return min(max(j, 0), grid.size - 1)


def calculate_synthesis_rate(Ai, tf_scale, u_raw):
    """global_model/models.py:28-65."""
    u = u_raw / (1.0 + abs(u_raw))
    if u >= 0.0:
        return Ai * (1.0 + (tf_scale * u) / (1.0 + u + 1e-6))
    return Ai / (1.0 + tf_scale * abs(u))


def csr_matvec(indptr, indices, data, x, n_rows):
    """jacspeedup.py:71-98."""
    out = np.zeros(n_rows)
    for i in range(n_rows):
        s = 0.0
        for p in range(indptr[i], indptr[i + 1]):
            s += data[p] * x[indices[p]]
        out[i] = s
    return out


def site_rates(net: Network, p: Params, t):
    """S_all = W (K(t) * c_k)  -- jacspeedup.py:198-204; for model 2 the same numbers come from S_cache[:, bucket]
    (build_S_cache_into, jacspeedup.py:117-145)."""
    jb = time_bucket(t, net.kin_grid)
    Kt = net.kin_Kmat[:, jb] * p.c_k
    if net.model == 2:
        out = np.zeros(net.total_sites)
        for i in range(net.total_sites):
            s = 0.0
            for q in range(net.W_indptr[i], net.W_indptr[i + 1]):
                k = net.W_indices[q]
                s += net.W_data[q] * (net.kin_Kmat[k, jb] * p.c_k[k])
            out[i] = s
        return out, Kt
    return csr_matvec(net.W_indptr, net.W_indices, net.W_data, Kt, net.total_sites), Kt


def tf_inputs(net: Network, p: Params, y, Kt):
    """P_vec (driven proteins read K(t) c_k; model 2 ignores driver_map, jacspeedup.py:319-327) -> TF CSR matvec -> / tf_deg ->
    first squash (models 0/1/2: jacspeedup.py:225-228; model 4 leaves it to the kernel, :371-373)."""
    P_vec = np.zeros(net.N)
    for i in range(net.N):
        st = net.offset_y[i]
        if net.model == 2:
            tot = 0.0
            for m in range(int(net.n_states[i])):
                tot += y[st + 1 + m]
            P_vec[i] = tot
        else:
            d = net.driver_map[i]
            if d >= 0:
                P_vec[i] = Kt[d]
            else:
                tot = y[st + 1]
                for j in range(int(net.n_sites[i])):
                    tot += y[st + 2 + j]
                P_vec[i] = tot
    TF_in = csr_matvec(net.TF_indptr, net.TF_indices, net.TF_data, P_vec, net.N)
    for i in range(net.N):
        val = TF_in[i] / net.tf_deg[i]
        TF_in[i] = val if net.model == 4 else val / (1.0 + abs(val))
    return TF_in


def rhs(net: Network, p: Params, y, t):
    """rhs_odeint (jacspeedup.py:392-394) for the four topologies: models.py:150-212 (0), :216-306 (1), :323-432 (2), :72-146 (4)."""
    y = np.asarray(y, float)
    dy = np.zeros_like(y)
    S_all, Kt = site_rates(net, p, t)
    TF_in = tf_inputs(net, p, y, Kt)
    for i in range(net.N):
        st = int(net.offset_y[i]); ss = int(net.offset_s[i]); ns = int(net.n_sites[i])
        R = y[st]
        Ai, Bi, Ci, Di, Ei = p.A_i[i], p.B_i[i], p.C_i[i], p.D_i[i], p.E_i[i]
        synth = calculate_synthesis_rate(Ai, p.tf_scale, TF_in[i])
        dy[st] = synth - Bi * R
        if net.model == 0:
            P = y[st + 1]
            if ns == 0:
                dy[st + 1] = Ci * R - Di * P
            else:
                sum_S = 0.0; sum_back = 0.0
                for j in range(ns):
                    s_rate = S_all[ss + j]; ps = y[st + 2 + j]
                    sum_S += s_rate; sum_back += Ei * ps
                    dy[st + 2 + j] = s_rate * P - (Ei + p.Dp_i[ss + j] + Di) * ps
                dy[st + 1] = Ci * R - (Di + sum_S) * P + sum_back
        elif net.model == 4:
            P = y[st + 1]
            trans = (Ci * R) / (1.0 + R)
            if ns == 0:
                dy[st + 1] = trans - Di * P
            else:
                sum_f = 0.0; sum_b = 0.0
                for j in range(ns):
                    ps = y[st + 2 + j]
                    fwd = (S_all[ss + j] * P) / (1.0 + P)
                    back = Ei * ps
                    sum_f += fwd; sum_b += back
                    dy[st + 2 + j] = fwd - (p.Dp_i[ss + j] + Di) * ps - back
                dy[st + 1] = trans - Di * P - sum_f + sum_b
        elif net.model == 1:
            P0 = y[st + 1]
            if ns == 0:
                dy[st + 1] = Ci * R - Di * P0
                continue
            base = st + 2
            k0 = S_all[ss]; P1 = y[base]
            dy[st + 1] = Ci * R - Di * P0 - k0 * P0 + Ei * P1
            if ns == 1:
                dy[base] = k0 * P0 - (Ei + p.Dp_i[ss] + Di) * P1
                continue
            k1 = S_all[ss + 1]; P2 = y[base + 1]
            dy[base] = k0 * P0 + Ei * P2 - (k1 + Ei + p.Dp_i[ss] + Di) * P1
            for j in range(1, ns - 1):
                ix = base + j
                dy[ix] = S_all[ss + j] * y[ix - 1] + Ei * y[ix + 1] - (S_all[ss + j + 1] + Ei + p.Dp_i[ss + j] + Di) * y[ix]
            il = base + ns - 1
            dy[il] = S_all[ss + ns - 1] * y[il - 1] - (Ei + p.Dp_i[ss + ns - 1] + Di) * y[il]
        else:  # model 2
            if ns == 0:
                dy[st + 1] = Ci * R - Di * y[st + 1]
                continue
            base = st + 1
            nst = int(net.n_states[i])
            dy[base] += Ci * R
            dy[base] += -Di * y[base]
            for m in range(1, nst):
                Pm = y[base + m]
                if Pm == 0.0:
                    continue
                mm = m; dp_rate = 0.0
                while mm != 0:
                    lsb = mm & -mm
                    mm -= lsb
                    j = lsb.bit_length() - 1
                    flux = Ei * Pm
                    dy[base + m] -= flux
                    dy[base + (m ^ lsb)] += flux
                    dp_rate += p.Dp_i[ss + j] + Di
                dy[base + m] -= dp_rate * Pm
            for m in range(nst):                       # build_random_transitions order (models.py:435-485): m ascending, j ascending
                for j in range(ns):
                    if (m & (1 << j)) == 0:
                        flux = S_all[ss + j] * y[base + m]
                        dy[base + m] -= flux
                        dy[base + (m | (1 << j))] += flux
    return dy


def fd_jacobian(net: Network, p: Params, y, t, eps=1e-8):
    """jacspeedup.py:398-588 (fd_jacobian_nb_core_*): forward differences, h = eps * max(1, |y_j|), J row-major."""
    y = np.asarray(y, float)
    n = y.size
    J = np.empty((n, n))
    f0 = rhs(net, p, y, t)
    for j in range(n):
        yp = y.copy()
        aj = y[j]
        h = eps * (1.0 if abs(aj) < 1.0 else abs(aj))
        yp[j] = aj + h
        J[:, j] = (rhs(net, p, yp, t) - f0) * (1.0 / h)
    return J


def default_y0(net: Network):
    """System.y0 (network.py:421-441): R = 1, P = 1 (state_0 = 1), phospho states 0.01."""
    y = np.zeros(net.S)
    for i in range(net.N):
        st = int(net.offset_y[i])
        y[st] = 1.0
        y[st + 1] = 1.0
        nrest = (int(net.n_states[i]) - 1) if net.model == 2 else int(net.n_sites[i])
        y[st + 2: st + 2 + nrest] = 0.01
    return y


def simulate_odeint(net: Network, p: Params, t_eval, rtol, atol, mxstep, y0=None, use_fd_jac=True):
    """simulate.py:34-80: odeint(rhs_odeint, y0, t, Dfun=fd_jacobian_odeint, col_deriv=False, rtol, atol, mxstep)."""
    y0 = default_y0(net) if y0 is None else np.asarray(y0, float)
    f = lambda y, t: rhs(net, p, y, t)
    kw = dict(Dfun=(lambda y, t: fd_jacobian(net, p, y, t)), col_deriv=False) if use_fd_jac else {}
    return np.ascontiguousarray(odeint(f, y0, np.asarray(t_eval, float), rtol=rtol, atol=atol, mxstep=mxstep, **kw))


# Dormand-Prince 5(4) tableau as used by the reference (solvers.py:338-365); rows of A for stages 2..6, 5th-order weights B (= row 7,
# FSAL), error weights E (5th minus 4th order)
DP_A = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
)
DP_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)
DP_E = (71 / 57600, 0.0, -71 / 16695, 71 / 1920, -17253 / 339200, 22 / 525, -1 / 40)


def simulate_rk45(net: Network, p: Params, t_eval, rtol=1e-5, atol=1e-7, y0=None, dt_init=0.05, dt_min=1e-6, dt_max=1.0, safety=0.9,
                  max_steps=2_000_000, return_steps=False):
    """The reference's opt-in explicit integrator (solvers.py:293-577 models 0/1/4, :580-758 model 2; reached through
    jacspeedup.solve_custom, jacspeedup.py:31-64), restated step for step:
      * the kinase bucket `jb` is carried by the integrator (advanced when tcur >= grid[jb + 1]); every stage of a step uses it;
      * a step is cut to land exactly on the next bucket edge and on t_final, and floored at dt_min;
      * error = max_i |dt * sum_j E_j k_j| / max(atol + rtol * max(|y_i|, |y_new_i|), 1e-12);
      * accepted: outputs inside (tcur, t_next] by cubic Hermite on (y, k1) / (y_new, k7); FSAL unless the step ended on a bucket edge;
        PI controller fac = safety * err^-(0.2 - 0.04) * err_prev^0.04 in [0.2, 5] (5 if err < 1e-12), dt <= dt_max, err_prev >= 1e-4;
      * rejected: fac = max(0.1, safety * err^-0.2), dt = max(dt_use * fac, dt_min), err_prev = 1.
    """
    t_eval = np.asarray(t_eval, float)
    grid = net.kin_grid
    y = (default_y0(net) if y0 is None else np.asarray(y0, float)).copy()
    T = t_eval.size
    Y = np.empty((T, y.size))
    Y[0] = y
    f = lambda yy, jb: rhs(net, p, yy, grid[jb])          # any t inside bucket jb gives that bucket's right-hand side
    jb = 0
    tcur, t_final = t_eval[0], t_eval[-1]
    while jb + 1 < grid.size and tcur >= grid[jb + 1]:
        jb += 1
    nxt = 1
    beta = 0.04
    alpha = 0.2 - beta
    err_prev = 1.0
    k = [None] * 7
    k[0] = f(y, jb)
    dt = dt_init
    steps = acc = 0
    hit = False
    while tcur < t_final and nxt < T:
        steps += 1
        if steps > max_steps:
            raise RuntimeError("Max steps exceeded")
        while jb + 1 < grid.size and tcur >= grid[jb + 1]:
            jb += 1
            hit = True
        if hit:
            k[0] = f(y, jb)
            hit = False
            err_prev = 1.0
        dt_use = dt
        dist = 1e9
        if jb + 1 < grid.size:
            dist = grid[jb + 1] - tcur
            if dist > 1e-15 and dt_use > dist:
                dt_use = dist
        dt_use = min(dt_use, t_final - tcur)
        dt_use = max(dt_use, dt_min)
        for s in range(1, 6):
            inc = DP_A[s - 1][0] * k[0]
            for j in range(1, s):
                inc = inc + DP_A[s - 1][j] * k[j]
            k[s] = f(y + dt_use * inc, jb)
        y_new = y + dt_use * (DP_B[0] * k[0] + DP_B[2] * k[2] + DP_B[3] * k[3] + DP_B[4] * k[4] + DP_B[5] * k[5])
        k[6] = f(y_new, jb)
        diff = dt_use * (DP_E[0] * k[0] + DP_E[2] * k[2] + DP_E[3] * k[3] + DP_E[4] * k[4] + DP_E[5] * k[5] + DP_E[6] * k[6])
        sc = np.maximum(atol + rtol * np.maximum(np.abs(y), np.abs(y_new)), 1e-12)
        err = float(np.max(np.abs(diff) / sc))
        if err <= 1.0:
            acc += 1
            t_next = tcur + dt_use
            while nxt < T and t_eval[nxt] <= t_next:
                te = t_eval[nxt]
                if te >= tcur:
                    h = t_next - tcur
                    if h < 1e-16:
                        Y[nxt] = y_new
                    else:
                        tau = (te - tcur) / h
                        t2, t3 = tau * tau, tau * tau * tau
                        Y[nxt] = ((2 * t3 - 3 * t2 + 1) * y + (t3 - 2 * t2 + tau) * h * k[0] + (-2 * t3 + 3 * t2) * y_new + (t3 - t2) * h * k[6])
                nxt += 1
            y = y_new
            tcur = t_next
            if abs(dt_use - dist) < 1e-14:
                hit = True
            else:
                k[0] = k[6]
            fac = 5.0 if err < 1e-12 else safety * err ** (-alpha) * err_prev ** beta
            fac = min(5.0, max(0.2, fac))
            dt = min(dt * fac, dt_max)
            err_prev = max(err, 1e-4)
        else:
            fac = max(0.1, safety * err ** (-0.2))
            dt = max(dt_use * fac, dt_min)
            err_prev = 1.0
    return (Y, acc, steps - acc) if return_steps else Y


# ------------------------------------------------------------------------------------------------ loss / objective
EPS_LOSS = 1e-9   # lossfn.py:24


def pointwise_loss(mode: int, diff: float, obs: float, pred: float) -> float:
    """The LOSS_MODE switch of lossfn.py:149-165 with the eight robust losses of lossfn.py:28-110."""
    if mode == 0:
        return diff * diff
    if mode == 1:
        a = abs(diff); d = 0.5
        return 0.5 * diff * diff if a <= d else d * (a - 0.5 * d)
    if mode == 2:
        with np.errstate(invalid="ignore", divide="ignore"):
            dl = float(np.log(diff + EPS_LOSS) - np.log(obs + EPS_LOSS))      # log of a possibly negative residual: NaN, as in the reference
        x = dl / 0.5
        return (0.5 * 0.5) * ((1.0 + x * x) ** 0.5 - 1.0)
    if mode == 3:
        s = abs(diff)
        return s - 0.69314718056 if s > 20.0 else float(np.log(np.cosh(diff)))
    if mode == 4:
        return float(np.log(1.0 + (diff / 1.0) ** 2))
    if mode == 5:
        return (diff * diff) / (abs(pred) + 1e-6)
    if mode == 6:
        x2 = diff * diff
        return x2 / (x2 + 1.0)
    return (diff * diff + 1e-3 * 1e-3) ** 0.5 - 1e-3


def loss_function(model: int, Y, ld: dict, mode: int):
    """loss_function_noncomb (lossfn.py:114-246) / loss_function_comb (:250-382): three raw weighted sums."""
    fc = lambda a, b: (a if a > EPS_LOSS else EPS_LOSS) / (b if b > EPS_LOSS else EPS_LOSS)
    pm = ld["prot_map"]
    lp = 0.0
    for k in range(ld["p_prot"].size):
        st, cnt = int(pm[ld["p_prot"][k], 0]), int(pm[ld["p_prot"][k], 1])
        t = int(ld["t_prot"][k]); b = int(ld["prot_base_idx"])
        if model == 2:
            tt = 0.0; tb = 0.0
            for m in range(cnt):
                tt += Y[t, st + 1 + m]; tb += Y[b, st + 1 + m]
        else:
            tt = Y[t, st + 1]; tb = Y[b, st + 1]
            for s in range(cnt):
                tt += Y[t, st + 2 + s]; tb += Y[b, st + 2 + s]
        pred = fc(tt, tb)
        lp += ld["w_prot"][k] * pointwise_loss(mode, ld["obs_prot"][k] - pred, ld["obs_prot"][k], pred)
    lr = 0.0
    for k in range(ld["p_rna"].size):
        st = int(pm[ld["p_rna"][k], 0])
        pred = fc(Y[int(ld["t_rna"][k]), st], Y[int(ld["rna_base_idx"]), st])
        lr += ld["w_rna"][k] * pointwise_loss(mode, ld["obs_rna"][k] - pred, ld["obs_rna"][k], pred)
    lph = 0.0
    for k in range(ld["p_pho"].size):
        st, cnt = int(pm[ld["p_pho"][k], 0]), int(pm[ld["p_pho"][k], 1])
        t = int(ld["t_pho"][k]); b = int(ld["pho_base_idx"]); j = int(ld["s_pho"][k])
        if model == 2:
            a = 0.0; c = 0.0
            for m in range(cnt):
                if m & (1 << j):
                    a += Y[t, st + 1 + m]; c += Y[b, st + 1 + m]
        else:
            a = Y[t, st + 2 + j]; c = Y[b, st + 2 + j]
        pred = fc(a, c)
        lph += ld["w_pho"][k] * pointwise_loss(mode, ld["obs_pho"][k] - pred, ld["obs_pho"][k], pred)
    return lp, lr, lph


def objectives(net: Network, x_phys, defaults, Y, ld: dict, mode: int, lambdas: dict, fail_value: float = 1e12):
    """GlobalODE_MOO._evaluate after the simulation (optproblem.py:99-160): prior penalty on A, B, C, D, E, finite check,
    normalised weighted objectives."""
    nK, N, sites = net.n_K, net.N, net.total_sites
    sl = {"A_i": (nK, nK + N), "B_i": (nK + N, nK + 2 * N), "C_i": (nK + 2 * N, nK + 3 * N), "D_i": (nK + 3 * N, nK + 4 * N),
          "E_i": (nK + 4 * N + sites, nK + 5 * N + sites)}
    acc = 0.0; cnt = 0
    for k, (a, b) in sl.items():
        diff = (x_phys[a:b] - defaults[a:b]) / (defaults[a:b] + 1e-6)
        acc += float(np.sum(diff ** 2)); cnt += diff.size
    prior = lambdas["prior"] * (acc / max(1, cnt))
    if Y is None or not np.all(np.isfinite(Y)):
        return np.full(3, fail_value)
    lp, lr, lph = loss_function(net.model, Y, ld, mode)
    norm = lambda w: 1.0 / max(1e-6, float(np.sum(w)))
    return np.array([lp * norm(ld["w_prot"]) * lambdas["protein"] + prior, lr * norm(ld["w_rna"]) * lambdas["rna"] + prior,
                     lph * norm(ld["w_pho"]) * lambdas["phospho"] + prior])


def frechet_distance(true_coords, pred_coords) -> float:
    """frechet/distance.py:9-56: pairwise Euclidean distances of the points, then c[i][j] = max(min(c[i-1][j], c[i][j-1], c[i-1][j-1]), d[i][j])."""
    a = np.asarray(true_coords, float); b = np.asarray(pred_coords, float)
    n, m = len(a), len(b)
    d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(axis=2))
    c = np.full((n, m), np.inf)
    c[0, 0] = d[0, 0]
    for i in range(1, n):
        c[i, 0] = max(c[i - 1, 0], d[i, 0])
    for j in range(1, m):
        c[0, j] = max(c[0, j - 1], d[0, j])
    for i in range(1, n):
        for j in range(1, m):
            c[i, j] = max(min(c[i - 1, j], c[i, j - 1], c[i - 1, j - 1]), d[i, j])
    return float(c[-1, -1])


# ------------------------------------------------------------------------------------------ decision vector (global_model/params.py)
PARAM_KEYS = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")
BOUNDS_CONFIG = {"c_k": (1e-3, 4.0), "A_i": (1e-6, 10.0), "B_i": (1e-3, 1.0), "C_i": (1e-3, 2.0), "D_i": (0.1, 0.5), "Dp_i": (0.05, 5.0),
                 "E_i": (1e-4, 10.0), "tf_scale": (2.0, 10.0)}                       # config.toml:368-397


def softplus(x):
    """global_model/utils.py:229-241, element by element."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    for i in range(x.size):
        xi = x.flat[i]
        out.flat[i] = xi if xi > 20.0 else np.log1p(np.exp(xi))
    return out


def inv_softplus(y):
    """global_model/utils.py:244-253."""
    y = np.asarray(y, dtype=np.float64)
    out = np.empty_like(y)
    for i in range(y.size):
        yi = y.flat[i]
        if yi < 1e-12:
            yi = 1e-12
        out.flat[i] = np.log(np.expm1(yi))
    return out


def init_raw_params(defaults: dict, custom_bounds: Optional[dict] = None):
    """global_model/params.py:24-103: (theta0, slices, xl, xu) in raw space, groups in PARAM_KEYS order then tf_scale."""
    custom_bounds = custom_bounds or {}
    vecs, slices, bounds, curr = [], {}, [], 0
    for k in PARAM_KEYS:
        raw = inv_softplus(defaults[k])
        vecs.append(raw)
        slices[k] = slice(curr, curr + len(raw)); curr += len(raw)
        pmin, pmax = custom_bounds[k] if k in custom_bounds else BOUNDS_CONFIG[k]
        bounds.extend([(inv_softplus(np.array([pmin]))[0], inv_softplus(np.array([pmax]))[0])] * len(raw))
    vecs.append(inv_softplus(np.array([defaults["tf_scale"]])))
    slices["tf_scale"] = slice(curr, curr + 1)
    pmin, pmax = custom_bounds["tf_scale"] if "tf_scale" in custom_bounds else BOUNDS_CONFIG["tf_scale"]
    bounds.append((inv_softplus(np.array([pmin]))[0], inv_softplus(np.array([pmax]))[0]))
    return np.concatenate(vecs), slices, np.array([b[0] for b in bounds], float), np.array([b[1] for b in bounds], float)


def unpack_params(theta, slices) -> Params:
    """global_model/params.py:106-132."""
    theta = np.asarray(theta, float)
    return Params(*(softplus(theta[slices[k]]) for k in PARAM_KEYS), float(softplus(theta[slices["tf_scale"]])[0]))


def params_to_row(p: Params) -> np.ndarray:
    return np.concatenate([np.ravel(getattr(p, k)) for k in PARAM_KEYS] + [[p.tf_scale]])


# ------------------------------------------------------------------------------------------ observables (global_model/simulate.py:83-202)
def simulate_and_measure(net: Network, p: Params, t_points_p, t_points_r, t_points_pho, y0=None, rtol=1e-5, atol=1e-7, mxstep=5000):
    """The three pred_fc frames of simulate_and_measure as arrays, rows in the reference's order (protein-major, then [site,] time):
    one LSODA solve at rtol 1e-5 / atol 1e-7, mxstep 5000 on the union grid, fold change against t = 0 (protein, phospho) / t = 4 (RNA)
    with the 1e-12 floors, rows kept where the time is in the modality's list.
    Returns dict(p_i, p_t, p_fc, r_i, r_t, r_fc, ph_i, ph_s, ph_t, ph_fc)."""
    times = np.unique(np.concatenate([t_points_p, t_points_r, t_points_pho]).astype(np.float64))
    Y = simulate_odeint(net, p, times, rtol, atol, mxstep, y0=y0)          # defaults = the reference's hard-wired values (simulate.py:110)
    bidx = lambda t0: int(np.argmin(np.abs(times - float(t0))))
    pb, rb, phb = bidx(0.0), bidx(4.0), bidx(0.0)
    keep_p, keep_r, keep_ph = (np.isin(times, np.asarray(x, float)) for x in (t_points_p, t_points_r, t_points_pho))
    out = {k: [] for k in ("p_i", "p_t", "p_fc", "r_i", "r_t", "r_fc", "ph_i", "ph_s", "ph_t", "ph_fc")}
    for i in range(net.N):
        st = int(net.offset_y[i])
        R = Y[:, st]
        fc_r = np.maximum(R, 1e-12) / np.maximum(R[rb], 1e-12)
        out["r_i"].append(np.full(keep_r.sum(), i)); out["r_t"].append(times[keep_r]); out["r_fc"].append(fc_r[keep_r])
        ns = int(net.n_sites[i])
        if net.model == 2:
            cnt = int(net.n_states[i]) if net.n_states is not None else (1 << ns)
            states = Y[:, st + 1: st + 1 + cnt]
            tot = states.sum(axis=1)
            if ns > 0:
                m = np.arange(cnt, dtype=np.uint32)[:, None]; j = np.arange(ns, dtype=np.uint32)[None, :]
                sites = states @ ((m >> j) & 1).astype(np.float64)
            else:
                sites = None
        else:
            P0 = Y[:, st + 1]
            sites = Y[:, st + 2: st + 2 + ns] if ns > 0 else None
            tot = P0 + (sites.sum(axis=1) if ns > 0 else np.zeros_like(P0))
        fc_p = np.maximum(tot, 1e-12) / np.maximum(tot[pb], 1e-12)
        out["p_i"].append(np.full(keep_p.sum(), i)); out["p_t"].append(times[keep_p]); out["p_fc"].append(fc_p[keep_p])
        if sites is not None:
            for s in range(ns):
                sig = sites[:, s]
                fc = np.maximum(sig, 1e-12) / np.maximum(sig[phb], 1e-12)
                out["ph_i"].append(np.full(keep_ph.sum(), i)); out["ph_s"].append(np.full(keep_ph.sum(), s))
                out["ph_t"].append(times[keep_ph]); out["ph_fc"].append(fc[keep_ph])
    cat = lambda v, dt: (np.concatenate(v).astype(dt) if v else np.zeros(0, dt))
    return {k: cat(v, np.int32 if k.endswith("_i") or k.endswith("_s") else np.float64) for k, v in out.items()}


# ------------------------------------------------------------------------------------------ network Morris helpers (global_model/sensitivity.py)
def compute_bounds(params_dict: dict, perturbation: float = 0.05):
    """global_model/sensitivity.py:41-80 (default perturbation = config.toml:351 sensitivity_perturbation)."""
    bounds, names = [], []
    for key, value in params_dict.items():
        if isinstance(value, np.ndarray):
            for i, v in enumerate(value):
                lb = v * (1 - perturbation); ub = v * (1 + perturbation)
                if abs(v) < 1e-6:
                    lb, ub = 0.0, 0.01
                bounds.append([max(0.0, lb), ub]); names.append(f"{key}_{i}")
        else:
            v = float(value)
            lb = v * (1 - perturbation); ub = v * (1 + perturbation)
            if abs(v) < 1e-6:
                lb, ub = 0.0, 0.01
            bounds.append([max(0.0, lb), ub]); names.append(key)
    return {"num_vars": len(names), "names": names, "bounds": bounds}


def reconstruct_params(param_vector, original_shapes: dict) -> dict:
    """global_model/sensitivity.py:83-103."""
    out, curr = {}, 0
    for key, shape in original_shapes.items():
        if shape == ():
            out[key] = param_vector[curr]; curr += 1
        else:
            size = int(np.prod(shape))
            out[key] = np.array(param_vector[curr: curr + size]); curr += size
    return out


def compute_scalar_metric(v_prot, v_rna, v_pho, metric: str = "total_signal") -> float:
    """global_model/sensitivity.py:106-140 on the pred_fc columns (protein, rna, phospho order)."""
    combined = np.concatenate([np.asarray(v_prot, float), np.asarray(v_rna, float), np.asarray(v_pho, float)])
    if len(combined) == 0:
        return 0.0
    if metric == "total_signal":
        return float(np.sum(combined))
    if metric == "mean":
        return float(np.mean(combined))
    if metric == "variance":
        return float(np.var(combined))
    if metric == "l2_norm":
        return float(np.linalg.norm(combined))
    return float(np.sum(combined))


# ------------------------------------------------------------------------------------------ one optimiser candidate (global_model/optproblem.py:87-160)
def evaluate(net: Network, x_raw, slices, defaults_row, ld: dict, mode: int, lambdas: dict, time_grid, fail_value: float = 1e12,
             rtol: float = 1e-8, atol: float = 1e-8, mxstep: int = 200000):
    """GlobalODE_MOO._evaluate: unpack (softplus) -> prior penalty -> simulate_odeint at the optimiser's tolerances -> LOSS_FN ->
    the three objectives; fail_value when the trajectory is not finite."""
    p = unpack_params(x_raw, slices)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            Y = simulate_odeint(net, p, np.asarray(time_grid, float), rtol, atol, mxstep)
        except Exception:
            Y = None
    return objectives(net, params_to_row(p), np.asarray(defaults_row, float), Y, ld, mode, lambdas, fail_value)
