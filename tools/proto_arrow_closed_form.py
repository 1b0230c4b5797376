"""Numpy experiment (dev tool): closed-form solution of the distributive model through the secular equation of its arrow matrix
(eigenvalues by bisection inside the pole intervals, eigenvectors in closed form, forcing integrals analytic).  Accuracy on config-3-like
inputs: band error 2e-4 .. 5e-4 against the reference RHS under SciPy odeint at 1e-13 / the matrix exponential.  Not hardened: exactly
equal poles, B = 0 and roots within 1e-9 of a pole need deflation / series forms."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from oracle import protein_models as pm

def solve_arrow_closed(theta, y0, n, t):
    A,B,C,D = theta[:4]; S = theta[4:4+n].astype(float); d = 1.0 + theta[4+n:4+2*n]
    Dsum = D + S.sum()
    R0,P0 = y0[0], y0[1]; X0 = y0[2:2+n].astype(float)
    # active sites (S>0); inactive: X_i(t) = X0 e^{-d t}, and they still feed P!  (dP has + sum X_i regardless of S_i)
    # => treat exactly: K = [[-Dsum, 1^T],[S, -diag d]] nonsymmetric when S_i = 0. Use scaling only for S_i>0; S_i == 0 sites act as decaying forcing on P.
    act = S > 0
    sig = np.sqrt(S[act]); da = d[act]; na = act.sum()
    # secular: g(mu) = Dsum - mu - sum S_i/(d_i - mu) ; roots interlace sorted poles
    order = np.argsort(da); dp = da[order]; Sp = S[act][order]
    # merge (near-)equal poles? keep simple
    roots = []
    def g(mu): return Dsum - mu - np.sum(Sp/(dp - mu))
    # intervals: (-inf, dp[0]), (dp[0],dp[1]), ..., (dp[-1], +inf)
    edges = np.concatenate([[-np.inf], dp, [np.inf]])
    for k in range(na+1):
        lo, hi = edges[k], edges[k+1]
        if hi - lo == 0:      # duplicate pole: root equals the pole (deflation)
            roots.append(lo); continue
        # g is decreasing in mu on each interval? dg/dmu = -1 - sum S/(d-mu)^2 < 0. g -> +inf at lo+ (for finite lo: -S/(d-mu) with mu>d => +), -> -inf at hi-
        if not np.isfinite(lo):
            lo = min(dp[0], Dsum) - (abs(Dsum) + Sp.sum()/max(1e-300,1) + 1.0)   # g(lo) > 0 far left?
            lo = -1.0 - abs(Dsum) - Sp.sum()
        if not np.isfinite(hi): hi = max(dp[-1], Dsum) + Sp.sum() + abs(Dsum) + 1.0
        a_, b_ = lo, hi
        # bisection with offsets
        for it in range(200):
            m = 0.5*(a_+b_)
            if m == a_ or m == b_: break
            val = g(m)
            if val > 0: a_ = m
            else: b_ = m
        roots.append(0.5*(a_+b_))
    mu = np.array(roots)          # na+1 decay rates
    # eigenvectors of symmetric arrow (in scaled coordinates [P, w_i]): v_k = [1, sig_i/(d_i - mu_k)] / norm
    sigp = np.sqrt(Sp)
    V = np.empty((na+1, na+1))
    for k in range(na+1):
        v = np.concatenate([[1.0], sigp/(dp - mu[k])])
        V[:,k] = v/np.linalg.norm(v)
    out = np.empty((len(t), 2+n))
    Rinf = A/B if B != 0 else np.nan
    w0 = np.concatenate([[P0], (X0[act][order])/np.where(sigp>0,sigp,1)])
    c0 = V.T @ w0
    e0 = V[0,:]                   # V^T e_0
    inact = ~act
    for ti, tt in enumerate(t):
        tau = tt - t[0]
        R = Rinf + (R0 - Rinf)*np.exp(-B*tau)
        em = np.exp(-mu*tau)
        # forcing on P: C R(s) + sum_{inactive} X0_j e^{-d_j s}
        def conv_const(): return np.where(np.abs(mu) > 0, -np.expm1(-mu*tau)/np.where(mu!=0,mu,1), tau)
        def conv_exp(b):       # int_0^tau e^{-mu (tau-s)} e^{-b s} ds
            dm = mu - b
            small = np.abs(dm*tau) < 1e-6
            val = np.where(small, tau*np.exp(-b*tau)*(1 - 0.5*dm*tau), (np.exp(-b*tau) - em)/np.where(dm!=0,dm,1))
            return val
        coef = c0*em + e0*(C*Rinf*conv_const() + C*(R0-Rinf)*conv_exp(B))
        for j in np.where(inact)[0]:
            coef = coef + e0*X0[j]*conv_exp(d[j])
        w = V @ coef
        out[ti,0] = R; out[ti,1] = w[0]
        Xa = np.empty(na); Xa[order] = w[1:]*sigp
        X = np.empty(n); X[act] = Xa; X[inact] = X0[inact]*np.exp(-d[inact]*tau)
        out[ti,2:] = X
    return out

if __name__ == '__main__':
    g = np.load(str(pathlib.Path(__file__).resolve().parents[1] / 'tests/golden/protein_distmod_n30_c3bounds.npz'))
    worst=0
    for k in range(64):
        sol = solve_arrow_closed(g['theta'][k], g['y0'][k], 30, g['t'])
        e = pm.band_error(sol, g['sol_tight'][k]); worst=max(worst,e)
    print('c3bounds worst band', worst)
    rng = np.random.default_rng(0); worst=0; bad=0
    for k in range(300):
        n = 30; th = rng.uniform(0,20,64); y0=np.ones(32)
        sol = solve_arrow_closed(th, y0, n, pm.TIME_POINTS)
        ex = pm.solve_exact_lti(pm.DIST, th, y0, n, pm.TIME_POINTS)
        e = pm.band_error(sol, ex); worst=max(worst,e); bad += e>0.1
    print('random worst band', worst, 'bad', bad)
