"""Numpy model of the adaptive resolvent-form steppers (RODAS4-resolvent and LRP s/p) on the golden sets (dev tool)."""
import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from oracle import protein_models as pm
import restricted_pade as rp
import mpmath as mp

def make(s, gamma):
    g = mp.mpf(gamma)
    beta = rp.solve_weights(s, g, s - 1)
    bh = rp.solve_weights(s, g, s - 2, extra_zero=(s,))
    return [float(x) for x in beta], [float(x - y) for x, y in zip(beta, bh)], float(g), s - 1   # estimator order s-2 -> exponent 1/(s-1)

RODAS = ([0.25, -0.30261563040255475781, 2.0437958549435498747, -1.1490271157486588988, 0.12712918827688397052, 0.030717702930779034113],
         [0.0, -0.27896255898136486925, 0.80616997401331598084, -0.74473456815175834943, 0.18680945018902833694, 0.030717702930779034113], 0.25, 4)

def solve(M, b, y0, t, rtol, atol, meth, safe=0.9, facmax=6.0, facmin=0.2):
    beta, eps, gam, q = meth
    S = len(y0); I = np.eye(S)
    y = y0.copy(); out = np.empty((len(t), S)); out[0] = y
    f = lambda y: M @ y + b
    sc = atol + rtol*np.abs(y); f0 = f(y)
    d0 = np.max(np.abs(y)/sc); d1 = np.max(np.abs(f0)/sc)
    h = 0.01*d0/d1 if (d0 > 1e-5 and d1 > 1e-5) else 1e-6
    nst = nrej = 0
    for k in range(1, len(t)):
        tc, te = t[k-1], t[k]
        while True:
            last = tc + 1.0001*h >= te
            hs = te - tc if last else (0.5*(te-tc) if tc + 2*h > te else h)
            Minv = np.linalg.inv(I - gam*hs*M)
            z = Minv @ (hs*f(y)); yn = y + beta[0]*z; e = eps[0]*z
            for j in range(1, len(beta)):
                z = Minv @ z; yn = yn + beta[j]*z; e = e + eps[j]*z
            scl = atol + rtol*np.maximum(np.abs(y), np.abs(yn))
            err = np.max(np.abs(e)/scl)
            fac = max(1/facmax, min(1/facmin, err**(1.0/q)/safe))
            hnew = hs/fac; nst += 1
            if err <= 1.0:
                y = yn; tc += hs
                if last:
                    h = max(hnew, h) if hs < h else hnew
                    break
                h = hnew
            else:
                nrej += 1; h = hnew
        out[k] = y
    return out, nst, nrej

if __name__ == '__main__':
    meths = {'rodas4': RODAS}
    for s, gm in ((6, '0.25'), (6, '0.28'), (6, '0.22'), (8, '0.2'), (8, '0.22'), (5, '0.28'), (7, '0.25')):
        meths['lrp%d_g%s' % (s, gm)] = make(s, gm)
    files = ['tests/golden/protein_distmod_n30_c3bounds.npz', 'tests/golden/protein_succmod_n14_c2bounds.npz', 'tests/golden/protein_randmod_n4_bounds.npz']
    for fn in files:
        g = np.load(fn); model = pm.MODEL_IDS[str(g['model'])]; n = int(g['n_sites'])
        K = min(12, g['theta'].shape[0])
        sys_ = [pm.lti_matrix(model, g['theta'][k], n) for k in range(K)]
        for name, meth in meths.items():
            for rtol, atol in [(1e-6, 1e-8), (1e-7, 1e-9)]:
                worst = 0; steps = []; rej = []
                for k in range(K):
                    sol, nst, nrej = solve(sys_[k][0], sys_[k][1], g['y0'][k], g['t'], rtol, atol, meth)
                    worst = max(worst, pm.band_error(sol, g['sol_tight'][k])); steps.append(nst); rej.append(nrej)
                print('%-34s %-12s rtol %.0e: band %.4f steps mean %5.0f max %4d rej %4.1f  solves/replica %6.0f' % (fn.split('/')[-1], name, rtol, worst, np.mean(steps), max(steps), np.mean(rej), np.mean(steps)*len(meth[0])))
