"""Numpy model of the adaptive RODAS4 stepper the HIP kernel implements (dev tool: picks tolerances,
counts steps).  Same control flow as csrc/pk_protein.hip: interval-by-interval, land exactly on t_k."""
import numpy as np, glob, sys
sys.path.insert(0, '.')
from oracle import protein_models as pm

G = 0.25
A21=0.1544000000000000e+01; A31=0.9466785280815826e+00; A32=0.2557011698983284e+00
A41=0.3314825187068521e+01; A42=0.2896124015972201e+01; A43=0.9986419139977817e+00
A51=0.1221224509226641e+01; A52=0.6019134481288629e+01; A53=0.1253708332932087e+02; A54=-0.6878860361058950e+00
C21=-0.5668800000000000e+01; C31=-0.2430093356833875e+01; C32=-0.2063599157091915e+00
C41=-0.1073529058151375e+00; C42=-0.9594562251023355e+01; C43=-0.2047028614809616e+02
C51=0.7496443313967647e+01; C52=-0.1024680431464352e+02; C53=-0.3399990352819905e+02; C54=0.1170890893206160e+02
C61=0.8083246795921522e+01; C62=-0.7981132988064893e+01; C63=-0.3152159432874371e+02; C64=0.1631930543123136e+02; C65=-0.6058818238834054e+01

def rodas4_solve(M, b, y0, t, rtol, atol, norm='max', safe=0.9, facmax=6.0, facmin=0.2, max_steps=100000):
    S = len(y0); I = np.eye(S)
    f = lambda y: M @ y + b
    y = y0.copy(); out = np.empty((len(t), S)); out[0] = y
    nst = nrej = nlu = 0
    # initial step (Hairer hinit-lite)
    sc = atol + rtol*np.abs(y); f0 = f(y)
    d0 = np.max(np.abs(y)/sc); d1 = np.max(np.abs(f0)/sc)
    h = 0.01*d0/d1 if (d0 > 1e-5 and d1 > 1e-5) else 1e-6
    h = min(h, t[1]-t[0])
    hold_err = None
    for k in range(1, len(t)):
        tc, te = t[k-1], t[k]
        last = False
        while True:
            if tc + h*1.0001 >= te:
                hs = te - tc; last = True
            else:
                hs = h; last = False
            W = I/(G*hs) - M; nlu += 1
            Winv = np.linalg.inv(W)
            solve = lambda r: Winv @ r
            f0 = f(y)
            u1 = solve(f0)
            u2 = solve(f(y + A21*u1) + (C21*u1)/hs)
            u3 = solve(f(y + A31*u1 + A32*u2) + (C31*u1 + C32*u2)/hs)
            u4 = solve(f(y + A41*u1 + A42*u2 + A43*u3) + (C41*u1 + C42*u2 + C43*u3)/hs)
            yn = y + A51*u1 + A52*u2 + A53*u3 + A54*u4
            u5 = solve(f(yn) + (C51*u1 + C52*u2 + C53*u3 + C54*u4)/hs)
            yn = yn + u5
            u6 = solve(f(yn) + (C61*u1 + C62*u2 + C63*u3 + C64*u4 + C65*u5)/hs)
            yn = yn + u6
            sc = atol + rtol*np.maximum(np.abs(y), np.abs(yn))
            err = np.max(np.abs(u6)/sc) if norm == 'max' else np.sqrt(np.mean((u6/sc)**2))
            fac = max(1.0/facmax, min(1.0/facmin, err**0.25/safe))
            hnew = hs/fac
            nst += 1
            if nst > max_steps: raise RuntimeError('max steps')
            if err <= 1.0:
                y = yn; tc = tc + hs
                if last:
                    # keep the un-truncated proposal for the next interval
                    h = max(hnew, h) if hs < h else hnew
                    break
                h = hnew
            else:
                nrej += 1
                h = hnew
        out[k] = y
    return out, nst, nrej

if __name__ == '__main__':
    files = sys.argv[1:] or ['tests/golden/protein_distmod_n30_c3bounds.npz', 'tests/golden/protein_succmod_n14_c2bounds.npz',
                             'tests/golden/protein_randmod_n4_bounds.npz', 'tests/golden/protein_distmod_n4_edge.npz']
    for fn in files:
        g = np.load(fn); model = pm.MODEL_IDS[str(g['model'])]; n = int(g['n_sites'])
        for rtol, atol in [(1e-6,1e-8),(1e-7,1e-9),(1e-8,1e-10),(1e-9,1e-11)]:
            worst = 0; steps = []; rej = []; wd = 0
            for k in range(min(16, g['theta'].shape[0])):
                M, b = pm.lti_matrix(model, g['theta'][k], n)
                sol, nst, nrej = rodas4_solve(M, b, g['y0'][k], g['t'], rtol, atol)
                worst = max(worst, pm.band_error(sol, g['sol_tight'][k]))
                wd = max(wd, pm.band_error(np.clip(sol,0,None), g['sol_default'][k]))
                steps.append(nst); rej.append(nrej)
            print(fn.split('/')[-1], 'rtol %.0e atol %.0e: band err vs tight %.3f, vs default %.3f; steps med %d max %d, rej med %d' % (rtol, atol, worst, wd, np.median(steps), max(steps), np.median(rej)))
        ref = max(pm.band_error(g['sol_default'][k], np.clip(g['sol_tight'][k],0,None)) for k in range(min(16, g['theta'].shape[0])))
        print('   reference default-vs-tight band err: %.3f' % ref)
