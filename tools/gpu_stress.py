"""Dev tool: randomized stress of the per-protein solve against the closed-form LTI oracle (band error, flags)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
from phoskintime_amd import batch
from oracle import protein_models as pm

rng = np.random.default_rng(123)
worst = {}
cases = []
for model, ns in ((0, (1, 3, 7, 15, 31, 47, 62)), (1, (1, 2, 5, 13, 29, 62)), (2, (1, 2, 3, 4, 5, 6))):
    for n in ns:
        cases.append((model, n))
for model, n in cases:
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    for kind in ('u20', 'log', 'tiny', 'mixed0', 'bigy0', 'longt', 'shortt'):
        B = 6
        if kind == 'u20': th = rng.uniform(0, 20, (B, P))
        elif kind == 'log': th = np.exp(rng.uniform(np.log(1e-8), np.log(20), (B, P)))
        elif kind == 'tiny': th = rng.uniform(0, 1e-3, (B, P))
        elif kind == 'mixed0': th = rng.uniform(0, 20, (B, P)) * (rng.uniform(size=(B, P)) > 0.4)
        else: th = rng.uniform(0.05, 5, (B, P))
        y0 = rng.uniform(0, 50, S) if kind == 'bigy0' else np.ones(S)
        t = pm.TIME_POINTS
        if kind == 'longt': t = np.array([0.0, 1.0, 1e2, 1e4, 1e5])
        if kind == 'shortt': t = np.array([0.0, 1e-6, 1e-4, 1e-2])
        r = batch.solve_ode_batch(model, th, y0, n, t, clip_nonneg=False)
        sol = r.sol.cpu().numpy(); st = r.status.cpu().numpy(); nsx = r.n_steps.cpu().numpy()
        for b in range(2 if S > 40 else 3):
            ex = pm.solve_exact_lti(model, th[b], y0, n, t)
            e = pm.band_error(sol[b], ex)
            key = (model, kind)
            worst[key] = max(worst.get(key, 0), e)
            if e > 0.3 or st[b] != 0:
                print('!! model', model, 'n', n, kind, 'replica', b, 'band', e, 'status', st[b], 'steps', nsx[b])
for k in sorted(worst): print(k, '%.4f' % worst[k])
