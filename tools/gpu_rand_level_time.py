import sys, time, pathlib, numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from phoskintime_amd import batch
from oracle import protein_models as pm
n, B = 8, 256
th = np.random.default_rng(8).uniform(0.05, 2.0, (B, pm.n_params(2, n)))
kw = dict(clip_nonneg=False, max_steps=32)
batch.solve_ode_batch("randmod", th, np.ones(257), n, pm.TIME_POINTS, **kw); torch.cuda.synchronize()
t0 = time.perf_counter(); r = batch.solve_ode_batch("randmod", th, np.ones(257), n, pm.TIME_POINTS, **kw); torch.cuda.synchronize()
ns = r.n_steps.cpu().numpy().sum(1).mean()
print("ms", 1e3 * (time.perf_counter() - t0), "steps", ns, "us/step", 1e6 * (time.perf_counter() - t0) / ns)
