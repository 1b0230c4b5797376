"""Dev tool: cProfile of one rows-batched LM fit (where does the host time of an iteration go?)."""
import sys, pathlib, cProfile, pstats, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch
from phoskintime_amd.paramest import multistart as ms
from oracle import protein_models as pm
model, n, R = "distmod", 4, 48
mid = pm.MODEL_IDS[model]; S, P = pm.n_states(mid, n), pm.n_params(mid, n)
rng = np.random.default_rng(3)
truth = rng.uniform(0.5, 1.5, size=P); y0 = np.ones(S); t = pm.TIME_POINTS
target = batch.solve_ode_batch(model, truth[None], y0, n, t, want_sol=False).flat.cpu().numpy()[0]
P0 = truth * rng.uniform(0.7, 1.4, size=(R, P)); lb, ub = np.full(P, 1e-3), np.full(P, 10.0)
ms.fit_rows_batch(model, n, t, P0, y0, target, bounds=(lb, ub), max_iter=60)
t0 = time.perf_counter(); f = ms.fit_rows_batch(model, n, t, P0, y0, target, bounds=(lb, ub), max_iter=60); print("fit %.1f ms, %d iterations, %d launches" % (1e3 * (time.perf_counter() - t0), f.n_iter, f.n_launches))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    ms.fit_rows_batch(model, n, t, P0, y0, target, bounds=(lb, ub), max_iter=60)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
