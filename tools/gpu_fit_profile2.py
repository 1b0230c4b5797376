"""Dev: cProfile of the lockstep LM driver at BASELINE config 3's size (480 rows, distmod n = 30), Jacobian from the sensitivity kernel."""
import sys, pathlib, cProfile, pstats, time
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from phoskintime_amd import batch
from phoskintime_amd.paramest import fit_rows_batch, multistart_candidates
TG = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
n, rows, iters = 30, 480, 12
jac = sys.argv[1] if len(sys.argv) > 1 else "sens"
P, S = 4 + 2 * n, n + 2
rng = np.random.default_rng(20260515 + 9)
th_true = rng.uniform(0.2, 2.0, P)
flat = batch.solve_ode_batch("distmod", th_true[None], np.ones(S), n, TG, want_sol=False).flat[0].cpu().numpy()
target = np.abs(flat * (1 + 0.02 * rng.standard_normal(flat.size)))
lb, ub = np.zeros(P), np.full(P, 20.0)
P0 = multistart_candidates("BENCH", rng.uniform(lb, ub), lb, ub, n_starts=rows)
fit_rows_batch("distmod", n, TG, P0[:4], np.ones(S), target, bounds=(lb, ub), max_iter=2, jacobian=jac)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fit = fit_rows_batch("distmod", n, TG, P0, np.ones(S), target, bounds=(lb, ub), max_iter=iters, jacobian=jac)
torch.cuda.synchronize()
pr.disable()
print("wall ms", 1e3 * (time.perf_counter() - t0), "iters", fit.n_iter, "launches", fit.n_launches)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
