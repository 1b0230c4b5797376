"""Dev: host vs device Levenberg-Marquardt algebra of paramest.fit_rows_batch -- same iterates?  wall time?  (run on the GPU box)"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from phoskintime_amd import batch
from phoskintime_amd.paramest import fit_rows_batch, multistart_candidates

TG = bench.TGRID
for n, rows, iters in ((8, 48, 30), (8, 480, 30), (30, 480, 12)):
    P, S = 4 + 2 * n, n + 2
    rng = np.random.default_rng(20260515 + 9)
    th = rng.uniform(0.2, 2.0, P)
    flat = batch.solve_ode_batch("distmod", th[None], np.ones(S), n, TG, want_sol=False).flat[0].cpu().numpy()
    target = np.abs(flat * (1 + 0.02 * rng.standard_normal(flat.size)))
    lb, ub = np.zeros(P), np.full(P, 20.0)
    P0 = multistart_candidates("BENCH", rng.uniform(lb, ub), lb, ub, n_starts=rows)
    res = {}
    for lm in ("host", "device"):
        for jac in ("sens", "fd"):
            kw = dict(bounds=(lb, ub), device_algebra=True, jacobian=jac, lm_algebra=lm)
            fit_rows_batch("distmod", n, TG, P0, np.ones(S), target, max_iter=1, **kw)
            torch.cuda.synchronize(); t = time.perf_counter()
            f = fit_rows_batch("distmod", n, TG, P0, np.ones(S), target, max_iter=iters, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            res[(lm, jac)] = f
            print(n, rows, lm, jac, "wall_ms %.1f it %d solves %d best %.6e med %.6e" % (1e3 * dt, f.n_iter, f.n_solves, f.cost.min(), np.median(f.cost)), flush=True)
    for jac in ("sens", "fd"):
        a, b = res[("host", jac)], res[("device", jac)]
        print("  ", jac, "max |dp| %.3e  max rel dcost %.3e  JTJ rel %.3e" % (np.abs(a.p - b.p).max(), np.abs(a.cost - b.cost).max() / a.cost.max(),
                                                                             np.abs(a.JTJ - b.JTJ).max() / np.abs(a.JTJ).max()), flush=True)
