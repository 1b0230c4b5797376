"""Dev: wide randmod kernel (method chosen by PK_WIDE_RAND_ROSW) on the n = 7, 8 fixtures: band, steps, one-theta latency."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pathlib import Path
from oracle import protein_models as pm
from phoskintime_amd import batch, models
for f in sorted(Path("tests/golden").glob("protein_randmod_n[78]_*.npz")):
    g = np.load(f); n = int(g["n_sites"])
    r = batch.solve_ode_batch(2, g["theta"], g["y0"], n, g["t"], clip_nonneg=False)
    sol = r.sol.cpu().numpy(); ns = r.n_steps.cpu().numpy()
    bands = [pm.band_error(sol[k], g["sol_tight"][k]) for k in range(sol.shape[0])]
    print(f.name, "bands", np.round(bands, 3), "steps", ns[:, 0], "rej", ns[:, 1], flush=True)
models.set_model("randmod")
rng = np.random.default_rng(0)
for n in (7, 8, 9):
    P, S = pm.n_params(pm.RAND, n), pm.n_states(pm.RAND, n)
    th = rng.uniform(0.05, 5.0, P)
    models.solve_ode(th, np.ones(S), n, pm.TIME_POINTS)
    t0 = time.perf_counter()
    for _ in range(3): models.solve_ode(th, np.ones(S), n, pm.TIME_POINTS)
    print("randmod n=%d one-theta solve_ode: %.2f ms" % (n, 1e3 * (time.perf_counter() - t0) / 3), flush=True)
