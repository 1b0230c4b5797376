"""Dev: the popcount-level exact kernel of randmod n = 8 (csrc/pk_rand_level.hpp): band error against the oracle's closed form, step counts,
time per batch; n = 6 through the same kernel (PK_RAND_LEVEL6=1) as an A/B against the one-wave kernel."""
import os, sys, time, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from phoskintime_amd import batch
from oracle import protein_models as pm
t = pm.TIME_POINTS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P, S = pm.n_params(2, n), pm.n_states(2, n)
rng = np.random.default_rng(8)
sets = {"U(0.05,2)": rng.uniform(0.05, 2.0, (B, P)), "U(0,20)": rng.uniform(0.0, 20.0, (B, P)), "logU(1e-3,1e2)": np.exp(rng.uniform(np.log(1e-3), np.log(1e2), (B, P)))}
for name, th in sets.items():
    y0 = np.ones(S)
    r = batch.solve_ode_batch("randmod", th, y0, n, t, clip_nonneg=False)
    torch.cuda.synchronize()
    dts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = batch.solve_ode_batch("randmod", th, y0, n, t, clip_nonneg=False)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    print("   reps ms:", " ".join(f"{1e3 * x:.1f}" for x in dts))
    sol = r.sol.cpu().numpy(); st = r.status.cpu().numpy(); ns = r.n_steps.cpu().numpy()
    worst = 0.0
    for b in range(0, B, max(1, B // 24)):
        worst = max(worst, pm.band_error(sol[b], pm.solve_exact_lti(2, th[b], y0, n, t)))
    print(f"n={n} {name}: B={B} {1e3 * dt:.1f} ms = {B / dt:.0f} replicas/s; steps mean {ns[:, 0].mean():.1f} max {ns[:, 0].max()} rejected mean {ns[:, 1].mean():.2f}; "
          f"flagged {int((st != 0).sum())}; worst band error on 24 samples {worst:.3f}", flush=True)
