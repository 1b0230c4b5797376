"""Dev tool: wall time of the wide randmod kernel against the batch size (are the workgroups running concurrently?)."""
import sys, time, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch
from oracle import protein_models as pm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
model = sys.argv[2] if len(sys.argv) > 2 else "randmod"
mid = pm.MODEL_IDS[model]
S, P = pm.n_states(mid, n), pm.n_params(mid, n)
for dist in ("U(0.2,2)", "U(0,20)"):
    rng = np.random.default_rng(20260515)
    th_all = rng.uniform(0.2, 2.0, (1024, P)) if dist == "U(0.2,2)" else rng.uniform(0.0, 20.0, (1024, P))
    for B in (1, 4, 64, 256, 1024):
        th = torch.as_tensor(th_all[:B], device="cuda")
        batch.solve_ode_batch(model, th, np.ones(S), n, pm.TIME_POINTS, want_flat=False); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = batch.solve_ode_batch(model, th, np.ones(S), n, pm.TIME_POINTS, want_flat=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ns = r.n_steps.cpu().numpy()
        print(model + " n=%d %s B=%d: %.1f ms  steps mean %.0f max %d  flagged %d  -> %.1f us per (max) step" % (n, dist, B, dt * 1e3, ns[:, 0].mean(), ns[:, 0].max(), int((r.status != 0).sum()), dt * 1e6 / ns.sum(axis=1).max()), flush=True)
