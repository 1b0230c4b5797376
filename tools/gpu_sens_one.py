#!/usr/bin/env python3
"""Dev tool: run ONE forward-sensitivity configuration a few times (for rocprofv3 passes): gpu_sens_one.py MODEL N_SITES B [iters].
Prints one JSON line (ms per launch, Jacobians per second)."""
import json, pathlib, sys, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch
from oracle import protein_models as pm
model, n, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
mid = pm.MODEL_IDS[model]; S, P = pm.n_states(mid, n), pm.n_params(mid, n)
th = torch.as_tensor(np.random.default_rng(1).uniform(0.2, 2.0, size=(B, P)), device="cuda")
y0 = np.ones(S)
batch.solve_ode_sens_batch(model, th[:8], y0, n, pm.TIME_POINTS); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    r = batch.solve_ode_sens_batch(model, th, y0, n, pm.TIME_POINTS)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(json.dumps({"workload": "solve_ode_sens_batch %s n=%d B=%d (P=%d columns + 1)" % (model, n, B, P), "ms_per_launch": 1e3 * dt, "jacobians_per_s": B / dt,
                  "mean_steps": float(r.n_steps[:, 0].double().mean()), "flagged": int((r.status != 0).sum())}))
