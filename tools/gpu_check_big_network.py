#!/usr/bin/env python3
"""Dev check: a network beyond the register kernel's limits (N = 300 > 256 proteins, S = 1000) must run through the LDS kernel and agree
with the explicit DP5 path (which only needs right-hand sides) at tight tolerance."""
import pathlib, sys, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd.global_model import NetworkEngine, synthetic

for model in (0, 1, 4):
    net = synthetic.make_network(N=300, total_sites=400, n_K=60, n_tf_edges=700, model=model, seed=77)
    eng = NetworkEngine(**net)
    X = synthetic.random_candidates(net, 64, seed=2)
    t = np.unique(np.concatenate([net["kin_grid"], [15.0]]))
    t0 = time.perf_counter(); Y, st, ns = eng.simulate_batch(X, t, rtol=1e-7, atol=1e-9); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    Yd, std, nsd = eng.simulate_batch(X[:4], t, rtol=1e-9, atol=1e-11, max_steps=2_000_000, method="dp5")
    a, b = Y[:4].cpu().numpy(), Yd.cpu().numpy()
    band = np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b)))
    print("model %d N %d S %d: W-method (LDS kernel) %.1f ms for 64 candidates, steps %.0f, flagged %d; band vs DP5@1e-9 = %.4f (DP5 flagged %d)" % (
        model, eng.N, eng.S, dt * 1e3, ns[:, 0].double().mean().item(), int((st != 0).sum()), band, int((std != 0).sum())), flush=True)
    eng.close()
