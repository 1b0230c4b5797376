"""Dev: 8 192 candidates (log-normal 0.5 around the fixture's parameter set 0) of netlarge_m<M>.npz at 1e-8 / 1e-8, default integrator.
    M=1 python tools/gpu_net_topo_time.py"""
import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from phoskintime_amd.global_model import NetworkEngine
for M in [int(v) for v in os.environ.get("M", "0,1,2,4").split(",")]:
    g = np.load(f"tests/golden/netlarge_m{M}.npz")
    eng = NetworkEngine.from_npz(g)
    base = eng.pack_params(g["c_k"][0], g["A_i"][0], g["B_i"][0], g["C_i"][0], g["D_i"][0], g["Dp_i"][0], g["E_i"][0], g["tf_scale"][0])
    rng = np.random.default_rng(20260515 + 4)
    X = base[None, :] * np.exp(0.5 * rng.standard_normal((8192, base.size))); X[0] = base
    Xd = torch.as_tensor(np.log(np.expm1(np.maximum(X, 1e-12))), device="cuda")
    tn = g["t_eval"]; kw = dict(raw=True, rtol=1e-8, atol=1e-8)
    eng.simulate_batch(Xd[:256], tn, **kw); torch.cuda.synchronize()
    best = 1e9
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); Y, st, ns = eng.simulate_batch(Xd, tn, **kw); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    band = float(np.max(np.abs(Y[0].cpu().numpy() - g["Y_tight"][0]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][0]))))
    print(os.environ.get("TAG", ""), "model", M, "S", eng.S, "best_ms %.2f" % best, "k cand/s %.1f" % (8192 / best), "steps", [round(v, 2) for v in ns.double().mean(dim=0).tolist()],
          "flagged", int((st != 0).sum()), "band0 %.3f" % band, flush=True)
    eng.close()
