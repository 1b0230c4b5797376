"""Dev: ARK436 on the combinatorial topology with the exact (parity-eliminated) block solve against round 2's approximate factorisation
(PK_ARK2_EXACT=0) and the order-3 default: band error vs the reference's LSODA@1e-12, steps, time for a config-5-shaped population."""
import sys, time, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from phoskintime_amd.global_model import NetworkEngine
for name in ("network_m2_small.npz", "network_m2_medium.npz", "netlarge_m2.npz"):
    f = ROOT / "tests" / "golden" / name
    if not f.exists():
        continue
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    K = g["Y_tight"].shape[0]
    X = np.stack([eng.pack_params(g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], g["tf_scale"][k]) for k in range(K)])
    for method in ("ark", "rosw"):
        Y, st, ns = eng.simulate_batch(X, g["t_eval"], rtol=1e-8, atol=1e-8, method=method)
        Yh = Y.cpu().numpy()
        band = max(float(np.max(np.abs(Yh[k] - g["Y_tight"][k]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][k])))) for k in range(K))
        print(f"{name} {method}: band {band:.3f} steps {ns[:, 0].cpu().numpy()} rejected {ns[:, 1].cpu().numpy()} flagged {int((st != 0).sum())}", flush=True)
    if name == "netlarge_m2.npz":
        rng = np.random.default_rng(4)
        Xp = X[0][None, :] * np.exp(0.5 * rng.standard_normal((8192, X.shape[1]))); Xp[0] = X[0]
        Xd = torch.as_tensor(Xp, device="cuda")
        for method in ("ark", "rosw"):
            eng.simulate_batch(Xd[:256], g["t_eval"], rtol=1e-8, atol=1e-8, method=method); torch.cuda.synchronize()
            t0 = time.perf_counter()
            Y, st, ns = eng.simulate_batch(Xd, g["t_eval"], rtol=1e-8, atol=1e-8, method=method)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"  population 8192 {method}: {1e3 * dt:.1f} ms = {8192 / dt:.0f} candidates/s, mean steps {ns[:, 0].double().mean().item():.0f}, flagged {int((st != 0).sum())}", flush=True)
            if method == "ark":
                Yark = Y
            else:
                print("  max band between the two integrators over the population:", float(((Y - Yark).abs() / (1e-8 + 1e-6 * Y.abs())).max()))
    eng.close()
