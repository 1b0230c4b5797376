"""Dev: steps / band error of the network integrator under the max norm and the RMS norm (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, ".")
from pathlib import Path
from phoskintime_amd.global_model import NetworkEngine
band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
for f in sorted(Path("tests/golden").glob("netlarge_m[0-9].npz")) + sorted(Path("tests/golden").glob("network_m*_medium.npz")) + sorted(Path("tests/golden").glob("network_m*_small.npz")):
    g = np.load(f); eng = NetworkEngine.from_npz(g)
    X = np.stack([eng.pack_params(g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], g["tf_scale"][k]) for k in range(g["Y_tight"].shape[0])])
    out = [f.name]
    for rtol, atol in ((1e-8, 1e-8), (1e-7, 1e-9), (1e-5, 1e-7)):
        for norm, meth in (("max", "rosw"), ("rms", "rosw"), ("max", "auto")):
            Y, st, ns = eng.simulate_batch(X, g["t_eval"], rtol=rtol, atol=atol, err_norm=norm, method=meth)
            b = max(band(Y[k].cpu().numpy(), g["Y_tight"][k]) for k in range(X.shape[0]))
            out.append(f"{rtol:g}/{atol:g} {meth}/{norm}: {int(ns[:, 0].double().mean())}+{int(ns[:, 1].double().mean())} st band {b:.3f} fl {int((st != 0).sum())}")
    ref = max(band(g["Y_lsoda8"][k], g["Y_tight"][k]) for k in range(g["Y_tight"].shape[0]))
    print(" | ".join(out), f"| ref LSODA 1e-8: {ref:.3f}", flush=True)
    eng.close()
