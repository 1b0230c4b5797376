"""Dev: is a candidate's trajectory independent of the batch it is integrated in (bit for bit)?  PK_ARK_PAIR=0/1/3 per process."""
import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from phoskintime_amd.global_model import NetworkEngine
for name in ("pins_network_m0", "netlarge_m0", "network_m4_small"):
    g = np.load(f"tests/golden/{name}.npz")
    eng = NetworkEngine.from_npz(g)
    rng = np.random.default_rng(1)
    if "c_k" in g.files:
        base = eng.pack_params(g["c_k"][0], g["A_i"][0], g["B_i"][0], g["C_i"][0], g["D_i"][0], g["Dp_i"][0], g["E_i"][0], g["tf_scale"][0])
    else:
        base = g["X_phys"][2]
    X = base[None, :] * np.exp(0.3 * rng.standard_normal((64, base.size)))
    t = np.array([0.0, 1.0, 4.0, 15.0, 60.0, 240.0, 960.0])
    Ya, sa, na = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8)
    Yb, sb, nb = eng.simulate_batch(X[::2], t, rtol=1e-8, atol=1e-8)
    Yc, sc, nc = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8)
    same_rerun = bool((Ya == Yc).all()); same_sub = bool((Ya[::2] == Yb).all())
    d = (Ya[::2] - Yb).abs().max().item()
    print(os.environ.get("PK_ARK_PAIR", "default"), name, "N", eng.N, "rerun identical", same_rerun, "subset identical", same_sub, "max diff", d,
          "steps equal", bool((na[::2] == nb).all()), flush=True)
    eng.close()
