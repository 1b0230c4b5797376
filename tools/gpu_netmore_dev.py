"""Dev: the default integrator at the optimiser's tolerance against every reference-run 1e-12 trajectory of tests/golden/netlarge_more_m*.npz."""
import sys
sys.path.insert(0, ".")
from pathlib import Path
import numpy as np
from phoskintime_amd.global_model import NetworkEngine

G = Path("tests/golden")
band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
for m in (0, 1, 2, 4):
    g, q = np.load(G / f"netlarge_m{m}.npz"), np.load(G / f"netlarge_more_m{m}.npz")
    eng = NetworkEngine.from_npz(g)
    K = int(q["done"])
    X = np.stack([eng.pack_params(q["c_k"][k], q["A_i"][k], q["B_i"][k], q["C_i"][k], q["D_i"][k], q["Dp_i"][k], q["E_i"][k], q["tf_scale"][k]) for k in range(K)])
    for label, kw in (("default", {}), ("rosw", dict(method="rosw")), ("rosw_rms", dict(method="rosw", err_norm="rms")), ("default_1e-10", dict(rtol=1e-10, atol=1e-10))):
        opt = dict(rtol=1e-8, atol=1e-8); opt.update(kw)
        Y, st, ns = eng.simulate_batch(X, q["t_eval"], **opt)
        Y = Y.cpu().numpy()
        print(m, label, "status", st.cpu().numpy().tolist(), "bands", [round(band(Y[k], q["Y_tight"][k]), 3) for k in range(K)], "steps", ns.cpu().numpy()[:, 0].tolist(), flush=True)
    print(m, "reference LSODA 1e-8 on candidate 0:", round(band(g["Y_lsoda8"][1], q["Y_tight"][0]), 3), "tf_scale", np.round(q["tf_scale"], 3).tolist(), flush=True)
    eng.close()
