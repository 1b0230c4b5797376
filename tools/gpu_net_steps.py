"""Dev: where does the network integrator spend its steps?  Cumulative accepted steps up to each output time (prefix runs)."""
import sys, numpy as np
sys.path.insert(0, ".")
from phoskintime_amd.global_model import NetworkEngine
g = np.load("tests/golden/netlarge_m0.npz"); eng = NetworkEngine.from_npz(g)
X = np.stack([eng.pack_params(g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], g["tf_scale"][k]) for k in range(2)])
t = g["t_eval"]
for rtol, atol in ((1e-8, 1e-8), (1e-7, 1e-9)):
    prev = np.zeros(2)
    for k in range(1, t.size):
        Y, st, ns = eng.simulate_batch(X, t[:k + 1], rtol=rtol, atol=atol)
        cum = ns[:, 0].cpu().numpy().astype(float)
        print(f"rtol {rtol:g}: ({t[k-1]:7.2f}, {t[k]:7.2f}]  steps {cum - prev}  mean h {(t[k]-t[k-1])/np.maximum(cum-prev,1)}", flush=True)
        prev = cum
