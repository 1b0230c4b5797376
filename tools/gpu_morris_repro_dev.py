import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from phoskintime_amd.global_model import NetworkEngine
from phoskintime_amd.global_model.sensitivity import run_sensitivity_batch, scalar_metric_batch
from phoskintime_amd.global_model.simulate import measure_tolerances
g = np.load("tests/golden/pins_network_m0.npz")
eng = NetworkEngine.from_npz(g)
keys = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i", "tf_scale")
sl = {k: slice(int(a), int(b)) for k, (a, b) in zip(keys, g["slice_bounds"])}
row = g["X_phys"][2]
fitted = {k: (row[sl[k]] if k != "tf_scale" else float(row[sl[k]][0])) for k in keys}
out = run_sensitivity_batch(eng, fitted, g["tp"], g["tr"], g["tph"], trajectories=3, num_levels=8, seed=5)
times = np.unique(np.concatenate([g["tp"], g["tr"], g["tph"]]).astype(np.float64))
lists, ld_ = eng.make_index_lists(times, g["tp"], g["tr"], g["tph"])
X = out["param_values"]
tol = measure_tolerances(eng)
Ya, st_, na = eng.simulate_batch(X, times, max_steps=5000 * times.size, **tol)
pred = eng.observables_batch(lists, Ya, ld_["p_prot"].size + ld_["p_rna"].size + ld_["p_pho"].size, eps=1e-12)
one = scalar_metric_batch(pred, "total_signal").cpu().numpy()
print("world-1 driver vs direct: differing rows", int((one != out["Y"]).sum()), "of", one.size, "max", float(np.abs(one - out["Y"]).max()))
# halves
idx = np.arange(0, X.shape[0], 2)
Yb, sb, nb = eng.simulate_batch(X[idx], times, max_steps=5000 * times.size, **tol)
d = (Ya[idx] - Yb).abs()
print("subset vs all: differing entries", int((d != 0).sum()), "max", float(d.max()), "steps equal", bool((na[idx] == nb).all()), "X min", float(X.min()), "zeros in X", int((X == 0).sum()))
bad = torch.nonzero(d.reshape(d.shape[0], -1).max(dim=1).values > 0).flatten().tolist()
print("candidates that differ:", bad[:10], "steps", na[idx][bad[:10]].tolist(), nb[bad[:10]].tolist())
Yc, sc, nc = eng.simulate_batch(X, times, max_steps=5000 * times.size, **tol)
print("rerun identical", bool((Ya == Yc).all()))
