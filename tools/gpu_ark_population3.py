"""Dev: tail of the order-4 integrator's error on the uniform-in-bounds population under controller variants (env PK_ARK_*)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from phoskintime_amd.global_model import NetworkEngine, params as gp
g = np.load("tests/golden/netlarge_m0.npz"); eng = NetworkEngine.from_npz(g)
defaults = dict(c_k=g["c_k"][0], A_i=g["A_i"][0], B_i=g["B_i"][0], C_i=g["C_i"][0], D_i=g["D_i"][0], Dp_i=g["Dp_i"][0], E_i=g["E_i"][0], tf_scale=float(g["tf_scale"][0]))
theta0, slices, xl, xu = gp.init_raw_params(defaults)
X = np.random.default_rng(7).uniform(xl, xu, (2048, xl.size)); t = g["t_eval"]
Yt, _, _ = eng.simulate_batch(X, t, raw=True, rtol=1e-10, atol=1e-10, method="rosw")
band = lambda a: ((a - Yt).abs() / (1e-8 + 1e-6 * Yt.abs())).amax(dim=(1, 2)).cpu().numpy()
eng.simulate_batch(X[:64], t, raw=True, rtol=1e-8, atol=1e-8); torch.cuda.synchronize()
t0 = time.perf_counter(); Y, st, ns = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8); torch.cuda.synchronize(); dt = time.perf_counter() - t0
b = band(Y)
print("ark 1e-8: %.1f ms, steps %.0f rej %.0f | band median %.3f p90 %.3f p99 %.3f max %.2f | frac > 1: %.4f" % (dt * 1e3, ns[:, 0].double().mean(), ns[:, 1].double().mean(), np.median(b), np.percentile(b, 90), np.percentile(b, 99), b.max(), (b > 1).mean()), flush=True)
