"""Dev: the order-4 network integrator on the OPTIMISER's kind of population -- raw decision vectors uniform in the raw bounds of
params.init_raw_params (config.toml [global_model.bounds]) -- against the order-3 method run two decades tighter."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from phoskintime_amd.global_model import NetworkEngine, params as gp
for fn in ("tests/golden/netlarge_m0.npz", "tests/golden/netlarge_m1.npz", "tests/golden/netlarge_m4.npz"):
    g = np.load(fn); eng = NetworkEngine.from_npz(g)
    defaults = dict(c_k=g["c_k"][0], A_i=g["A_i"][0], B_i=g["B_i"][0], C_i=g["C_i"][0], D_i=g["D_i"][0], Dp_i=g["Dp_i"][0], E_i=g["E_i"][0], tf_scale=float(g["tf_scale"][0]))
    theta0, slices, xl, xu = gp.init_raw_params(defaults)
    rng = np.random.default_rng(7)
    B = 2048
    X = rng.uniform(xl, xu, (B, xl.size))
    t = g["t_eval"]
    Ya, sa, na = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8)
    Yr, sr, nr = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8, method="rosw")
    Yt, stt, nt = eng.simulate_batch(X, t, raw=True, rtol=1e-10, atol=1e-10, method="rosw")
    band = lambda a, b: ((a - b).abs() / (1e-8 + 1e-6 * b.abs())).amax(dim=(1, 2))
    ba, br = band(Ya, Yt).cpu().numpy(), band(Yr, Yt).cpu().numpy()
    print(fn, "flagged ark/rosw/tight", int((sa != 0).sum()), int((sr != 0).sum()), int((stt != 0).sum()),
          "| steps ark %.0f rosw %.0f | band vs tight: ark max %.3f p99 %.3f median %.3f ; rosw max %.3f" % (
              na[:, 0].double().mean(), nr[:, 0].double().mean(), np.nanmax(ba), np.nanpercentile(ba, 99), np.nanmedian(ba), np.nanmax(br)), flush=True)
    eng.close()

# worst candidate of the first network: save it (and the three GPU trajectories) for an LSODA 1e-12 run on the CPU oracle
g = np.load("tests/golden/netlarge_m0.npz"); eng = NetworkEngine.from_npz(g)
defaults = dict(c_k=g["c_k"][0], A_i=g["A_i"][0], B_i=g["B_i"][0], C_i=g["C_i"][0], D_i=g["D_i"][0], Dp_i=g["Dp_i"][0], E_i=g["E_i"][0], tf_scale=float(g["tf_scale"][0]))
theta0, slices, xl, xu = gp.init_raw_params(defaults)
X = np.random.default_rng(7).uniform(xl, xu, (2048, xl.size)); t = g["t_eval"]
Ya, _, na = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8)
Yr, _, nr = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8, method="rosw")
Yt, _, nt = eng.simulate_batch(X, t, raw=True, rtol=1e-10, atol=1e-10, method="rosw")
Yt2, _, nt2 = eng.simulate_batch(X, t, raw=True, rtol=1e-11, atol=1e-11, method="ark")
band = lambda a, b: ((a - b).abs() / (1e-8 + 1e-6 * b.abs())).amax(dim=(1, 2))
ba = band(Ya, Yt).cpu().numpy()
w = int(np.nanargmax(ba))
print("worst candidate", w, "ark-vs-rosw1e-10", ba[w], "rosw1e-8 vs rosw1e-10", float(band(Yr, Yt)[w]), "ark1e-11 vs rosw1e-10", float(band(Yt2, Yt)[w]),
      "steps ark/rosw/tight/arktight", int(na[w, 0]), int(nr[w, 0]), int(nt[w, 0]), int(nt2[w, 0]))
print("population: ark1e-11 vs rosw1e-10 max", float(band(Yt2, Yt).max()), " ark1e-8 vs ark1e-11 max", float(band(Ya, Yt2).max()), "rosw1e-8 vs ark1e-11 max", float(band(Yr, Yt2).max()))
np.savez("gpurun_out/ark_worst.npz", x_raw=X[w], Ya=Ya[w].cpu().numpy(), Yr=Yr[w].cpu().numpy(), Yt=Yt[w].cpu().numpy(), Yt2=Yt2[w].cpu().numpy(), idx=w)
