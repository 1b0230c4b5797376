"""Dev: where the wall time of BASELINE config 4 (128 trajectories x 200 varied parameters, N = 100 network) goes: cProfile of run_sensitivity_batch."""
import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, torch
from phoskintime_amd.global_model import NetworkEngine
from phoskintime_amd.global_model import sensitivity as gs, config as gcfg
g = np.load("tests/golden/netlarge_m0.npz"); eng = NetworkEngine.from_npz(g)
keys = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")
fitted = {k: g[k][0] for k in keys}; fitted["tf_scale"] = float(g["tf_scale"][0])
vary = np.random.default_rng(4).choice(eng.n_var, 200, replace=False)
kw = dict(trajectories=128, num_levels=4, seed=7, vary=vary)
tp, tr, tph = gcfg.TIME_POINTS_PROTEIN, gcfg.TIME_POINTS_RNA, gcfg.TIME_POINTS_PHOSPHO
gs.run_sensitivity_batch(eng, fitted, tp, tr, tph, **kw); torch.cuda.synchronize()
t = time.perf_counter(); out = gs.run_sensitivity_batch(eng, fitted, tp, tr, tph, **kw); torch.cuda.synchronize(); print("wall ms", 1e3 * (time.perf_counter() - t), "steps", out["mean_steps"])
pr = cProfile.Profile(); pr.enable(); gs.run_sensitivity_batch(eng, fitted, tp, tr, tph, **kw); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
