"""Dev tool: the straggler replicas of the wide randmod kernel under theta ~ U(0, 20): step counts under both methods, parameter features."""
import os, sys, time, pathlib, subprocess, json
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import protein_models as pm
n = 7
S, P = pm.n_states(2, n), pm.n_params(2, n)
rng = np.random.default_rng(20260515)
th_all = rng.uniform(0.0, 20.0, (1024, P))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from phoskintime_amd import batch
    r = batch.solve_ode_batch("randmod", th_all, np.ones(S), n, pm.TIME_POINTS, want_flat=False)
    ns = r.n_steps.cpu().numpy(); sol = r.sol.cpu().numpy()
    np.save(sys.argv[2], ns); np.save(sys.argv[2] + ".sol.npy", sol)
    sys.exit(0)
out = {}
for tag, env in (("ark", {}), ("rosw", {"PK_WIDE_RAND_ROSW": "1"})):
    f = "/tmp/ns_%s.npy" % tag
    subprocess.check_call([sys.executable, __file__, "child", f], env={**os.environ, **env})
    out[tag] = np.load(f); out[tag + "_sol"] = np.load(f + ".sol.npy")
a, r = out["ark"], out["rosw"]
print("ARK  steps: mean %.0f  p99 %.0f  max %d ; rejected mean %.0f max %d" % (a[:, 0].mean(), np.percentile(a[:, 0], 99), a[:, 0].max(), a[:, 1].mean(), a[:, 1].max()))
print("ROSW steps: mean %.0f  p99 %.0f  max %d ; rejected mean %.0f max %d" % (r[:, 0].mean(), np.percentile(r[:, 0], 99), r[:, 0].max(), r[:, 1].mean(), r[:, 1].max()))
worst = np.argsort(-a[:, 0])[:6]
for b in worst:
    th = th_all[b]
    ref = pm.solve_exact_lti(2, th, np.ones(S), n, pm.TIME_POINTS)
    ea = pm.band_error(out["ark_sol"][b], np.clip(ref, 0, None)); er = pm.band_error(out["rosw_sol"][b], np.clip(ref, 0, None))
    print("replica %d: ARK acc %d rej %d (band %.3f) | ROSW acc %d rej %d (band %.3f) | A %.2f B %.3f C %.2f D %.3f  S %s  min Ddeg %.3f" % (
        b, a[b, 0], a[b, 1], ea, r[b, 0], r[b, 1], er, th[0], th[1], th[2], th[3], np.round(th[4:4 + n], 2), th[4 + n:].min()))
