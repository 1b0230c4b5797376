"""Dev: the staged (line-buffer) trajectory stores of the thread-per-replica kernel against the direct stores: equal bits, time per launch."""
import os, sys, time, pathlib, subprocess
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, str(ROOT))
    from phoskintime_amd import batch
    from oracle import protein_models as pm
    out = {}
    for mdl, n, B in (("distmod", 4, 524288), ("distmod", 3, 70001), ("succmod", 4, 100000), ("distmod", 1, 65537)):
        P, S = batch.n_params(mdl, n), batch.n_states(mdl, n)
        th = torch.as_tensor(np.random.default_rng(5).uniform(0.0, 20.0, (B, P)), device="cuda")
        if mdl == "distmod" and n == 3:
            th[7, 2] = float("nan")                                  # a failed replica: NaN rows through the same store path
        r = batch.solve_ode_batch(mdl, th, np.ones(S), n, pm.TIME_POINTS, want_flat=False, kernel="tpr")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            batch.solve_ode_batch(mdl, th, np.ones(S), n, pm.TIME_POINTS, want_flat=False, kernel="tpr", out=r)
        torch.cuda.synchronize()
        ms = 1e2 * (time.perf_counter() - t0)
        print(f"stage={os.environ.get('PK_TPR_STAGE', '1')} {mdl} n={n} B={B}: {ms:.3f} ms per launch = {B / ms / 1e3:.1f} M replicas/s", flush=True)
        out[f"{mdl}{n}"] = r.sol.cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
res = {}
for st in ("0", "1"):
    f = f"/tmp/tpr_stage_{st}.npz"
    subprocess.run([sys.executable, __file__, "child", f], check=True, env={**os.environ, "PK_TPR_STAGE": st})
    res[st] = np.load(f)
for k in res["0"].files:
    a, b = res["0"][k], res["1"][k]
    print(k, "bitwise equal:", bool(np.array_equal(a, b, equal_nan=True)), "nan rows:", int(np.isnan(a).any(axis=(1, 2)).sum()))
