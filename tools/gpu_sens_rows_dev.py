"""Dev: the rows-per-lane sensitivity kernel (csrc/pk_sens_rows.hpp) against differencing on the throughput kernels, and the LM leg."""
import sys, time, json, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from phoskintime_amd import batch
TG = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
dev = torch.device("cuda", 0)
tt = torch.as_tensor(TG, device=dev)


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


for mdl, n in (("distmod", 30), ("succmod", 30), ("distmod", 62), ("succmod", 62), ("distmod", 16)):
    P, S = batch.n_params(mdl, n), batch.n_states(mdl, n)
    for B in (48, 480, 4096):
        th = torch.as_tensor(np.random.default_rng(1).uniform(0.2, 2.0, (B, P)), device=dev)
        y0 = np.ones(S)
        ms_s = timeit(lambda: batch.solve_ode_sens_batch(mdl, th, y0, n, tt))
        r = batch.solve_ode_sens_batch(mdl, th, y0, n, tt)
        thf = th.repeat_interleave(1 + P, dim=0)
        ms_f = timeit(lambda: batch.solve_ode_batch(mdl, thf, y0, n, tt, want_sol=False))
        print(f"{mdl} n={n} B={B}: sens {ms_s:.3f} ms ({r.n_steps[:, 0].double().mean().item():.1f} steps, flagged {int((r.status != 0).sum())}) vs differencing {B * (1 + P)} replicas {ms_f:.3f} ms", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "lm":
    import bench
    print(json.dumps(bench.lm_leg()["lambda_scan_480_rows_distmod_n30"], indent=1))
