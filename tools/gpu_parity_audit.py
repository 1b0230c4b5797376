"""Dev tool: large randomised parity audit of the per-protein solve against the oracle's integrator-free solution (matrix exponentials).
For every model size and three parameter distributions: B replicas on the GPU (default options), every one compared with
oracle.protein_models.solve_exact_lti on the host cores.  Prints one line per case: worst band error (rtol 1e-6 / atol 1e-8), the share of
replicas beyond 0.5 and 1.0 band widths, flagged replicas, mean steps.  Usage: python tools/gpu_parity_audit.py [B] [part]  (part 0 / 1 / 2)"""
import sys, time, pathlib, zlib
import multiprocessing as mp
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import protein_models as pm

T = pm.TIME_POINTS


def _ref(args):
    mid, th, y0, n = args
    return np.clip(pm.solve_exact_lti(mid, th, y0, n, T), 0.0, None)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    part = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    cases = [("distmod", n) for n in (1, 4, 14, 30, 62)] + [("succmod", n) for n in (1, 4, 14, 30, 62)] + [("randmod", n) for n in (1, 2, 3, 4, 5, 6)]
    if part >= 0:
        cases = cases[part::3]
    if len(sys.argv) > 3:                                   # explicit list: model:n,model:n,...
        cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[3].split(",")]
    dists = {"U(0.05,2)": lambda r, s: r.uniform(0.05, 2.0, s), "U(0,20)": lambda r, s: r.uniform(0.0, 20.0, s),
             "logU(1e-3,1e2)": lambda r, s: 10.0 ** r.uniform(-3.0, 2.0, s)}
    logf = open(pathlib.Path(__file__).resolve().parents[1] / "gpurun_out" / ("parity_audit_%d.log" % max(part, 0)), "a")
    pool = mp.get_context("spawn").Pool(16)          # spawned before the first GPU call of this process
    import torch
    from phoskintime_amd import batch
    for model, n in cases:
        mid = pm.MODEL_IDS[model]; S, P = pm.n_states(mid, n), pm.n_params(mid, n)
        for dname, draw in dists.items():
            rng = np.random.default_rng(zlib.crc32(('%s %d %s' % (model, n, dname)).encode()))
            Bc = B if S <= 33 else max(64, B // 4)
            th = draw(rng, (Bc, P)); y0 = np.ones(S) if dname != "U(0.05,2)" else rng.uniform(0.2, 3.0, S)
            t0 = time.perf_counter()
            res = batch.solve_ode_batch(model, th, y0, n, T)
            sol = res.sol.cpu().numpy(); st = res.status.cpu().numpy(); ns = res.n_steps.cpu().numpy()
            refs = pool.map(_ref, [(mid, th[b], y0, n) for b in range(Bc)], chunksize=8)
            err = np.array([np.max(np.abs(sol[b] - refs[b]) / (1e-8 + 1e-6 * np.abs(refs[b]))) if st[b] == 0 else np.nan for b in range(Bc)])
            ok = np.isfinite(err)
            line = "%-8s n=%-3d S=%-3d %-15s B=%-5d worst %.3f  p99 %.3f  >0.5: %.2f%%  >1: %.2f%%  flagged %d  steps %.0f  (%.0f s)" % (
                model, n, S, dname, Bc, np.nanmax(err) if ok.any() else float("nan"), np.nanpercentile(err, 99) if ok.any() else float("nan"),
                100.0 * np.mean(err[ok] > 0.5), 100.0 * np.mean(err[ok] > 1.0), int((st != 0).sum()), ns[:, 0].mean(), time.perf_counter() - t0)
            print(line, flush=True); logf.write(line + "\n"); logf.flush()
    pool.close()


if __name__ == "__main__":
    main()
