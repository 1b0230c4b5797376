"""Dev tool: randomised audit of the forward sensitivities (pk_solve_protein_sens_batch) against central differences of the oracle's
integrator-free solution, over every size with a kernel and two parameter distributions.  One line per case: worst
|d flat / d theta - oracle| / (1 + |oracle|) over the replicas, flagged replicas, steps.  Usage: python tools/gpu_sens_audit.py [replicas]"""
import sys, time, pathlib, zlib
import multiprocessing as mp
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import protein_models as pm

T = pm.TIME_POINTS


def _flat(mid, th, y0, n):
    return pm.flatten_observables(mid, np.clip(pm.solve_exact_lti(mid, th, y0, n, T), 0.0, None), n)


def _jac(args):
    mid, th, y0, n = args
    cols = []
    for c in range(th.size):
        # five-point stencil: a two-point difference has a truncation error (h t)^2 / 6 on parameters that set a time scale (t up to 960),
        # which at h = 1e-5 is 1e-5 ... 1e-4 of the derivative -- above what is being measured here
        h = 1e-3 * max(1e-1, min(1.0, abs(th[c])))             # (smaller steps measure the rounding noise of the matrix exponentials: 1e-5 at 2e-5)
        f = lambda s: _flat(mid, np.where(np.arange(th.size) == c, th + s * h, th), y0, n)
        cols.append((f(-2.0) - 8.0 * f(-1.0) + 8.0 * f(1.0) - f(2.0)) / (12.0 * h))
    return np.stack(cols, axis=1)


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cases = [("distmod", n) for n in (1, 2, 3, 5, 8, 11, 14)] + [("succmod", n) for n in (1, 3, 6, 10, 14)] + [("randmod", n) for n in (1, 2, 3, 4, 5)]
    if len(sys.argv) > 2:
        cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[2].split(",")]
    dists = {"U(0.05,3)": lambda r, s: r.uniform(0.05, 3.0, s), "logU(1e-2,20)": lambda r, s: 10.0 ** r.uniform(-2.0, np.log10(20.0), s)}
    logf = open(pathlib.Path(__file__).resolve().parents[1] / "gpurun_out" / "sens_audit.log", "a")
    pool = mp.get_context("spawn").Pool(16)
    import torch
    from phoskintime_amd import batch
    for model, n in cases:
        mid = pm.MODEL_IDS[model]; S, P = pm.n_states(mid, n), pm.n_params(mid, n)
        for dname, draw in dists.items():
            rng = np.random.default_rng(zlib.crc32(('%s %d %s' % (model, n, dname)).encode()))
            th = draw(rng, (R, P)); y0 = rng.uniform(0.3, 2.0, S)
            t0 = time.perf_counter()
            r = batch.solve_ode_sens_batch(model, th, y0, n, T, rtol=1e-9, atol=1e-11)
            d = r.dflat.cpu().numpy(); st = r.status.cpu().numpy(); ns = r.n_steps.cpu().numpy()
            refs = pool.map(_jac, [(mid, th[b], y0, n) for b in range(R)], chunksize=1)
            err = np.array([np.max(np.abs(d[b] - refs[b]) / (1.0 + np.abs(refs[b]))) for b in range(R)])
            # the same differences against the scale of their COLUMN (what a least-squares step sees): max_f |diff| / (1 + max_f |d flat_f / d theta_c|)
            errc = np.array([np.max(np.abs(d[b] - refs[b]).max(axis=0) / (1.0 + np.abs(refs[b]).max(axis=0))) for b in range(R)])
            line = "%-8s n=%-2d P=%-2d %-14s replicas %d: worst entrywise %.2e (median %.2e)  worst columnwise %.2e  max|d| %.2e  flagged %d  steps %.0f  (%.0f s)" % (
                model, n, P, dname, R, err.max(), np.median(err), errc.max(), max(np.abs(x).max() for x in refs), int((st != 0).sum()), ns[:, 0].mean(), time.perf_counter() - t0)
            print(line, flush=True); logf.write(line + "\n"); logf.flush()
    pool.close()


if __name__ == "__main__":
    main()
