#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  Nothing of the reference
is copied into the repo: the outputs are numbers (.npz).  Recipe:

  * a writable temp copy of the reference tree is used as CWD / sys.path root, because
    config/constants.py:141-143 creates directories at import time and the mount is read-only;
  * two import shims live in a temp dir: an identity-decorator ``numba`` (numba is not installed
    here; the RHS then runs as strict-IEEE pure Python) and ``tomllib`` -> ``tomli``;
  * models.{distmod,succmod,randmod}.solve_ode is called VERBATIM (SciPy odeint defaults) for
    ``sol_default`` / ``flat_default``; ``sol_tight`` is the same SciPy odeint driving the
    reference's own RHS at rtol = atol = 1e-13, mxstep = 500000;
  * config.config.score_fit is the reference's (imported); sensitivity.analysis._compute_Y cannot
    be imported here (SALib / seaborn absent) and is therefore NOT in the fixtures.

Usage:  python tools/make_golden.py            (rewrites tests/golden/*.npz)
"""
import os, sys, shutil, tempfile, types, importlib, pathlib
import numpy as np

REPO = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path("/root/reference")
OUT = REPO / "tests" / "golden"

TIME_POINTS = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])


def import_reference(prepare=None):
    """``prepare(tree)``: optional hook on the writable temp copy before anything is imported (config / data files of a test case)."""
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="pk_ref_"))
    tree = tmp / "ref"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns(".git", "docs", "static", "app", "background"))
    shim = tmp / "shim"
    (shim / "numba").mkdir(parents=True)
    (shim / "numba" / "__init__.py").write_text(
        "def _ident(*a, **k):\n"
        "    if len(a) == 1 and callable(a[0]) and not k:\n"
        "        return a[0]\n"
        "    return lambda f: f\n"
        "njit = jit = vectorize = _ident\n"
        "prange = range\n")
    (shim / "tomllib.py").write_text("from tomli import *\nfrom tomli import load, loads\n")
    if prepare is not None:
        prepare(tree)
    os.chdir(tree)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, str(shim))
    sys.path.insert(0, str(tree))
    mods = {}
    for name in ("distmod", "succmod", "randmod"):
        mods[name] = importlib.import_module(f"models.{name}")
    cfg = importlib.import_module("config.config")
    return mods, cfg, tmp


def main():
    from scipy.integrate import odeint
    mods, cfg, tmp = import_reference()
    OUT.mkdir(parents=True, exist_ok=True)

    def ref_rhs(name, y, theta, n):
        m = mods[name]
        if name == "randmod":
            A, B, C, D, S, Dd = m.unpack_params(theta, n)
            return np.asarray(m.ode_system(np.asarray(y, float), 0.0, A, B, C, D, n, S, Dd, *m._precompute_indices(n)))
        A, B, C, D, S, Dd = m.unpack_params(theta, n)
        return np.asarray(m.ode_core(np.asarray(y, float), 0.0, A, B, C, D, S, Dd))

    def ref_tight(name, theta, y0, n, t):
        m = mods[name]
        A, B, C, D, S, Dd = m.unpack_params(theta, n)
        if name == "randmod":
            args = (A, B, C, D, n, S, Dd, *m._precompute_indices(n)); f = m.ode_system
        else:
            args = (A, B, C, D, S, Dd); f = m.ode_core
        return np.asarray(odeint(f, y0, t, args=args, rtol=1e-13, atol=1e-13, mxstep=500000))

    def n_states(name, n):
        return 2 + (n if name != "randmod" else (1 << n) - 1)

    def n_params(name, n):
        return 4 + n + (n if name != "randmod" else (1 << n) - 1)

    def make_case_set(name, n, thetas, y0s, tag, tight=True):
        S = n_states(name, n)
        K = len(thetas)
        sol_d = np.empty((K, TIME_POINTS.size, S)); sol_t = np.empty_like(sol_d)
        flats = []; rhs0 = np.empty((K, S)); jac = np.empty((K, S, S)); bvec = np.empty((K, S))
        score = np.empty(K); rhs_rand_y = np.empty((K, S)); y_rand = np.empty((K, S))
        rng = np.random.default_rng(12345 + n)
        for k in range(K):
            th, y0 = thetas[k], y0s[k]
            sol, flat = mods[name].solve_ode(th, y0, n, TIME_POINTS)
            sol_d[k] = sol; flats.append(flat)
            if tight:
                sol_t[k] = ref_tight(name, th, y0, n, TIME_POINTS)
            rhs0[k] = ref_rhs(name, y0, th, n)
            f0 = ref_rhs(name, np.zeros(S), th, n)
            bvec[k] = f0
            for j in range(S):
                e = np.zeros(S); e[j] = 1.0
                jac[k][:, j] = ref_rhs(name, e, th, n) - f0      # exact: the reference RHS is affine in y
            y_rand[k] = rng.uniform(0, 3, S)
            rhs_rand_y[k] = ref_rhs(name, y_rand[k], th, n)
            target = np.abs(flat * (1 + 0.1 * rng.standard_normal(flat.size)))
            score[k] = cfg.score_fit(np.asarray(th, float), target, flat)
            if k == 0:
                tgt0 = target
        d = dict(model=name, n_sites=n, t=TIME_POINTS, theta=np.asarray(thetas), y0=np.asarray(y0s),
                 sol_default=sol_d, flat_default=np.asarray(flats), rhs_y0=rhs0, jac=jac, forcing=bvec,
                 y_rand=y_rand, rhs_y_rand=rhs_rand_y, score_fit=score, score_target0=tgt0)
        if tight:
            d["sol_tight"] = sol_t
        np.savez_compressed(OUT / f"protein_{name}_n{n}_{tag}.npz", **d)
        print("wrote", name, n, tag, "K =", K, flush=True)

    plan = {"distmod": [1, 2, 4, 8, 30], "succmod": [1, 2, 4, 8, 14], "randmod": [1, 2, 3, 4, 5, 6]}
    only = None
    if len(sys.argv) == 3:                      # `make_golden.py randmod 6`: (re)generate the two case sets of one model / size only
        only = (sys.argv[1], int(sys.argv[2]))
        plan = {only[0]: [only[1]]}
    for name, ns in plan.items():
        for n in ns:
            P, S = n_params(name, n), n_states(name, n)
            rng = np.random.default_rng(1000 * n + len(name))
            K = 6 if S > 20 else 8
            # (i) config bounds U(0, 20) (config.toml:189-195); randmod also in log-space like normest.py:367-369
            th_b = [rng.uniform(0, 20, P) for _ in range(K)]
            if name == "randmod":
                th_b[-2:] = [np.exp(rng.uniform(np.log(1e-8), np.log(20.0), P)) for _ in range(2)]
            y0_b = [np.ones(S) for _ in range(K)]
            y0_b[-1] = rng.uniform(0.1, 2.0, S)
            make_case_set(name, n, th_b, y0_b, "bounds")
            # (ii) "realistic" U(0.05, 2)
            th_r = [rng.uniform(0.05, 2.0, P) for _ in range(K)]
            make_case_set(name, n, th_r, [np.ones(S) for _ in range(K)], "real")
        if only:
            continue
        # (iii) edge cases at n = 4: knock-outs (knockout/helper.py:20-36), params on the bounds 0 and 20
        n = 4; P, S = n_params(name, n), n_states(name, n)
        rng = np.random.default_rng(77)
        base = rng.uniform(0.2, 5.0, P)
        edge = []
        for ko in ("A", "C", "Sall", "S1", "AC"):
            th = base.copy()
            if "A" in ko and ko != "Sall": th[0] = 0.0
            if "C" in ko: th[2] = 0.0
            if ko == "Sall": th[4:4 + n] = 0.0
            if ko == "S1": th[5] = 0.0
            edge.append(th)
        edge.append(np.zeros(P)); edge.append(np.full(P, 20.0))
        lo = base.copy(); lo[1] = 0.0; lo[3] = 0.0; edge.append(lo)        # no degradation at all
        make_case_set(name, n, edge, [np.ones(S) for _ in edge], "edge")

    # (iv) bench-shaped subsamples: first 64 replicas of the C2 / C3 synthetic batches
    for name, n, cfg_idx, lo_hi, tag in () if only else (("succmod", 14, 1, (0.0, 20.0), "c2bounds"), ("succmod", 14, 1, (0.05, 2.0), "c2benign"),
                                         ("distmod", 30, 2, (0.0, 20.0), "c3bounds"), ("distmod", 30, 2, (0.05, 2.0), "c3benign")):
        P, S = n_params(name, n), n_states(name, n)
        seed = 20260515 + cfg_idx + (1000 if "benign" in tag else 0)
        rng = np.random.default_rng(seed)
        theta = rng.uniform(lo_hi[0], lo_hi[1], (64, P))
        make_case_set(name, n, list(theta), [np.ones(S)] * 64, tag)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
