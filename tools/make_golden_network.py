#!/usr/bin/env python3
"""Golden vectors for the global_model NETWORK path, made by RUNNING THE REFERENCE (build container only).

One process per kinetic topology, because the reference fixes MODEL at import time (global_model/config.py:59-61):
    python tools/make_golden_network.py distributive|sequential|combinatorial|saturation
or  python tools/make_golden_network.py all     (spawns the four)

Same import recipe as tools/make_golden.py (writable temp copy, identity-numba + tomllib shims) plus: config.toml's
[global_model.models] default_model is patched in the temp copy, and a bare `global_model` package object is registered so that
global_model/__init__.py (which pulls in optuna / pymoo / SALib) is not executed.  The synthetic networks go through the reference's
own Index / KinaseInput / System / build_W_parallel / build_tf_matrix classes; only numbers are written (tests/golden/network_*.npz)."""
import os, sys, shutil, tempfile, types, importlib, pathlib, subprocess, re
import numpy as np

REPO = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path("/root/reference")
OUT = REPO / "tests" / "golden"
MODEL_ID = {"distributive": 0, "sequential": 1, "combinatorial": 2, "saturation": 4}


def import_reference(model_name):
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="pk_refnet_"))
    tree = tmp / "ref"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns(".git", "docs", "static", "app", "background"))
    cfg = (tree / "config.toml").read_text()
    cfg2 = re.sub(r'default_model\s*=\s*"[a-z]+"', f'default_model = "{model_name}"', cfg)
    assert cfg2 != cfg or model_name == "distributive"
    (tree / "config.toml").write_text(cfg2)
    shim = tmp / "shim"
    (shim / "numba").mkdir(parents=True)
    (shim / "numba" / "__init__.py").write_text(
        "def _ident(*a, **k):\n    if len(a) == 1 and callable(a[0]) and not k:\n        return a[0]\n    return lambda f: f\n"
        "njit = jit = vectorize = _ident\nprange = range\n")
    (shim / "tomllib.py").write_text("from tomli import *\nfrom tomli import load, loads\n")
    os.chdir(tree)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, str(shim)); sys.path.insert(0, str(tree))
    pkg = types.ModuleType("global_model"); pkg.__path__ = [str(tree / "global_model")]
    sys.modules["global_model"] = pkg
    mods = {n: importlib.import_module(f"global_model.{n}") for n in ("config", "models", "jacspeedup", "network", "buildmat", "simulate")}
    assert mods["config"].MODEL == MODEL_ID[model_name], (mods["config"].MODEL, model_name)
    return mods, tmp


def synth_frames(rng, N, max_sites, n_ext_kin, n_kin_prot, n_tf_edges):
    import pandas as pd
    prots = [f"P{i:03d}" for i in range(N)]
    # the last protein is an ORPHAN TF (no sites, regulates a kinase-protein -> proxy redirection); the one before has no sites at all
    kin_prots = list(rng.choice(prots[:N - 2], size=n_kin_prot, replace=False))
    kinases = kin_prots + [f"K{i:02d}" for i in range(n_ext_kin)]
    rows = []
    for p in prots[:N - 2]:
        ns = int(rng.integers(1, max_sites + 1))
        pos = sorted(rng.choice(np.arange(5, 900), size=ns, replace=False))
        for s in pos:
            for k in rng.choice(kinases, size=int(rng.integers(1, 3)), replace=False):
                rows.append(dict(protein=p, psite=f"S{int(s)}", kinase=str(k), alpha=float(rng.uniform(0.2, 1.0))))
    inter = pd.DataFrame(rows)
    tf_rows = []
    orphan = prots[N - 1]
    tf_rows.append(dict(tf=orphan, target=kin_prots[0], alpha=0.7))            # orphan -> kinase: proxy
    tf_rows.append(dict(tf=orphan, target=prots[1], alpha=-0.4))
    seen = {(orphan, kin_prots[0]), (orphan, prots[1])}
    while len(tf_rows) < n_tf_edges:
        a, b = rng.choice(prots[:N - 1], size=2, replace=False)
        if (a, b) in seen:
            continue
        seen.add((a, b))
        tf_rows.append(dict(tf=str(a), target=str(b), alpha=float(rng.uniform(-1, 1))))
    tf_net = pd.DataFrame(tf_rows)
    # kinase fold-change table on the protein grid
    return prots, kinases, inter, tf_net


def main(model_name):
    import pandas as pd
    mods, tmp = import_reference(model_name)
    cfg, net, bm, js, sim = mods["config"], mods["network"], mods["buildmat"], mods["jacspeedup"], mods["simulate"]
    MODEL = cfg.MODEL
    OUT.mkdir(parents=True, exist_ok=True)
    for tag, N, n_ext, n_kp, n_tf, seed in (("small", 6, 2, 2, 8, 11), ("medium", 20, 4, 5, 40, 12)):
        rng = np.random.default_rng(seed + 100 * MODEL)
        max_sites = 3
        prots, kinases, inter, tf_net = synth_frames(rng, N, max_sites, n_ext, n_kp, n_tf)
        idx = net.Index(inter, tf_interactions=tf_net, kin_beta_map={k: float(rng.uniform(0.5, 1.5)) for k in kinases}, tf_beta_map={})
        grid = np.asarray(cfg.TIME_POINTS_PROTEIN, float)
        fc_rows = []
        for k in idx.kinases:
            base = 1.0 + 0.5 * np.sin(rng.uniform(0, 6) + np.arange(grid.size) * rng.uniform(0.2, 0.8))
            for t, v in zip(grid, base):
                if rng.uniform() < 0.9:
                    fc_rows.append(dict(protein=k, time=float(t), fc=float(max(v, 1e-6))))
        kin_in = net.KinaseInput(idx.kinases, pd.DataFrame(fc_rows))
        W = bm.build_W_parallel(inter, idx, n_cores=1)
        tf_mat = bm.build_tf_matrix(tf_net, idx, tf_beta_map={}, kin_beta_map={})
        tf_deg = np.asarray(np.abs(tf_mat).sum(axis=1)).ravel().astype(np.float64)
        tf_deg[tf_deg < 1e-12] = 1.0
        K = 4
        psets = []
        for k in range(K):
            psets.append(dict(c_k=rng.uniform(0.3, 2.0, len(idx.kinases)), A_i=rng.uniform(0.3, 2.0, idx.N), B_i=rng.uniform(0.05, 1.0, idx.N),
                              C_i=rng.uniform(0.1, 2.0, idx.N), D_i=rng.uniform(0.01, 0.5, idx.N), Dp_i=rng.uniform(0.01, 0.5, idx.total_sites),
                              E_i=rng.uniform(0.2, 3.0, idx.N), tf_scale=float(rng.uniform(0.1, 4.0))))
        psets[0].update(c_k=np.array([max(0.01, 1.0)] * len(idx.kinases)), A_i=np.ones(idx.N), B_i=np.full(idx.N, 0.2), C_i=np.full(idx.N, 0.5),
                        D_i=np.full(idx.N, 0.05), Dp_i=np.full(idx.total_sites, 0.05), E_i=np.ones(idx.N), tf_scale=0.1)   # runner.py:515-524
        # System keeps references to the arrays it is given (np.ascontiguousarray does not copy) and update() writes in place:
        # hand it copies so that the stored parameter sets stay what was actually simulated
        sysm = net.System(idx, W, tf_mat, kin_in, {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in psets[0].items()}, tf_deg)
        S = idx.state_dim
        t_eval = np.unique(np.concatenate([cfg.TIME_POINTS_PROTEIN, cfg.TIME_POINTS_RNA, cfg.TIME_POINTS_PHOSPHO])).astype(float)
        t_probe = np.array([0.0, 0.3, 0.5, 0.6, 3.0, 16.0, 100.0, 960.0, 2000.0, -1.0])
        y0 = sysm.y0()
        y_rand = rng.uniform(0.0, 2.0, (K, S))
        rhs_y0 = np.empty((K, t_probe.size, S)); rhs_rand = np.empty((K, t_probe.size, S))
        fdjac = np.empty((2, S, S)); Y8 = np.empty((K, t_eval.size, S)); Ytight = np.empty((2, t_eval.size, S))
        Scache = None
        Yrk = np.empty((2, t_eval.size, S)); Yrk_tight = np.empty((1, t_eval.size, S))
        for k, ps in enumerate(psets):
            sysm.update(**ps)
            if MODEL == 2:
                js.build_S_cache_into(sysm.S_cache, sysm.W_indptr, sysm.W_indices, sysm.W_data, sysm.kin_Kmat, sysm.c_k)
                args = sysm.odeint_args(sysm.S_cache)
                if k == 1: Scache = sysm.S_cache.copy()
            else:
                args = sysm.odeint_args()
            for ti, t in enumerate(t_probe):
                rhs_y0[k, ti] = js.rhs_odeint(y0.copy(), float(t), *args)
                rhs_rand[k, ti] = js.rhs_odeint(y_rand[k].copy(), float(t), *args)
            if k < 2:
                fdjac[k] = js.fd_jacobian_odeint(y_rand[k].copy(), 3.0, *args)
            Y8[k] = sim.simulate_odeint(sysm, t_eval, 1e-8, 1e-8, 200000)
            if k < 2:
                Ytight[k] = sim.simulate_odeint(sysm, t_eval, 1e-12, 1e-12, 500000)
            # the reference's opt-in explicit integrator (solvers.py:293-758 through jacspeedup.solve_custom, jacspeedup.py:31-64)
            if k < 2:
                Yrk[k] = js.solve_custom(sysm, y0.copy(), t_eval, 1e-5, 1e-7)          # simulate.py defaults for the custom solver
            if k < 1:
                Yrk_tight[k] = js.solve_custom(sysm, y0.copy(), t_eval, 1e-9, 1e-11)
            print(model_name, tag, "param set", k, "done", flush=True)
        driver_map = np.asarray(sysm.odeint_args(sysm.S_cache)[-3] if MODEL == 2 else sysm.odeint_args()[-1], dtype=np.int32)
        d = dict(model=MODEL, N=idx.N, n_K=len(idx.kinases), total_sites=idx.total_sites, S=S,
                 offset_y=idx.offset_y, offset_s=idx.offset_s, n_sites=idx.n_sites,
                 W_indptr=sysm.W_indptr, W_indices=sysm.W_indices, W_data=sysm.W_data, n_W_rows=sysm.n_W_rows,
                 TF_indptr=sysm.TF_indptr, TF_indices=sysm.TF_indices, TF_data=sysm.TF_data, tf_deg=sysm.tf_deg,
                 driver_map=driver_map, kin_grid=sysm.kin_grid, kin_Kmat=sysm.kin_Kmat,
                 t_eval=t_eval, t_probe=t_probe, y0=y0, y_rand=y_rand, rhs_y0=rhs_y0, rhs_rand=rhs_rand, fd_jac=fdjac, fd_jac_t=3.0,
                 Y_lsoda8=Y8, Y_tight=Ytight, Y_rk45=Yrk, Y_rk45_tight=Yrk_tight,
                 c_k=np.stack([p["c_k"] for p in psets]), A_i=np.stack([p["A_i"] for p in psets]), B_i=np.stack([p["B_i"] for p in psets]),
                 C_i=np.stack([p["C_i"] for p in psets]), D_i=np.stack([p["D_i"] for p in psets]), Dp_i=np.stack([p["Dp_i"] for p in psets]),
                 E_i=np.stack([p["E_i"] for p in psets]), tf_scale=np.array([p["tf_scale"] for p in psets]))
        if MODEL == 2:
            d.update(n_states=idx.n_states, trans_from=sysm.trans_from, trans_to=sysm.trans_to, trans_site=sysm.trans_site,
                     trans_off=sysm.trans_off, trans_n=sysm.trans_n, S_cache_set1=Scache)
        np.savez_compressed(OUT / f"network_m{MODEL}_{tag}.npz", **d)
        print("wrote", model_name, tag, "N", idx.N, "S", S, "n_K", len(idx.kinases), "sites", idx.total_sites, "proxy", idx.proxy_map, flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


def main_large(model_name):
    """One N = 100 network per topology at BASELINE config 4 / 5 size: S ~ 500 states, n_K = 40 kinases (20 of them
    proteins => driven), 250 TF edges, 1-6 sites per protein (combinatorial: 1-3 => 2^ns states).  Slow in pure Python (the finite-
    difference Dfun alone is S + 1 right-hand sides per refresh); run once:  python tools/make_golden_network.py large [model]
    Stored: two parameter sets, rhs probes, the VERBATIM simulate_odeint at the optimiser's tolerance (1e-8 / 1e-8, config.toml:403-404)
    for both and at 1e-12 for the first, the reference's RK45 at its defaults for the first."""
    import pandas as pd, time
    mods, tmp = import_reference(model_name)
    cfg, net, bm, js, sim = mods["config"], mods["network"], mods["buildmat"], mods["jacspeedup"], mods["simulate"]
    MODEL = cfg.MODEL
    rng = np.random.default_rng(20260515 + 3 + 100 * MODEL)
    N, n_ext, n_kp, n_tf = 100, 20, 20, 250
    max_sites = 3 if MODEL == 2 else 6
    prots, kinases, inter, tf_net = synth_frames(rng, N, max_sites, n_ext, n_kp, n_tf)
    idx = net.Index(inter, tf_interactions=tf_net, kin_beta_map={k: float(rng.uniform(0.5, 1.5)) for k in kinases}, tf_beta_map={})
    grid = np.asarray(cfg.TIME_POINTS_PROTEIN, float)
    fc_rows = []
    for k in idx.kinases:
        base = 1.0 + 0.3 * np.sin(rng.uniform(0, 6) + np.arange(grid.size) * rng.uniform(0.2, 0.8))
        for t, v in zip(grid, base):
            fc_rows.append(dict(protein=k, time=float(t), fc=float(max(v, 1e-6))))
    kin_in = net.KinaseInput(idx.kinases, pd.DataFrame(fc_rows))
    W = bm.build_W_parallel(inter, idx, n_cores=1)
    tf_mat = bm.build_tf_matrix(tf_net, idx, tf_beta_map={}, kin_beta_map={})
    tf_deg = np.asarray(np.abs(tf_mat).sum(axis=1)).ravel().astype(np.float64)
    tf_deg[tf_deg < 1e-12] = 1.0
    nK = len(idx.kinases)
    psets = [dict(c_k=np.ones(nK), A_i=np.ones(idx.N), B_i=np.full(idx.N, 0.2), C_i=np.full(idx.N, 0.5), D_i=np.full(idx.N, 0.05),
                  Dp_i=np.full(idx.total_sites, 0.05), E_i=np.ones(idx.N), tf_scale=0.1),                       # runner.py:515-524
             dict(c_k=rng.uniform(0.3, 2.0, nK), A_i=rng.uniform(0.3, 2.0, idx.N), B_i=rng.uniform(0.05, 1.0, idx.N),
                  C_i=rng.uniform(0.1, 2.0, idx.N), D_i=rng.uniform(0.01, 0.5, idx.N), Dp_i=rng.uniform(0.01, 0.5, idx.total_sites),
                  E_i=rng.uniform(0.2, 3.0, idx.N), tf_scale=float(rng.uniform(0.1, 4.0)))]
    K = len(psets)
    sysm = net.System(idx, W, tf_mat, kin_in, {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in psets[0].items()}, tf_deg)
    S = idx.state_dim
    t_eval = np.unique(np.concatenate([cfg.TIME_POINTS_PROTEIN, cfg.TIME_POINTS_RNA, cfg.TIME_POINTS_PHOSPHO])).astype(float)
    t_probe = np.array([0.0, 0.5, 3.0, 960.0])
    y0 = sysm.y0()
    y_rand = rng.uniform(0.0, 2.0, (K, S))
    rhs_y0 = np.empty((K, t_probe.size, S)); rhs_rand = np.empty((K, t_probe.size, S))
    Y8 = np.empty((K, t_eval.size, S)); Ytight = np.empty((1, t_eval.size, S)); Yrk = np.empty((1, t_eval.size, S))
    print(model_name, "large: N", idx.N, "S", S, "n_K", nK, "sites", idx.total_sites, flush=True)
    for k, ps in enumerate(psets):
        sysm.update(**ps)
        if MODEL == 2:
            js.build_S_cache_into(sysm.S_cache, sysm.W_indptr, sysm.W_indices, sysm.W_data, sysm.kin_Kmat, sysm.c_k)
            args = sysm.odeint_args(sysm.S_cache)
        else:
            args = sysm.odeint_args()
        for ti, t in enumerate(t_probe):
            rhs_y0[k, ti] = js.rhs_odeint(y0.copy(), float(t), *args)
            rhs_rand[k, ti] = js.rhs_odeint(y_rand[k].copy(), float(t), *args)
        t0 = time.time()
        Y8[k] = sim.simulate_odeint(sysm, t_eval, 1e-8, 1e-8, 200000)
        print(model_name, "large set", k, "LSODA 1e-8 done in", round(time.time() - t0), "s", flush=True)
        if k == 0:
            t0 = time.time()
            Yrk[0] = js.solve_custom(sysm, y0.copy(), t_eval, 1e-5, 1e-7)
            print(model_name, "large RK45 done in", round(time.time() - t0), "s", flush=True)
            t0 = time.time()
            Ytight[0] = sim.simulate_odeint(sysm, t_eval, 1e-12, 1e-12, 500000)
            print(model_name, "large LSODA 1e-12 done in", round(time.time() - t0), "s", flush=True)
    driver_map = np.asarray(sysm.odeint_args(sysm.S_cache)[-3] if MODEL == 2 else sysm.odeint_args()[-1], dtype=np.int32)
    d = dict(model=MODEL, N=idx.N, n_K=nK, total_sites=idx.total_sites, S=S, offset_y=idx.offset_y, offset_s=idx.offset_s, n_sites=idx.n_sites,
             W_indptr=sysm.W_indptr, W_indices=sysm.W_indices, W_data=sysm.W_data, n_W_rows=sysm.n_W_rows,
             TF_indptr=sysm.TF_indptr, TF_indices=sysm.TF_indices, TF_data=sysm.TF_data, tf_deg=sysm.tf_deg,
             driver_map=driver_map, kin_grid=sysm.kin_grid, kin_Kmat=sysm.kin_Kmat,
             t_eval=t_eval, t_probe=t_probe, y0=y0, y_rand=y_rand, rhs_y0=rhs_y0, rhs_rand=rhs_rand,
             Y_lsoda8=Y8, Y_tight=Ytight, Y_rk45=Yrk,
             c_k=np.stack([p["c_k"] for p in psets]), A_i=np.stack([p["A_i"] for p in psets]), B_i=np.stack([p["B_i"] for p in psets]),
             C_i=np.stack([p["C_i"] for p in psets]), D_i=np.stack([p["D_i"] for p in psets]), Dp_i=np.stack([p["Dp_i"] for p in psets]),
             E_i=np.stack([p["E_i"] for p in psets]), tf_scale=np.array([p["tf_scale"] for p in psets]))
    if MODEL == 2:
        d.update(n_states=idx.n_states, trans_from=sysm.trans_from, trans_to=sysm.trans_to, trans_site=sysm.trans_site,
                 trans_off=sysm.trans_off, trans_n=sysm.trans_n)
    np.savez_compressed(OUT / f"netlarge_m{MODEL}.npz", **d)
    print("wrote netlarge", model_name, flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


def main_large_more(model_name, n_more=6):
    """More reference-run truth at config 4 / 5 size (population parity must not be self-referential): the SAME N = 100
    network as ``main_large`` (same generator stream, checked against the committed fixture), the second parameter set of that fixture
    (which had no 1e-12 run) plus ``n_more`` new candidates -- half log-normal(0, 0.5) around the optimiser's defaults (the shape of
    bench.py's config-5 population), half uniform over wide ranges -- each integrated by the reference's ``simulate_odeint`` VERBATIM at
    rtol = atol = 1e-12.  Written after every candidate (tests/golden/netlarge_more_m<M>.npz), so an interrupted run keeps what it has.
        python tools/make_golden_network.py large_more [model]"""
    import pandas as pd, time
    mods, tmp = import_reference(model_name)
    cfg, net, bm, js, sim = mods["config"], mods["network"], mods["buildmat"], mods["jacspeedup"], mods["simulate"]
    MODEL = cfg.MODEL
    rng = np.random.default_rng(20260515 + 3 + 100 * MODEL)
    N, n_ext, n_kp, n_tf = 100, 20, 20, 250
    max_sites = 3 if MODEL == 2 else 6
    prots, kinases, inter, tf_net = synth_frames(rng, N, max_sites, n_ext, n_kp, n_tf)
    idx = net.Index(inter, tf_interactions=tf_net, kin_beta_map={k: float(rng.uniform(0.5, 1.5)) for k in kinases}, tf_beta_map={})
    grid = np.asarray(cfg.TIME_POINTS_PROTEIN, float)
    fc_rows = []
    for k in idx.kinases:
        base = 1.0 + 0.3 * np.sin(rng.uniform(0, 6) + np.arange(grid.size) * rng.uniform(0.2, 0.8))
        for t, v in zip(grid, base):
            fc_rows.append(dict(protein=k, time=float(t), fc=float(max(v, 1e-6))))
    kin_in = net.KinaseInput(idx.kinases, pd.DataFrame(fc_rows))
    W = bm.build_W_parallel(inter, idx, n_cores=1)
    tf_mat = bm.build_tf_matrix(tf_net, idx, tf_beta_map={}, kin_beta_map={})
    tf_deg = np.asarray(np.abs(tf_mat).sum(axis=1)).ravel().astype(np.float64)
    tf_deg[tf_deg < 1e-12] = 1.0
    nK = len(idx.kinases)
    old = np.load(OUT / f"netlarge_m{MODEL}.npz")
    defaults = dict(c_k=np.ones(nK), A_i=np.ones(idx.N), B_i=np.full(idx.N, 0.2), C_i=np.full(idx.N, 0.5), D_i=np.full(idx.N, 0.05),
                    Dp_i=np.full(idx.total_sites, 0.05), E_i=np.ones(idx.N), tf_scale=0.1)
    sysm = net.System(idx, W, tf_mat, kin_in, {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in defaults.items()}, tf_deg)
    # the same network as the committed fixture, array for array
    for key, val in (("W_indptr", sysm.W_indptr), ("W_indices", sysm.W_indices), ("W_data", sysm.W_data), ("TF_data", sysm.TF_data), ("kin_Kmat", sysm.kin_Kmat),
                     ("offset_y", idx.offset_y), ("n_sites", idx.n_sites)):
        np.testing.assert_array_equal(np.asarray(val), old[key], err_msg=key)
    keys = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")
    psets = [{**{k: old[k][1].copy() for k in keys}, "tf_scale": float(old["tf_scale"][1])}]
    r2 = np.random.default_rng(20260515 + 977 + 100 * MODEL)
    for j in range(n_more):
        if j % 2 == 0:
            ps = {k: defaults[k] * np.exp(0.5 * r2.standard_normal(defaults[k].shape)) for k in keys}
            ps["tf_scale"] = float(0.1 * np.exp(0.5 * r2.standard_normal()))
        else:
            ps = dict(c_k=r2.uniform(0.3, 2.0, nK), A_i=r2.uniform(0.3, 2.0, idx.N), B_i=r2.uniform(0.05, 1.0, idx.N), C_i=r2.uniform(0.1, 2.0, idx.N),
                      D_i=r2.uniform(0.01, 0.5, idx.N), Dp_i=r2.uniform(0.01, 0.5, idx.total_sites), E_i=r2.uniform(0.2, 3.0, idx.N),
                      tf_scale=float(r2.uniform(0.1, 4.0)))
        psets.append(ps)
    t_eval = old["t_eval"]
    S = idx.state_dim
    Yt = np.full((len(psets), t_eval.size, S), np.nan)
    secs = np.zeros(len(psets))
    for k, ps in enumerate(psets):
        sysm.update(**ps)
        if MODEL == 2:
            js.build_S_cache_into(sysm.S_cache, sysm.W_indptr, sysm.W_indices, sysm.W_data, sysm.kin_Kmat, sysm.c_k)
        t0 = time.time()
        Yt[k] = sim.simulate_odeint(sysm, t_eval, 1e-12, 1e-12, 500000)
        secs[k] = time.time() - t0
        print(model_name, "large_more candidate", k, "LSODA 1e-12 done in", round(secs[k]), "s", flush=True)
        d = dict(model=MODEL, done=k + 1, t_eval=t_eval, y0=sysm.y0(), Y_tight=Yt[:k + 1], seconds=secs[:k + 1], from_netlarge_index=np.array([1] + [-1] * k)[:k + 1],
                 tf_scale=np.array([p["tf_scale"] for p in psets[:k + 1]]), **{kk: np.stack([p[kk] for p in psets[:k + 1]]) for kk in keys})
        np.savez_compressed(OUT / f"netlarge_more_m{MODEL}.npz", **d)
    print("wrote netlarge_more", model_name, flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "large_more":
        if len(sys.argv) > 2:
            main_large_more(sys.argv[2]); sys.exit(0)
        procs = [subprocess.Popen([sys.executable, __file__, "large_more", m]) for m in MODEL_ID]
        sys.exit(max(p.wait() for p in procs))
    if which == "large":
        if len(sys.argv) > 2:
            main_large(sys.argv[2]); sys.exit(0)
        procs = [subprocess.Popen([sys.executable, __file__, "large", m]) for m in MODEL_ID]
        sys.exit(max(p.wait() for p in procs))
    if which == "all":
        procs = [subprocess.Popen([sys.executable, __file__, m]) for m in MODEL_ID]
        sys.exit(max(p.wait() for p in procs))
    main(which)
