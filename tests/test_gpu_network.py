"""GPU parity of the network path against the golden vectors made by running the reference's global_model classes:
right-hand side (all four kinetic topologies, inside buckets and exactly on bucket edges), analytic Jacobian against the
reference's finite-difference Jacobian, softplus unpack."""
from pathlib import Path

import numpy as np
import pytest

from oracle import network_models as nm

pytestmark = pytest.mark.gpu
GOLD = sorted((Path(__file__).resolve().parent / "golden").glob("network_m*.npz"))


def _x(eng, g, k):
    return eng.pack_params(g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], g["tf_scale"][k])


@pytest.mark.parametrize("f", GOLD, ids=lambda f: f.stem)
def test_network_rhs_matches_reference(f):
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    assert eng.S == int(g["S"]) and eng.n_var == int(g["n_K"]) + 5 * int(g["N"]) + int(g["total_sites"]) + 1
    np.testing.assert_array_equal(eng.default_y0(), g["y0"])
    X = np.stack([_x(eng, g, k) for k in range(4)])
    for ti, t in enumerate(g["t_probe"]):
        d0 = eng.rhs_batch(X, g["y0"], float(t)).cpu().numpy()
        dr = eng.rhs_batch(X, g["y_rand"], float(t)).cpu().numpy()
        scale = 1.0 + np.abs(g["rhs_rand"][:, ti]).max()
        np.testing.assert_allclose(d0, g["rhs_y0"][:, ti], rtol=1e-12, atol=1e-13 * scale)
        np.testing.assert_allclose(dr, g["rhs_rand"][:, ti], rtol=1e-12, atol=1e-13 * scale)
    # per-candidate times in one launch
    tt = np.array([0.3, 0.5, 16.0, 2000.0])
    d = eng.rhs_batch(X, g["y_rand"], tt).cpu().numpy()
    for k in range(4):
        ti = int(np.where(g["t_probe"] == tt[k])[0][0])
        np.testing.assert_allclose(d[k], g["rhs_rand"][k, ti], rtol=1e-12, atol=1e-12)
    eng.close()


@pytest.mark.parametrize("f", GOLD, ids=lambda f: f.stem)
def test_network_analytic_jacobian_vs_reference_fd(f):
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    net = nm.Network.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(2)])
    J = eng.jacobian_batch(X, g["y_rand"][:2], float(g["fd_jac_t"])).cpu().numpy()
    for k in range(2):
        fd = g["fd_jac"][k]                                   # forward differences, h = 1e-8 max(1, |y_j|): ~1e-7 absolute noise
        assert np.abs(J[k] - fd).max() <= 2e-6 * (1.0 + np.abs(fd).max())
        # sparsity: where the forward difference is EXACTLY zero the true derivative is below ulp(f) / h ~ 2e-8 |f|, so the analytic
        # entry must be (numerically) zero there -- and the bulk of those entries are structural zeros that must be exactly 0.0
        zero = fd == 0.0
        assert (np.abs(J[k][zero]) <= 1e-7 * (1.0 + np.abs(fd).max())).all()
        assert (J[k][zero] == 0.0).mean() > 0.9
        # tighter, independent check: central differences of the ORACLE rhs at a step that balances truncation / rounding
        p = nm.Params.from_npz(g, k)
        y = g["y_rand"][k]
        cols = np.linspace(0, net.S - 1, 7).astype(int)
        for c in cols:
            h = 1e-5
            yp = y.copy(); ym = y.copy(); yp[c] += h; ym[c] -= h
            cd = (nm.rhs(net, p, yp, 3.0) - nm.rhs(net, p, ym, 3.0)) / (2 * h)
            np.testing.assert_allclose(J[k][:, c], cd, rtol=2e-7, atol=2e-8)
    eng.close()


def test_network_softplus_unpack_and_raw_path():
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(GOLD[0])
    eng = NetworkEngine.from_npz(g)
    rng = np.random.default_rng(0)
    raw = rng.uniform(-30, 30, (5, eng.n_var)); raw[0, :3] = [25.0, 20.0, 20.000001]
    phys = eng.unpack_batch(raw).cpu().numpy()
    want = np.where(raw > 20.0, raw, np.log1p(np.exp(raw)))         # utils.py:229-241
    np.testing.assert_allclose(phys, want, rtol=1e-15, atol=0)
    a = eng.rhs_batch(raw, g["y0"], 3.0, raw=True).cpu().numpy()
    b = eng.rhs_batch(phys, g["y0"], 3.0, raw=False).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    eng.close()


def test_network_create_rejects_bad_topology():
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd._capi import PhoskinError
    g = dict(np.load(GOLD[0]))
    bad = dict(g); bad["W_indices"] = g["W_indices"].copy(); bad["W_indices"][0] = 999
    with pytest.raises(PhoskinError):
        NetworkEngine.from_npz(bad)
    bad = dict(g); bad["offset_y"] = g["offset_y"].copy(); bad["offset_y"][1] += 1
    with pytest.raises(PhoskinError):
        NetworkEngine.from_npz(bad)


@pytest.mark.parametrize("f", GOLD, ids=lambda f: f.stem)
def test_network_simulate_within_band_of_reference_lsoda(f):
    """simulate_odeint batched (ROS34PW2, block-diagonal W) against the reference's LSODA run at 1e-12 (`Y_tight`); the reference's own
    production tolerance (1e-8 / 1e-8, `Y_lsoda8`) is checked to be no closer to the truth than we are required to be."""
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(4)])
    Y, st, ns = eng.simulate_batch(X, g["t_eval"])
    Y = Y.cpu().numpy()
    assert not st.cpu().numpy().any()
    np.testing.assert_array_equal(Y[:, 0, :], np.broadcast_to(g["y0"], (4, eng.S)))
    band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
    for k in range(2):
        assert band(Y[k], g["Y_tight"][k]) <= 0.5, (f.name, k, band(Y[k], g["Y_tight"][k]))
    for k in range(4):      # against the reference's own 1e-8 run: within its error + ours
        assert band(Y[k], g["Y_lsoda8"][k]) <= 1.5
    # determinism and independence of the batch composition
    Y2, _, _ = eng.simulate_batch(X[[2, 0]], g["t_eval"])
    np.testing.assert_array_equal(Y2.cpu().numpy(), Y[[2, 0]])
    eng.close()


def test_network_simulate_edge_cases():
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd._capi import PhoskinError, ST_MAXSTEPS
    g = np.load([x for x in GOLD if x.name == "network_m0_small.npz"][0])
    eng = NetworkEngine.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(3)])
    # output grid that does not contain the bucket edges: the integrator must still stop at every edge
    t = np.array([0.0, 0.6, 3.0, 50.0, 960.0])
    Y, st, ns = eng.simulate_batch(X, t)
    Yfull, _, _ = eng.simulate_batch(X, g["t_eval"])
    assert not st.cpu().numpy().any()
    net = nm.Network.from_npz(g)
    ref = nm.simulate_odeint(net, nm.Params.from_npz(g, 1), t, 1e-12, 1e-12, 500000)
    assert np.max(np.abs(Y.cpu().numpy()[1] - ref) / (1e-8 + 1e-6 * np.abs(ref))) <= 0.5
    np.testing.assert_allclose(Y.cpu().numpy()[:, -1], Yfull.cpu().numpy()[:, -1], rtol=1e-6, atol=1e-8)
    # T = 1, step budget exhaustion, NaN candidate
    Y1, st1, _ = eng.simulate_batch(X, [0.0])
    assert Y1.shape == (3, 1, eng.S)
    Ym, stm, _ = eng.simulate_batch(X, g["t_eval"], max_steps=10)
    assert (stm.cpu().numpy() & ST_MAXSTEPS).all() and np.isnan(Ym.cpu().numpy()[:, -1]).all() and np.isfinite(Ym.cpu().numpy()[:, 0]).all()
    Xb = X.copy(); Xb[1, 3] = np.nan
    Yb, stb, _ = eng.simulate_batch(Xb, g["t_eval"])
    stb = stb.cpu().numpy()
    assert stb[1] != 0 and stb[0] == 0 and stb[2] == 0
    np.testing.assert_array_equal(Yb.cpu().numpy()[[0, 2]], Yfull.cpu().numpy()[[0, 2]])
    with pytest.raises(PhoskinError):
        eng.simulate_batch(X, [0.0, 1.0, 1.0])
    eng.close()
    # combinatorial topology: beyond 16 sites per protein (65 536-state blocks) the request is refused, not mis-integrated
    from phoskintime_amd.global_model import synthetic
    net4 = synthetic.make_network(N=8, total_sites=12, n_K=4, n_tf_edges=10, model=0, seed=5)
    net4["model"] = 2
    ns = net4["n_sites"].copy(); ns[:] = 1; ns[0] = 17
    net4["n_sites"] = ns
    net4["offset_s"] = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(np.int32)
    net4["offset_y"] = np.concatenate([[0], np.cumsum(1 + (1 << ns.astype(np.int64)))[:-1]]).astype(np.int32)
    tot = int(ns.sum())
    net4["W_indptr"] = np.arange(tot + 1, dtype=np.int32); net4["W_indices"] = np.zeros(tot, np.int32); net4["W_data"] = np.ones(tot)
    with pytest.raises(PhoskinError):
        e2 = NetworkEngine(**net4)
        e2.simulate_batch(np.ones((1, e2.n_var)), [0.0, 1.0])


@pytest.mark.parametrize("m", [0, 2])
def test_network_loss_all_modes_match_reference(m):
    """LOSS_FN on the GPU against the sums the reference's lossfn produced (all eight LOSS_MODEs; mode 2 is NaN in the reference, too)."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine
    gl = np.load(Path(__file__).resolve().parent / "golden" / f"network_loss_m{m}.npz")
    g = np.load(Path(__file__).resolve().parent / "golden" / f"network_m{m}_small.npz")
    eng = NetworkEngine.from_npz(g)
    ld = {k: gl[k] for k in gl.files}
    T = gl["Y"].shape[1]
    loss = eng.make_loss(ld, T)
    Y = torch.as_tensor(gl["Y"], device="cuda")
    for mode in range(8):
        sums, F = eng.objective_batch(loss, Y, loss_mode=mode, lambdas=(2.0, 3.0, 5.0, 0.0))
        np.testing.assert_allclose(sums.cpu().numpy(), gl["loss_sums"][mode], rtol=1e-12, atol=0, equal_nan=True)
        norm = [1.0 / max(1e-6, gl[w].sum()) for w in ("w_prot", "w_rna", "w_pho")]
        want = gl["loss_sums"][mode] * np.array(norm) * np.array([2.0, 3.0, 5.0])
        np.testing.assert_allclose(F.cpu().numpy(), want, rtol=1e-12, equal_nan=True)
    # prior penalty + failure handling against the oracle's restatement of _evaluate
    net = nm.Network.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(4)])
    defaults = X[0] * 1.3
    lam = dict(protein=1.0, rna=0.5, phospho=2.0, prior=0.7)
    Yb = gl["Y"].copy(); Yb[2, 3, 1] = np.inf
    st = torch.zeros(4, dtype=torch.int32, device="cuda"); st[3] = 2
    sums, F = eng.objective_batch(loss, torch.as_tensor(Yb, device="cuda"), loss_mode=0, x=X, defaults=defaults,
                                  lambdas=(lam["protein"], lam["rna"], lam["phospho"], lam["prior"]), status=st)
    F = F.cpu().numpy()
    for k in range(4):
        want = nm.objectives(net, X[k], defaults, None if k == 3 else Yb[k], ld, 0, lam)
        np.testing.assert_allclose(F[k], want, rtol=1e-12)
    assert (F[2] == 1e12).all() and (F[3] == 1e12).all()
    ld_bad = dict(ld); ld_bad["t_prot"] = ld["t_prot"].copy(); ld_bad["t_prot"][0] = T
    from phoskintime_amd._capi import PhoskinError
    with pytest.raises(PhoskinError):
        eng.make_loss(ld_bad, T)
    eng.free_loss(loss)
    eng.close()


def test_simulate_and_measure_and_network_morris():
    """Array form of simulate_and_measure against a numpy restatement on the oracle trajectory; Morris driver end to end (small)."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd.global_model.simulate import measure_batch
    from phoskintime_amd.global_model.sensitivity import run_sensitivity_batch, compute_bounds, _reconstruct_params
    g = np.load([x for x in GOLD if x.name == "network_m0_small.npz"][0])
    eng = NetworkEngine.from_npz(g)
    times = g["t_eval"]
    tp = np.array([0.0, 0.5, 1.0, 4.0, 960.0]); tr = np.array([4.0, 15.0, 60.0]); tph = np.array([0.0, 2.0, 30.0])
    Y = torch.as_tensor(g["Y_lsoda8"], device="cuda")
    pred, ld = measure_batch(eng, Y, times, tp, tr, tph)
    pred = pred.cpu().numpy()
    # numpy restatement of simulate.py:119-202 for non-combinatorial layouts
    for b in range(2):
        Yb = g["Y_lsoda8"][b]; col = 0
        want = []
        for k in range(ld["p_prot"].size):
            i, t = ld["p_prot"][k], ld["t_prot"][k]; st = g["offset_y"][i]; ns = g["n_sites"][i]
            tot = Yb[:, st + 1] + Yb[:, st + 2:st + 2 + ns].sum(axis=1)
            want.append(max(tot[t], 1e-12) / max(tot[0], 1e-12))
        rb = int(np.argmin(np.abs(times - 4.0)))
        for k in range(ld["p_rna"].size):
            st = g["offset_y"][ld["p_rna"][k]]
            want.append(max(Yb[ld["t_rna"][k], st], 1e-12) / max(Yb[rb, st], 1e-12))
        for k in range(ld["p_pho"].size):
            st = g["offset_y"][ld["p_pho"][k]] + 2 + ld["s_pho"][k]
            want.append(max(Yb[ld["t_pho"][k], st], 1e-12) / max(Yb[0, st], 1e-12))
        np.testing.assert_allclose(pred[b], np.array(want), rtol=1e-13)
    assert ld["p_prot"].size == int(g["N"]) * tp.size and ld["p_pho"].size == int(g["total_sites"]) * tph.size
    # Morris on the fitted point = parameter set 1
    fitted = dict(c_k=g["c_k"][1], A_i=g["A_i"][1], B_i=g["B_i"][1], C_i=g["C_i"][1], D_i=g["D_i"][1], Dp_i=g["Dp_i"][1], E_i=g["E_i"][1],
                  tf_scale=float(g["tf_scale"][1]))
    out = run_sensitivity_batch(eng, fitted, tp, tr, tph, perturbation=0.05, trajectories=4, num_levels=8, seed=3)
    D = eng.n_var
    assert out["param_values"].shape == (4 * (D + 1), D) and out["Y"].shape == (4 * (D + 1),) and not out["status"].any()
    assert len(out["Si"]["mu_star"]) == D and np.isfinite(out["Si"]["mu_star"]).all()
    # one row re-done by hand: candidate -> simulate -> observables -> sum
    row = out["param_values"][7]
    from phoskintime_amd.global_model.simulate import measure_tolerances
    Yr, _, _ = eng.simulate_batch(row[None], np.unique(np.concatenate([tp, tr, tph])), **measure_tolerances(eng))
    pr, _ = measure_batch(eng, Yr, np.unique(np.concatenate([tp, tr, tph])), tp, tr, tph)
    assert out["Y"][7] == pytest.approx(float(pr.sum()), rel=1e-12)
    prob = compute_bounds(fitted, 0.05)
    assert prob["num_vars"] == D and prob["names"][0] == "c_k_0" and prob["names"][-1] == "tf_scale"
    shapes = {k: (np.asarray(v).shape if k != "tf_scale" else ()) for k, v in fitted.items()}
    rec = _reconstruct_params(row, None, shapes)
    np.testing.assert_array_equal(eng.pack_params(**rec), row)
    eng.close()


@pytest.mark.parametrize("model", [0, 1, 4])
def test_register_and_lds_network_kernels_agree(model):
    """The one-thread-per-protein register kernel and the general LDS kernel implement the same method: on a config-5-shaped
    synthetic network (S = 500, up to 6 sites per protein => MAXS = 8 path) they must agree far inside the parity band."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    net = synthetic.make_network(model=model)
    eng = NetworkEngine(**net)
    X = synthetic.random_candidates(net, 24, seed=3)
    t = np.unique(np.concatenate([net["kin_grid"], [15.0]]))
    Ya, sa, na = eng.simulate_batch(X, t, kernel="auto", method="rosw")
    Yb, sb, nb = eng.simulate_batch(X, t, kernel="lds", method="rosw")
    assert not sa.cpu().numpy().any() and not sb.cpu().numpy().any()
    Ya, Yb = Ya.cpu().numpy(), Yb.cpu().numpy()
    assert np.max(np.abs(Ya - Yb) / (1e-8 + 1e-6 * np.abs(Yb))) <= 0.2
    assert abs(int(na[:, 0].sum()) - int(nb[:, 0].sum())) <= 0.02 * int(nb[:, 0].sum())
    # the order-4 additive method (the default on this layout) against the order-3 W-method run much tighter: a different integrator,
    # the same trajectories inside the parity band, and at least 3x fewer steps at the optimiser's tolerance
    Yk, sk, nk = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8, method="ark")
    Yr, sr, nr = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8, method="rosw")
    Yt, _, _ = eng.simulate_batch(X, t, rtol=1e-10, atol=1e-11, method="rosw")
    assert not sk.cpu().numpy().any()
    assert np.max(np.abs(Yk.cpu().numpy() - Yt.cpu().numpy()) / (1e-8 + 1e-6 * np.abs(Yt.cpu().numpy()))) <= 0.5
    assert int(nk[:, 0].sum()) * 3 <= int(nr[:, 0].sum())
    eng.close()


def _fake_system(g, k):
    """A stand-in with the attribute surface of the reference's System / Index (network.py:28-526) built from a golden file."""
    from types import SimpleNamespace
    N = int(g["N"]); nK = int(g["n_K"])
    prots = [f"P{i:03d}" for i in range(N)]
    kin = [f"K{j:02d}" for j in range(nK)]
    # driver_map -> names: protein i driven by kinase d is "the kinase itself"; keep it simple: map through p2i / k2i
    p2i = {p: i for i, p in enumerate(prots)}
    k2i = {}
    kinases = []
    drv = g["driver_map"]
    used = {}
    for i in range(N):
        if drv[i] >= 0 and int(drv[i]) not in used:
            used[int(drv[i])] = prots[i]
    for j in range(nK):
        name = used.get(j, kin[j])
        kinases.append(name); k2i[name] = j
    proxy = {}
    for i in range(N):        # a second protein driven by an already-used kinase is an orphan proxied to it
        if drv[i] >= 0 and used[int(drv[i])] != prots[i]:
            proxy[prots[i]] = used[int(drv[i])]
    sites = [[f"S{10 * (j + 1)}" for j in range(int(ns))] for ns in g["n_sites"]]
    idx = SimpleNamespace(N=N, proteins=prots, kinases=kinases, p2i=p2i, k2i=k2i, proxy_map=proxy, sites=sites,
                          offset_y=g["offset_y"], offset_s=g["offset_s"], n_sites=g["n_sites"])
    sysm = SimpleNamespace(idx=idx, W_indptr=g["W_indptr"], W_indices=g["W_indices"], W_data=g["W_data"], TF_indptr=g["TF_indptr"],
                           TF_indices=g["TF_indices"], TF_data=g["TF_data"], tf_deg=g["tf_deg"], kin_grid=g["kin_grid"], kin_Kmat=g["kin_Kmat"],
                           c_k=g["c_k"][k].copy(), A_i=g["A_i"][k].copy(), B_i=g["B_i"][k].copy(), C_i=g["C_i"][k].copy(), D_i=g["D_i"][k].copy(),
                           Dp_i=g["Dp_i"][k].copy(), E_i=g["E_i"][k].copy(), tf_scale=float(g["tf_scale"][k]), y0=lambda: g["y0"].copy())

    def update(**kw):                                # the contract of System.update (network.py:293-302): arrays in place, tf_scale a float
        for name, val in kw.items():
            if name == "tf_scale":
                sysm.tf_scale = float(val)
            else:
                getattr(sysm, name)[:] = val
    sysm.update = update
    return sysm, idx


def test_dropin_simulate_odeint_and_measure_with_system_object():
    """global_model.simulate.simulate_odeint / simulate_and_measure take the reference's System / Index objects and see update()s."""
    from phoskintime_amd.global_model import simulate as gsim
    from phoskintime_amd.global_model import config as gcfg
    g = np.load([x for x in GOLD if x.name == "network_m0_small.npz"][0])
    gcfg.MODEL = 0
    sysm, idx = _fake_system(g, 1)
    eng = gsim.engine_for(sysm)
    np.testing.assert_array_equal(eng._keep[10], g["driver_map"])              # driver_map rebuilt exactly as network.py:454-469 does
    Y = gsim.simulate_odeint(sysm, g["t_eval"], 1e-8, 1e-8, 200000)
    assert Y.shape == g["Y_lsoda8"][1].shape and Y.flags["C_CONTIGUOUS"]
    assert np.max(np.abs(Y - g["Y_tight"][1]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][1]))) <= 0.5
    # mutate parameters like System.update does: the next call must see them
    sysm.A_i[:] = g["A_i"][0]; sysm.B_i[:] = g["B_i"][0]; sysm.C_i[:] = g["C_i"][0]; sysm.D_i[:] = g["D_i"][0]; sysm.Dp_i[:] = g["Dp_i"][0]
    sysm.E_i[:] = g["E_i"][0]; sysm.c_k[:] = g["c_k"][0]; sysm.tf_scale = float(g["tf_scale"][0])
    Y0 = gsim.simulate_odeint(sysm, g["t_eval"], 1e-8, 1e-8, 200000)
    assert np.max(np.abs(Y0 - g["Y_tight"][0]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][0]))) <= 0.5
    tp = np.array([0.0, 1.0, 960.0]); tr = np.array([4.0, 60.0]); tph = np.array([0.0, 30.0])
    dfp, dfr, dfph = gsim.simulate_and_measure(sysm, idx, tp, tr, tph)
    assert list(dfp.columns) == ["protein", "time", "pred_fc"] and list(dfph.columns) == ["protein", "psite", "time", "pred_fc"]
    assert len(dfp) == idx.N * 3 and len(dfr) == idx.N * 2 and len(dfph) == int(g["total_sites"]) * 2
    assert set(dfp["time"]) == set(tp) and set(dfr["time"]) == set(tr)
    assert np.allclose(dfp[dfp["time"] == 0.0]["pred_fc"], 1.0) and np.allclose(dfr[dfr["time"] == 4.0]["pred_fc"], 1.0)
    from phoskintime_amd.global_model.sensitivity import _compute_scalar_metric
    assert _compute_scalar_metric(dfp, dfr, dfph, "total_signal") == pytest.approx(dfp.pred_fc.sum() + dfr.pred_fc.sum() + dfph.pred_fc.sum())


def test_network_long_output_grid_uses_the_buffered_stop_list():
    """More than 64 landing points (e.g. the reference's 1000-point steady-state grid, global_model/analysis.py:29-56) go through the
    per-network device buffer instead of the by-value kernel argument: same trajectory at the shared time points."""
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load([x for x in GOLD if x.name == "network_m1_small.npz"][0])
    eng = NetworkEngine.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(2)])
    dense_t = np.unique(np.concatenate([np.logspace(-3, np.log10(960.0), 150), g["t_eval"]]))
    dense_t = np.concatenate([[0.0], dense_t[dense_t > 0]])
    Yd, sd, _ = eng.simulate_batch(X, dense_t)
    Ys, ss, _ = eng.simulate_batch(X, g["t_eval"])
    assert not sd.cpu().numpy().any() and dense_t.size > 64
    idx = [int(np.where(dense_t == t)[0][0]) for t in g["t_eval"]]
    a, b = Yd.cpu().numpy()[:, idx, :], Ys.cpu().numpy()
    assert np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))) <= 0.3
    assert np.isfinite(Yd.cpu().numpy()).all()
    eng.close()


@pytest.mark.parametrize("m", [0, 1, 2, 4])
def test_dropin_jacspeedup_odeint_convention(m):
    """rhs_odeint / fd_jacobian_odeint with the reference's odeint_args tuples (23 entries; 27 with S_cache for the combinatorial
    topology), build_S_cache_into, solve_custom."""
    from phoskintime_amd.global_model import jacspeedup as js, config as gcfg
    g = np.load([x for x in GOLD if x.name == f"network_m{m}_small.npz"][0])
    gcfg.MODEL = m
    k = 1
    N = int(g["N"])
    if m == 2:
        S_cache = np.zeros((int(g["total_sites"]), g["kin_grid"].size))
        js.build_S_cache_into(S_cache, g["W_indptr"], g["W_indices"], g["W_data"], g["kin_Kmat"], g["c_k"][k])
        np.testing.assert_allclose(S_cache, g["S_cache_set1"], rtol=1e-14)
        args = (g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], float(g["tf_scale"][k]), g["kin_grid"],
                S_cache, g["TF_indptr"], g["TF_indices"], g["TF_data"], N, g["offset_y"], g["offset_s"], g["n_sites"], g["n_states"],
                g["trans_from"], g["trans_to"], g["trans_site"], g["trans_off"], g["trans_n"], g["tf_deg"], g["driver_map"], np.zeros(N), np.zeros(N))
    else:
        args = (g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], float(g["tf_scale"][k]), g["kin_grid"],
                g["kin_Kmat"], g["W_indptr"], g["W_indices"], g["W_data"], int(g["total_sites"]), g["TF_indptr"], g["TF_indices"], g["TF_data"], N,
                g["offset_y"], g["offset_s"], g["n_sites"], g["tf_deg"], g["driver_map"])
    for ti, t in enumerate(g["t_probe"][:6]):
        dy = js.rhs_odeint(g["y_rand"][k], float(t), *args)
        np.testing.assert_allclose(dy, g["rhs_rand"][k, ti], rtol=1e-12, atol=1e-12)
    J = js.fd_jacobian_odeint(g["y_rand"][k], float(g["fd_jac_t"]), *args)
    assert J.shape == g["fd_jac"][k].shape and np.abs(J - g["fd_jac"][k]).max() <= 2e-6 * (1.0 + np.abs(g["fd_jac"][k]).max())
    if m != 2:
        sysm, idx = _fake_system(g, k)
        Y = js.solve_custom(sysm, g["y0"], g["t_eval"], 1e-5, 1e-7)          # the reference's RK45, same steps: agreement to round-off
        np.testing.assert_allclose(Y, g["Y_rk45"][k], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("f", GOLD, ids=lambda f: f.stem)
def test_network_explicit_rk45_reproduces_the_reference_integrator(f):
    """PK_METHOD_DP5 against outputs of the reference's own adaptive RK45 (solvers.py:293-758 via jacspeedup.solve_custom) at its
    default tolerances (1e-5 / 1e-7) and at 1e-9 / 1e-11: same accepted / rejected step counts as the oracle's restatement of that
    loop (which reproduces the reference output bit for bit, tests/test_oracle_network_golden.py) and trajectories equal to round-off."""
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    net = nm.Network.from_npz(g)
    X = np.stack([_x(eng, g, k) for k in range(2)])
    Y, st, ns = eng.simulate_batch(X, g["t_eval"], rtol=1e-5, atol=1e-7, max_steps=2_000_000, method="dp5")
    assert not st.cpu().numpy().any()
    np.testing.assert_allclose(Y.cpu().numpy(), g["Y_rk45"], rtol=1e-9, atol=1e-11)
    if "small" in f.name:
        for k in range(2):
            _, acc, rej = nm.simulate_rk45(net, nm.Params.from_npz(g, k), g["t_eval"], 1e-5, 1e-7, y0=g["y0"], return_steps=True)
            assert tuple(ns[k].cpu().numpy()) == (acc, rej)
    Yt, st, _ = eng.simulate_batch(X[:1], g["t_eval"], rtol=1e-9, atol=1e-11, max_steps=2_000_000, method="dp5")
    assert not st.cpu().numpy().any()
    np.testing.assert_allclose(Yt.cpu().numpy(), g["Y_rk45_tight"], rtol=1e-9, atol=1e-11)
    # step budget exhausted: flagged, NaN rows, never garbage (the reference raises RuntimeError there; solve_custom mirrors that)
    Yf, stf, _ = eng.simulate_batch(X[:1], g["t_eval"], rtol=1e-5, atol=1e-7, max_steps=50, method="dp5")
    assert int(stf[0]) == 2 and np.isnan(Yf.cpu().numpy()[0, -1]).all() and np.isfinite(Yf.cpu().numpy()[0, 0]).all()
    eng.close()


def test_population_evaluation_matches_elementwise_restatement():
    """GlobalODEBatch.evaluate(X_raw) == the oracle's restatement of GlobalODE_MOO._evaluate applied candidate by candidate (on the GPU
    trajectories: the simulation itself is checked elsewhere), including the fail_value of a candidate whose simulation is flagged."""
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd.global_model.optproblem import GlobalODEBatch
    g = np.load([x for x in GOLD if x.name == "network_m0_small.npz"][0])
    gl = np.load(Path(__file__).resolve().parent / "golden" / "network_loss_m0.npz")
    eng = NetworkEngine.from_npz(g)
    net = nm.Network.from_npz(g)
    ld = {k: gl[k] for k in gl.files}
    keys = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i", "tf_scale")
    defaults = {k: (g[k][0] if k != "tf_scale" else float(g[k][0])) for k in keys}
    lam = dict(protein=1.0, rna=2.0, phospho=0.5, prior=0.3)
    prob = GlobalODEBatch(eng, None, ld, defaults, lam, g["t_eval"], rtol=1e-7, atol=1e-9)
    Xphys = np.stack([_x(eng, g, k) for k in range(4)])
    Xraw = np.log(np.expm1(np.maximum(Xphys, 1e-12)))                 # inv_softplus (utils.py:245-253)
    Xraw[3, 2] = np.nan                                               # a broken candidate
    F = prob.evaluate(Xraw)
    assert F.shape == (4, 3) and (F[3] == 1e12).all()
    Y, _, _ = eng.simulate_batch(Xraw[:3], g["t_eval"], raw=True, rtol=1e-7, atol=1e-9)
    Y = Y.cpu().numpy()
    dflt = eng.pack_params(*(defaults[k] for k in keys))
    for k in range(3):
        xp = np.where(Xraw[k] > 20, Xraw[k], np.log1p(np.exp(Xraw[k])))
        want = nm.objectives(net, xp, dflt, Y[k], ld, 0, lam)
        np.testing.assert_allclose(F[k], want, rtol=1e-10)
    prob.close(); eng.close()


def test_combinatorial_blocks_beyond_three_sites():
    """The combinatorial topology with up to 5 sites per protein (33-state blocks; global_model/models.py:323-485 has no block-size limit).
    Round 1 had only the explicit RK45 twin for them; now the production W-method integrates them too (general LDS kernel, the block's
    two triangular sweeps done serially by the protein's thread).  Implicit path: against the oracle's LSODA at 1e-12 (same band as every
    other topology) and, for blocks <= 3 sites, against the register kernel.  Explicit path: step-for-step against the oracle's RK45."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    d = synthetic.make_network(N=5, total_sites=16, n_K=4, n_tf_edges=8, model=2, seed=9, max_sites=5)
    assert d["n_sites"].max() >= 4
    eng = NetworkEngine(**d)
    net = nm.Network(model=2, N=5, n_K=4, total_sites=16, S=eng.S, n_states=(1 << d["n_sites"].astype(np.int64)), **{k: d[k] for k in d if k != "model"})
    X = synthetic.random_candidates(d, 2, seed=4)
    t = np.array([0.0, 0.5, 1.0, 4.0, 16.0, 60.0, 240.0, 960.0])
    y0 = nm.default_y0(net)
    np.testing.assert_array_equal(eng.default_y0(), y0)
    Y, st, ns = eng.simulate_batch(X, t, rtol=1e-5, atol=1e-7, max_steps=2_000_000, method="dp5")
    assert not st.cpu().numpy().any()
    Yw, stw, nsw = eng.simulate_batch(X, t)                      # Rosenbrock-W, defaults (1e-7 / 1e-9)
    assert not stw.cpu().numpy().any()
    band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
    for k in range(2):
        nK, N, sites = 4, 5, 16
        x = X[k]
        p = nm.Params(c_k=x[:nK], A_i=x[nK:nK + N], B_i=x[nK + N:nK + 2 * N], C_i=x[nK + 2 * N:nK + 3 * N], D_i=x[nK + 3 * N:nK + 4 * N],
                      Dp_i=x[nK + 4 * N:nK + 4 * N + sites], E_i=x[nK + 4 * N + sites:nK + 5 * N + sites], tf_scale=float(x[-1]))
        Yo, acc, rej = nm.simulate_rk45(net, p, t, 1e-5, 1e-7, y0=y0, return_steps=True)
        np.testing.assert_allclose(Y[k].cpu().numpy(), Yo, rtol=1e-9, atol=1e-11)
        assert tuple(ns[k].cpu().numpy()) == (acc, rej)
        truth = nm.simulate_odeint(net, p, t, 1e-12, 1e-12, 500000, y0=y0)
        assert band(Yw[k].cpu().numpy(), truth) <= 0.5, (k, band(Yw[k].cpu().numpy(), truth))
    eng.close()
    # <= 3 sites: the LDS kernel (linsolve = structured) and the register kernel implement the same scheme
    g = np.load([x for x in GOLD if x.name == "network_m2_medium.npz"][0])
    e3 = NetworkEngine.from_npz(g)
    X3 = np.stack([_x(e3, g, k) for k in range(4)])
    Ya, sa, na = e3.simulate_batch(X3, g["t_eval"], kernel="auto")
    Yb, sb, nb = e3.simulate_batch(X3, g["t_eval"], kernel="lds")
    assert not sa.cpu().numpy().any() and not sb.cpu().numpy().any()
    assert band(Yb.cpu().numpy(), Ya.cpu().numpy()) <= 0.2
    assert band(Yb.cpu().numpy()[:2], g["Y_tight"]) <= 0.5
    # the additive order-4 kernel exists for this topology too (on request: its step costs more than it saves here): its implicit operator
    # is the approximate factorisation itself; same trajectories, >= 2x fewer steps
    Yk, sk, nk = e3.simulate_batch(X3, g["t_eval"], rtol=1e-8, atol=1e-8, method="ark")
    Yr, sr, nr = e3.simulate_batch(X3, g["t_eval"], rtol=1e-8, atol=1e-8, method="rosw")
    assert not sk.cpu().numpy().any() and band(Yk.cpu().numpy()[:2], g["Y_tight"]) <= 0.3
    assert 2 * int(nk[:, 0].sum()) <= int(nr[:, 0].sum())
    e3.close()


@pytest.mark.parametrize("model", [0, 2])
def test_full_size_network_config5_properties(model):
    """BASELINE config 5 shape at full size (8 192 candidates of the synthetic N = 100 / 300-site network, S = 500; 900 for the
    combinatorial topology) at the reference's production tolerance: unflagged, finite, first row = y0, batch-composition independence
    (a re-run of a shuffled subset is bit-identical), and a subsample inside the band of the parity-grade run (rtol 1e-7)."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    net = synthetic.make_network(model=model)
    eng = NetworkEngine(**net)
    B = 8192
    X = synthetic.random_candidates(net, B, seed=5)
    t_eval = np.unique(np.concatenate([net["kin_grid"], [15.0]]))
    opt = dict(rtol=1e-8, atol=1e-8)                            # the optimiser's settings (GlobalODEBatch): config.toml:403-404, strict norm
    Y, st, ns = eng.simulate_batch(X, t_eval, **opt)
    assert not st.cpu().numpy().any()
    assert bool(torch.isfinite(Y).all())
    np.testing.assert_array_equal(Y[:, 0, :].cpu().numpy(), np.broadcast_to(eng.default_y0(), (B, eng.S)))
    pick = np.random.default_rng(1).choice(B, 64, replace=False)
    Y2, _, _ = eng.simulate_batch(X[pick], t_eval, **opt)
    assert torch.equal(Y2, Y[torch.as_tensor(pick, device=Y.device)])
    Yt, stt, _ = eng.simulate_batch(X[pick[:16]], t_eval, rtol=1e-9, atol=1e-10, err_norm="max")
    assert not stt.cpu().numpy().any()
    a, b = Y2[:16].cpu().numpy(), Yt.cpu().numpy()
    # against a run 10-100x tighter under the strict norm: well inside the parity band (rtol 1e-6 / atol 1e-8)
    assert np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))) <= 0.5
    # ODEPACK's RMS norm at the same tolerances is NOT parity-safe on such populations (measured: 0.3 distributive, 2.1 combinatorial
    # band widths): it stays an opt-in
    Yr, _, nr = eng.simulate_batch(X[pick[:16]], t_eval, err_norm="rms", method="rosw", **opt)
    assert np.max(np.abs(Yr.cpu().numpy() - b) / (1e-8 + 1e-6 * np.abs(b))) <= 5.0
    # the settings behind simulate_and_measure's hard-wired 1e-5 / 1e-7: a few of THAT band's widths -- measured 3.4 (order 3, distributive),
    # 14.9 (combinatorial) in round 1; the order-4 method runs a decade tighter for it
    from phoskintime_amd.global_model.simulate import measure_tolerances
    Ys, _, _ = eng.simulate_batch(X[pick[:16]], t_eval, **measure_tolerances(eng))
    assert np.max(np.abs(Ys.cpu().numpy() - b) / (1e-7 + 1e-5 * np.abs(b))) <= (5.0 if model == 0 else 25.0)
    eng.close()


def test_full_size_network_config4_morris():
    """BASELINE config 4 size: 128 Morris trajectories x (841 varied parameters + 1) = 107 776 network simulations in one batch -> fold-change
    observables -> scalar metric -> elementary effects.  Size-independent checks: nothing flagged, every parameter gets a finite mu*,
    the Morris output of the unperturbed centre equals the direct evaluation, and mu* is invariant under a permutation of trajectories."""
    import time
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    from phoskintime_amd.global_model import sensitivity as gs
    net = synthetic.make_network(model=0)
    eng = NetworkEngine(**net)
    x0 = synthetic.default_candidate(net)
    nK, N, sites = eng.n_K, eng.N, eng.total_sites
    fitted = dict(c_k=x0[:nK], A_i=x0[nK:nK + N], B_i=x0[nK + N:nK + 2 * N], C_i=x0[nK + 2 * N:nK + 3 * N], D_i=x0[nK + 3 * N:nK + 4 * N],
                  Dp_i=x0[nK + 4 * N:nK + 4 * N + sites], E_i=x0[nK + 4 * N + sites:nK + 5 * N + sites], tf_scale=float(x0[-1]))
    tp = net["kin_grid"]; tr = np.array([4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
    t0 = time.perf_counter()
    out = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, perturbation=0.05, trajectories=128, num_levels=40, seed=3)      # BASELINE config 4: 128 trajectories
    wall = time.perf_counter() - t0
    B = out["param_values"].shape[0]
    assert B == 128 * (eng.n_var + 1) and not out["status"].any() and np.isfinite(out["Y"]).all()
    Si = out["Si"]
    assert np.isfinite(Si["mu_star"]).all() and (Si["mu_star"] >= 0).all() and Si["mu_star"].max() > 0
    # permuting whole trajectories changes nothing in mu / mu_star
    D = eng.n_var
    perm = np.random.default_rng(0).permutation(128)
    Xp = out["param_values"].reshape(128, D + 1, D)[perm].reshape(-1, D)
    Yp = out["Y"].reshape(128, D + 1)[perm].reshape(-1)
    from phoskintime_amd.sensitivity import morris
    Sp = morris.analyze(out["problem"], Xp, Yp, num_levels=40, num_resamples=0)
    np.testing.assert_allclose(Sp["mu_star"], Si["mu_star"], rtol=1e-12, atol=1e-14)
    print("config-4-sized network Morris: %d simulations, wall %.2f s" % (B, wall))
    eng.close()


def test_network_config4_named_shape_128_trajectories_x_200_parameters():
    """BASELINE config 4 in its NAMED shape (VERDICT r2 missing #6): 128 trajectories x 200 varied parameters = 25 728 simulations, the other
    entries of the parameter vector fixed at their fitted values (`vary=`).  Size-independent properties: the sample matrix has 200
    columns; every simulated candidate equals the fitted vector outside the varied entries; rows re-simulated one by one give the same Y;
    the varied design restricted to a parameter set equals the full design's elementary effects for those parameters when the SAME unit-cube
    design is used; a parameter given by NAME selects the same column as its index."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    from phoskintime_amd.global_model import sensitivity as gs
    from phoskintime_amd.global_model.simulate import measure_tolerances
    net = synthetic.make_network(model=0)
    eng = NetworkEngine(**net)
    x0 = synthetic.default_candidate(net)
    nK, N, sites = eng.n_K, eng.N, eng.total_sites
    fitted = dict(c_k=x0[:nK], A_i=x0[nK:nK + N], B_i=x0[nK + N:nK + 2 * N], C_i=x0[nK + 2 * N:nK + 3 * N], D_i=x0[nK + 3 * N:nK + 4 * N],
                  Dp_i=x0[nK + 4 * N:nK + 4 * N + sites], E_i=x0[nK + 4 * N + sites:nK + 5 * N + sites], tf_scale=float(x0[-1]))
    tp = net["kin_grid"]; tr = np.array([4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
    vary = np.sort(np.random.default_rng(4).choice(eng.n_var, size=200, replace=False))
    out = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, perturbation=0.05, trajectories=128, num_levels=40, seed=3, vary=vary, return_pred=True)
    X, Y = out["param_values"], out["Y"]
    assert X.shape == (128 * 201, 200) and Y.shape == (25728,) and not out["status"].any() and np.isfinite(Y).all()
    assert len(out["Si"]["mu_star"]) == 200 and np.isfinite(out["Si"]["mu_star"]).all() and out["Si"]["mu_star"].max() > 0
    full_names = gs.compute_bounds({k: (np.asarray(fitted[k], float) if k != "tf_scale" else fitted[k]) for k in gs._ORDER}, 0.05)["names"]
    assert out["problem"]["names"] == [full_names[i] for i in vary]
    lo = x0[vary] * 0.95; hi = x0[vary] * 1.05
    assert (X >= np.minimum(lo, hi) - 1e-12).all() and (X <= np.maximum(lo, hi) + 1e-12).all()
    # three rows re-simulated on their own: scatter the varied entries into the fitted vector, simulate, measure
    times = np.unique(np.concatenate([tp, tr, tp]).astype(np.float64))
    lists, ld = eng.make_index_lists(times, tp, tr, tp)
    for r in (0, 7777, 25727):
        x = x0.copy(); x[vary] = X[r]
        Yr, st, _ = eng.simulate_batch(x[None], times, max_steps=5000 * times.size, **measure_tolerances(eng))
        pred = eng.observables_batch(lists, Yr, ld["p_prot"].size + ld["p_rna"].size + ld["p_pho"].size, eps=1e-12)
        assert float(pred.sum()) == pytest.approx(Y[r], rel=1e-12)
    eng.free_loss(lists)
    # by name: same design as by index
    sub = [full_names[i] for i in vary[:5]]
    a = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, perturbation=0.05, trajectories=4, num_levels=8, seed=9, vary=sub)
    b = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, perturbation=0.05, trajectories=4, num_levels=8, seed=9, vary=vary[:5])
    np.testing.assert_array_equal(a["param_values"], b["param_values"]); np.testing.assert_array_equal(a["Y"], b["Y"])
    with pytest.raises(ValueError):
        gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, trajectories=2, num_levels=4, seed=1, vary=[3, 3])
    with pytest.raises(ValueError):
        gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, trajectories=2, num_levels=4, seed=1, vary=[eng.n_var])
    eng.close()


def test_frechet_distance_kernel_and_population_pick():
    """pk_frechet_batch against the reference's frechet_distance outputs (tests/golden/frechet.npz), batched series against the oracle,
    and the whole Pareto pick (simulate -> fold changes -> per-series Frechet -> weighted sum -> argmin) against a host composition."""
    from phoskintime_amd.frechet import frechet_distance, frechet_batch
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd.global_model.pick import frechet_pick_batch
    g = np.load(GOLD[0].parent / "frechet.npz")
    K = g["dist"].size
    for k in range(K):
        assert abs(frechet_distance(g[f"a{k}"], g[f"b{k}"]) - g["dist"][k]) <= 1e-12 * max(1.0, g["dist"][k])
    # all K series in one launch, 3 candidates whose predicted values differ
    rng = np.random.default_rng(2)
    obs = [g[f"a{k}"] for k in range(K)]
    pt = [g[f"b{k}"][:, 0] for k in range(K)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in pt])])
    pidx = [offs[k] + np.arange(len(pt[k])) for k in range(K)]
    base = np.concatenate([g[f"b{k}"][:, 1] for k in range(K)])
    pred = np.stack([base, base * 1.3, rng.uniform(0.2, 3.0, base.size)])
    out = frechet_batch(obs, pt, pidx, pred).cpu().numpy()
    np.testing.assert_allclose(out[0], g["dist"], rtol=1e-12)
    for b in (1, 2):
        want = [nm.frechet_distance(obs[k], np.stack([pt[k], pred[b, pidx[k]]], axis=1)) for k in range(K)]
        np.testing.assert_allclose(out[b], want, rtol=1e-12)
    with pytest.raises(ValueError):
        frechet_batch([np.zeros((40, 2))], [np.zeros(3)], [np.arange(3)], np.zeros((1, 3)))
    # population pick on a golden network
    gn = np.load([x for x in GOLD if x.name == "network_m0_small.npz"][0])
    eng = NetworkEngine.from_npz(gn)
    X = np.stack([_x(eng, gn, k) for k in range(4)])
    tp = gn["kin_grid"]; tr = np.array([4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
    times = np.unique(np.concatenate([tp, tr]))
    Y, _, _ = eng.simulate_batch(X, times, rtol=1e-5, atol=1e-7)
    Y = Y.cpu().numpy()
    fc = lambda y, base: np.maximum(y, 1e-12) / np.maximum(base, 1e-12)
    sel = lambda tt: np.where(np.isin(times, tt))[0]
    # observations = candidate 2's own curves with a few points dropped: candidate 2 must win with the smallest score
    obs_prot, obs_rna, obs_pho = {}, {}, {}
    oy, os_, nsites = gn["offset_y"], gn["offset_s"], gn["n_sites"]
    tot = lambda Yb, i: Yb[:, oy[i] + 1:oy[i] + 2 + nsites[i]].sum(axis=1)
    for i in range(int(gn["N"])):
        P = tot(Y[2], i)
        cur = np.stack([tp, fc(P[sel(tp)], P[0])], axis=1)
        obs_prot[i] = cur[::2] if i % 2 else cur
        R = Y[2][:, oy[i]]
        obs_rna[i] = np.stack([tr, fc(R[sel(tr)], R[np.where(times == 4.0)[0][0]])], axis=1)
        for j in range(int(nsites[i])):
            ph = Y[2][:, oy[i] + 2 + j]
            obs_pho[(i, j)] = np.stack([tp, fc(ph[sel(tp)], ph[0])], axis=1)
    res = frechet_pick_batch(eng, X, tp, tr, tp, obs_prot, obs_rna, obs_pho, lambdas=(1.0, 0.5, 2.0))
    assert res["best"] == 2 and res["per_series"].shape == (4, len(res["series"])) and not res["status"].any()
    # host composition for candidate 0
    want = 0.0
    for (kind, key), w in zip(res["series"], [1.0 if s[0] == "prot" else 0.5 if s[0] == "rna" else 2.0 for s in res["series"]]):
        if kind == "prot":
            P = tot(Y[0], key); pc = np.stack([tp, fc(P[sel(tp)], P[0])], axis=1); oc = obs_prot[key]
        elif kind == "rna":
            R = Y[0][:, oy[key]]; pc = np.stack([tr, fc(R[sel(tr)], R[np.where(times == 4.0)[0][0]])], axis=1); oc = obs_rna[key]
        else:
            ph = Y[0][:, oy[key[0]] + 2 + key[1]]; pc = np.stack([tp, fc(ph[sel(tp)], ph[0])], axis=1); oc = obs_pho[key]
        want += w * nm.frechet_distance(oc, pc)
    assert abs(res["scores"][0] - want) <= 1e-9 * max(1.0, want)
    eng.close()


def test_network_beyond_the_register_kernel_limits_runs_through_the_lds_kernel():
    """N = 300 proteins (> 256) and S = 1000 states: the register kernel is not eligible, the LDS kernel must take over and agree with the
    explicit DP5 path (right-hand sides only) run at a tight tolerance."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    net = synthetic.make_network(N=300, total_sites=400, n_K=60, n_tf_edges=700, model=0, seed=77)
    eng = NetworkEngine(**net)
    assert eng.N == 300 and eng.S == 1000
    X = synthetic.random_candidates(net, 8, seed=2)
    t = np.unique(np.concatenate([net["kin_grid"], [15.0]]))
    Y, st, _ = eng.simulate_batch(X, t, rtol=1e-7, atol=1e-9)
    Yd, std, _ = eng.simulate_batch(X[:2], t, rtol=1e-9, atol=1e-11, max_steps=2_000_000, method="dp5")
    assert not st.cpu().numpy().any() and not std.cpu().numpy().any()
    a, b = Y[:2].cpu().numpy(), Yd.cpu().numpy()
    assert np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))) <= 0.5
    eng.close()


LARGE = sorted((Path(__file__).resolve().parent / "golden").glob("netlarge_m[0-9].npz"))


@pytest.mark.parametrize("f", LARGE, ids=lambda f: f.stem)
def test_large_network_against_the_reference_run(f):
    """BASELINE config 4 / 5 size (N = 100, S ~ 550) against trajectories the REFERENCE produced for this very network (VERDICT r1 missing
    #4: round 1 checked S = 500 only against itself): RHS at the probe times, the parity-grade run against LSODA at 1e-12, and the run at
    the optimiser's tolerance (rtol = atol = 1e-8, config.toml:403-404) against the same truth -- no worse than the reference's own 1e-8 run."""
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(f)
    eng = NetworkEngine.from_npz(g)
    assert eng.S == int(g["S"])
    X = np.stack([_x(eng, g, k) for k in range(2)])
    for ti, t in enumerate(g["t_probe"]):
        dr = eng.rhs_batch(X, g["y_rand"], float(t)).cpu().numpy()
        scale = 1.0 + np.abs(g["rhs_rand"][:, ti]).max()
        np.testing.assert_allclose(dr, g["rhs_rand"][:, ti], rtol=1e-12, atol=1e-13 * scale)
    band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
    truth = g["Y_tight"][0]
    opt = dict(rtol=1e-8, atol=1e-8)                                                         # the optimiser's tolerances (config.toml:403-404)
    Yo, so, no = eng.simulate_batch(X, g["t_eval"], **opt)                                   # default integrator (order 4 where it applies)
    Yp, sp, npp = eng.simulate_batch(X, g["t_eval"], method="rosw", **opt)                   # round 1's order-3 Rosenbrock-W
    Yr, sr, nrr = eng.simulate_batch(X, g["t_eval"], method="rosw", err_norm="rms", **opt)   # ... under ODEPACK's RMS norm
    assert not sp.cpu().numpy().any() and not so.cpu().numpy().any() and not sr.cpu().numpy().any()
    ref_own = band(g["Y_lsoda8"][0], truth)
    e_par, e_opt, e_rms = band(Yp[0].cpu().numpy(), truth), band(Yo[0].cpu().numpy(), truth), band(Yr[0].cpu().numpy(), truth)
    print(f"{f.name} at 1e-8/1e-8: default {e_opt:.3f} ({int(no[0, 0])} steps) | ROS34PW2 {e_par:.3f} ({int(npp[0, 0])} steps), RMS norm {e_rms:.3f} ({int(nrr[0, 0])} steps)"
          f" | reference LSODA 1e-8: {ref_own:.3f}")
    assert e_par <= 0.1 and e_opt <= 0.15 and e_rms <= 0.5
    assert e_opt <= max(0.2, 2.5 * ref_own)                     # same nominal tolerance: the reference run's own accuracy class
    if eng.ark_eligible():
        # >= 2x fewer steps at 1e-8 with parity held -- measured 3.5-4.1x at the SAME band error as the order-3 method
        assert 3 * int(no[0, 0]) <= int(npp[0, 0])
    assert int(nrr[0, 0]) <= 0.7 * int(npp[0, 0])
    for k in range(2):                                                                 # both reference runs at 1e-8: within its error + ours
        assert band(Yo[k].cpu().numpy(), g["Y_lsoda8"][k]) <= ref_own + e_opt + 1.0
    eng.close()


MORE = sorted((Path(__file__).resolve().parent / "golden").glob("netlarge_more_m[0-9].npz"))


@pytest.mark.parametrize("f", MORE, ids=lambda f: f.stem)
def test_large_network_population_against_reference_runs(f):
    """VERDICT r2 (population parity must not be self-referential): EVERY candidate of netlarge_more_m<M>.npz -- the fixture's second
    parameter set, log-normal draws around the optimiser's defaults (bench.py's config-5 population shape) and wide uniform draws, tf_scale
    from 0.04 to 4 -- was integrated by the reference's simulate_odeint at rtol = atol = 1e-12 (tools/make_golden_network.py large_more).
    The default integrator and the order-3 Rosenbrock-W at the optimiser's tolerance (1e-8 / 1e-8) stay inside the parity band against
    each of them; the reference's own LSODA run at 1e-8 sits at 0.10-0.25 of the band on the candidate that has one."""
    from phoskintime_amd.global_model import NetworkEngine
    q = np.load(f)
    g = np.load(f.parent / f"netlarge_m{int(q['model'])}.npz")
    eng = NetworkEngine.from_npz(g)
    K = int(q["done"])
    assert K >= 3 and q["Y_tight"].shape == (K, q["t_eval"].size, eng.S) and np.isfinite(q["Y_tight"]).all()
    X = np.stack([_x(eng, q, k) for k in range(K)])
    band = lambda a, b: float(np.max(np.abs(a - b) / (1e-8 + 1e-6 * np.abs(b))))
    opt = dict(rtol=1e-8, atol=1e-8)
    Yo, so, no = eng.simulate_batch(X, q["t_eval"], **opt)
    Yp, sp, npp = eng.simulate_batch(X, q["t_eval"], method="rosw", **opt)
    Yt, stt, ntt = eng.simulate_batch(X, q["t_eval"], rtol=1e-10, atol=1e-10)
    assert not so.cpu().numpy().any() and not sp.cpu().numpy().any() and not stt.cpu().numpy().any()
    e_o = [band(Yo[k].cpu().numpy(), q["Y_tight"][k]) for k in range(K)]
    e_p = [band(Yp[k].cpu().numpy(), q["Y_tight"][k]) for k in range(K)]
    e_t = [band(Yt[k].cpu().numpy(), q["Y_tight"][k]) for k in range(K)]
    ref_own = band(g["Y_lsoda8"][1], q["Y_tight"][0])           # candidate 0 is the fixture's second parameter set: the reference's own 1e-8 run
    print(f"{f.name}: {K} candidates, default {np.round(e_o, 3).tolist()} in {no[:, 0].cpu().numpy().tolist()} steps | ROS34PW2 {np.round(e_p, 3).tolist()} | "
          f"default at 1e-10 {np.round(e_t, 4).tolist()} | reference LSODA 1e-8 on candidate 0: {ref_own:.3f}")
    assert max(e_o) <= 0.15 and max(e_p) <= 0.15 and max(e_t) <= 0.01      # measured over all 28 candidates: <= 0.116, <= 0.080, <= 0.001
    assert e_o[0] <= max(0.1, ref_own)
    if eng.ark_eligible():
        assert (3 * no[:, 0].cpu().numpy() <= 2 * npp[:, 0].cpu().numpy()).all()             # >= 1.5x fewer steps on every candidate (measured 1.7-3.7x)
    eng.close()


_ARK2_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from phoskintime_amd.global_model import NetworkEngine
out = {}
for name in ("network_m2_small", "netlarge_m2"):
    g = np.load(sys.argv[1] + "/tests/golden/" + name + ".npz")
    eng = NetworkEngine.from_npz(g)
    K = g["Y_tight"].shape[0]
    X = np.stack([eng.pack_params(g["c_k"][k], g["A_i"][k], g["B_i"][k], g["C_i"][k], g["D_i"][k], g["Dp_i"][k], g["E_i"][k], g["tf_scale"][k]) for k in range(K)])
    assert eng.resolved_method() == "ark"
    Y, st, ns = eng.simulate_batch(X, g["t_eval"], rtol=1e-8, atol=1e-8)
    out[name + "_Y"] = Y.cpu().numpy(); out[name + "_ns"] = ns.cpu().numpy(); out[name + "_st"] = st.cpu().numpy()
    eng.close()
np.savez(sys.argv[2], **out)
"""


def test_combinatorial_topology_default_is_the_additive_method_on_the_exact_block_solve(tmp_path):
    """[r3] VERDICT r2 item 4 (ii): ARK436 is the default integrator of the combinatorial topology, its implicit operator the block Jacobian
    itself, solved exactly by parity elimination of the bit-pattern block.  Against round 2's approximate factorisation as implicit operator
    (PK_ARK2_EXACT=0, read once per process: child processes): same trajectories far inside the band of the reference's LSODA at 1e-12, at
    most 0.6x the steps (measured 0.5x; the order-3 Rosenbrock-W default of round 2 took 4.6x the steps)."""
    import os, subprocess, sys
    root = str(Path(__file__).resolve().parents[1])
    res = {}
    for tag, env in (("exact", {}), ("approx", {"PK_ARK2_EXACT": "0"})):
        f = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", _ARK2_SCRIPT, root, str(f)], check=True, env={**os.environ, **env}, timeout=600)
        res[tag] = np.load(f)
    for name in ("network_m2_small", "netlarge_m2"):
        g = np.load(Path(root) / "tests" / "golden" / f"{name}.npz")
        for tag in ("exact", "approx"):
            Y = res[tag][name + "_Y"]
            assert not res[tag][name + "_st"].any()
            for k in range(g["Y_tight"].shape[0]):
                assert np.max(np.abs(Y[k] - g["Y_tight"][k]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][k]))) <= 0.1, (name, tag, k)
        assert (res["exact"][name + "_ns"][:, 0] <= 0.6 * res["approx"][name + "_ns"][:, 0]).all(), (res["exact"][name + "_ns"], res["approx"][name + "_ns"])


_ARKP_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from phoskintime_amd.global_model import NetworkEngine, synthetic
out = {}
t = np.array([0.0, 0.5, 1.0, 2.0, 4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
for model in (0, 1, 4):
    for cap, N, sites in ((4, 40, 90), (6, 100, 330), (8, 70, 300), (8, 180, 640)):
        net = synthetic.make_network(N=N, total_sites=sites, n_K=20, n_tf_edges=3 * N, model=model, seed=7 + cap + N, max_sites=cap)
        eng = NetworkEngine(**net)
        assert eng.resolved_method() == "ark"
        X = synthetic.random_candidates(net, 24, seed=cap + model, spread=0.6)
        Y, st, ns = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8)
        key = "m%d_c%d_N%d" % (model, cap, N)
        out[key + "_Y"] = Y.cpu().numpy(); out[key + "_ns"] = ns.cpu().numpy(); out[key + "_st"] = st.cpu().numpy()
        eng.close()
np.savez(sys.argv[2], **out)
"""


def test_dense_lane_layout_of_the_additive_integrator_against_the_thread_per_protein_kernel(tmp_path):
    """[r3] VERDICT r2 item 4 (i): the arrow topologies run the additive method in the dense two-lanes-per-protein layout
    (pk_network_solve_arkp.hpp: registers only, or -- the default -- on the register diet for 3 waves per SIMD).  Same method, same
    controller: against round 2's kernel (PK_ARK_PAIR=0, read once per process: child processes) the trajectories agree to rounding
    (the site sums run in another order) and the step counts are equal up to the odd step decided by that rounding, for every site class
    (3 / 4 / 5 rows per lane), both topologies, single-lane and paired proteins, TF rows longer than the register-resident entries,
    and a network that needs more than 256 lanes."""
    import os, subprocess, sys
    root = str(Path(__file__).resolve().parents[1])
    res = {}
    for tag, env in (("diet", {"PK_ARK_PAIR": "3"}), ("regs", {"PK_ARK_PAIR": "1"}), ("old", {"PK_ARK_PAIR": "0"})):
        f = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", _ARKP_SCRIPT, root, str(f)], check=True, env={**os.environ, **env}, timeout=600)
        res[tag] = np.load(f)
    keys = [k[:-2] for k in res["old"].files if k.endswith("_Y")]
    assert len(keys) == 12
    for key in keys:
        Yo, no = res["old"][key + "_Y"], res["old"][key + "_ns"]
        assert not res["old"][key + "_st"].any() and np.isfinite(Yo).all()
        for tag in ("diet", "regs"):
            Y, ns = res[tag][key + "_Y"], res[tag][key + "_ns"]
            assert not res[tag][key + "_st"].any(), (key, tag)
            band = np.max(np.abs(Y - Yo) / (1e-8 + 1e-6 * np.abs(Yo)))
            assert band <= 0.02, (key, tag, band)                                        # both sit ~0.05 from the truth; from each other: rounding
            assert np.max(np.abs(ns.sum(axis=1) - no.sum(axis=1))) <= 0.02 * no.sum(axis=1).max() + 2, (key, tag)
        np.testing.assert_array_equal(res["diet"][key + "_ns"], res["regs"][key + "_ns"])  # the two pair kernels: the same arithmetic
        np.testing.assert_array_equal(res["diet"][key + "_Y"], res["regs"][key + "_Y"])


@pytest.mark.parametrize("name", ["network_m0_small", "network_m1_small", "network_m4_small", "netlarge_m0", "netlarge_m1", "netlarge_m4"])
def test_fused_simulate_objective_equals_the_two_launch_path(name):
    """[r3] VERDICT r2 item 4 (iii) / SURVEY fused op (i): ``pk_network_simulate_objective_batch`` -- the integrator scores the observations
    at its output times, no trajectory in HBM -- against ``simulate_batch`` + ``objective_batch`` on the same candidates: the same loss
    sums and objectives for all eight LOSS_MODEs (up to the order of the sums), the reference's production baselines (protein / phospho at
    t = 0, rna at t = 4: runner.py:545-547), raw and physical candidates, the prior term, a candidate that fails (fail_value in every
    objective), batched y0, and the optional trajectory.  Loss data the fused path cannot take answer None (PK_ERR_UNSUPPORTED)."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(Path(__file__).resolve().parent / "golden" / f"{name}.npz")
    eng = NetworkEngine.from_npz(g)
    t = g["t_eval"]
    K = g["c_k"].shape[0]
    rng = np.random.default_rng(3)
    X = np.stack([_x(eng, g, k % K) for k in range(6)]) * np.exp(0.3 * rng.standard_normal((6, eng.n_var)))
    X[5, eng.n_K + eng.N: eng.n_K + 2 * eng.N] = np.nan                 # B_i = NaN: this candidate must come back as fail_value
    lists, ld = eng.make_index_lists(t, t, t[t >= 4.0], t)               # every protein / site at every time of its modality, rna from t = 4
    eng.free_loss(lists)
    assert ld["rna_base_idx"] > 0 and ld["prot_base_idx"] == 0
    for k in ("obs_prot", "obs_rna", "obs_pho"):
        ld[k] = np.abs(1.0 + 0.2 * rng.standard_normal(ld[k].size))
    for k in ("w_prot", "w_rna", "w_pho"):
        ld[k] = rng.uniform(0.5, 2.0, ld[k].size)
    loss = eng.make_loss(ld, t.size)
    defaults = X[0] * 1.1
    lam = (1.0, 0.5, 2.0, 0.7)
    opt = dict(rtol=1e-8, atol=1e-8)
    Y, st, ns = eng.simulate_batch(X, t, **opt)
    assert int(st[5]) != 0 and not st[:5].cpu().numpy().any()
    for mode in range(8):
        s2, F2 = eng.objective_batch(loss, Y, loss_mode=mode, x=X, defaults=defaults, lambdas=lam, status=st)
        out = eng.simulate_objective_batch(loss, X, t, loss_mode=mode, defaults=defaults, lambdas=lam, want_Y=(mode == 0), **opt)
        assert out is not None
        s1, F1, st1, ns1, Y1 = out
        np.testing.assert_array_equal(st1.cpu().numpy(), st.cpu().numpy()); np.testing.assert_array_equal(ns1.cpu().numpy(), ns.cpu().numpy())
        np.testing.assert_allclose(s1.cpu().numpy()[:5], s2.cpu().numpy()[:5], rtol=1e-11, equal_nan=True)
        np.testing.assert_allclose(F1.cpu().numpy(), F2.cpu().numpy(), rtol=1e-11, equal_nan=True)
        assert (F1[5].cpu().numpy() == 1e12).all()
        if mode == 0:
            np.testing.assert_array_equal(Y1.cpu().numpy()[:5], Y.cpu().numpy()[:5])
    # raw candidates + batched initial states
    Xraw = np.log(np.expm1(np.maximum(X[:5], 1e-12)))
    y0b = np.tile(g["y0"], (5, 1)) * rng.uniform(0.8, 1.2, size=(5, eng.S))
    Yb, stb, _ = eng.simulate_batch(Xraw, t, y0=y0b, raw=True, **opt)
    _, Fb2 = eng.objective_batch(loss, Yb, x=Xraw, raw=True, defaults=defaults, lambdas=lam, status=stb)
    _, Fb1, _, _, none = eng.simulate_objective_batch(loss, Xraw, t, y0=y0b, raw=True, defaults=defaults, lambdas=lam, **opt)
    assert none is None
    np.testing.assert_allclose(Fb1.cpu().numpy(), Fb2.cpu().numpy(), rtol=1e-11)
    eng.free_loss(loss)
    # an rna observation before its baseline, or a (state, time) observed twice: not fusable -- the caller falls back
    ld2 = dict(ld); ld2["t_rna"] = ld["t_rna"].copy(); ld2["t_rna"][0] = 0
    l2 = eng.make_loss(ld2, t.size)
    assert eng.simulate_objective_batch(l2, X[:2], t, **opt) is None
    eng.free_loss(l2)
    ld3 = {k: (np.concatenate([v, v[:1]]) if k.endswith("_prot") and isinstance(v, np.ndarray) else v) for k, v in ld.items()}
    l3 = eng.make_loss(ld3, t.size)
    assert eng.simulate_objective_batch(l3, X[:2], t, **opt) is None
    _, F3 = eng.objective_batch(l3, Y[:2], x=X[:2], defaults=defaults, lambdas=lam)     # ... and the two-launch path still takes them
    assert np.isfinite(F3.cpu().numpy()).all()
    eng.free_loss(l3)
    eng.close()


def test_fused_objective_is_what_the_optimisation_problem_runs():
    """GlobalODEBatch.evaluate_device takes the fused launch where the library offers it (distributive, sequential: the dense lane layout)
    and the two-launch path elsewhere (combinatorial: the thread-per-protein kernel) -- same F either way."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine
    from phoskintime_amd.global_model.optproblem import GlobalODEBatch
    for name, fused in (("network_m0_small", True), ("network_m1_small", True), ("network_m2_small", False)):
        g = np.load(Path(__file__).resolve().parent / "golden" / f"{name}.npz")
        eng = NetworkEngine.from_npz(g)
        t = g["t_eval"]
        lists, ld = eng.make_index_lists(t, t, t[t >= 4.0], t)
        eng.free_loss(lists)
        base = _x(eng, g, 0)
        keys = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")
        defaults = {k: g[k][0] for k in keys}; defaults["tf_scale"] = float(g["tf_scale"][0])
        prob = GlobalODEBatch(eng, None, ld, defaults, {"protein": 1.0, "rna": 1.0, "phospho": 1.0, "prior": 0.01}, t, rtol=1e-8, atol=1e-8)
        X = np.log(np.expm1(base[None, :] * np.exp(0.3 * np.random.default_rng(1).standard_normal((8, base.size)))))
        F = prob.evaluate_device(X).cpu().numpy()
        assert prob.fused is fused
        Y, st, _ = eng.simulate_batch(X, t, raw=True, rtol=1e-8, atol=1e-8, max_steps=prob.max_steps * t.size)
        _, F2 = eng.objective_batch(prob.loss, Y, x=X, raw=True, defaults=prob.defaults, lambdas=prob.lam, status=st)
        np.testing.assert_allclose(F, F2.cpu().numpy(), rtol=1e-11)
        prob.close(); eng.close()


def test_dense_lane_layout_takes_networks_beyond_256_proteins():
    """The thread-per-protein kernel stops at N = 256; the dense lane layout is bounded by its 512 lanes -- a 300-protein network (1 000
    states) runs the order-4 method by default now.  Against the order-3 Rosenbrock-W (the only integrator such a network had) at a
    tighter tolerance: inside the band, at a fraction of the steps."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    for model in (0, 1):
        net = synthetic.make_network(N=300, total_sites=400, n_K=30, n_tf_edges=700, model=model, seed=11, max_sites=4)
        eng = NetworkEngine(**net)
        assert eng.N == 300 and eng.S == 1000 and eng.resolved_method() == "ark"
        X = synthetic.random_candidates(net, 12, seed=2, spread=0.5)
        t = np.array([0.0, 1.0, 4.0, 15.0, 60.0, 240.0, 960.0])
        Ya, sa, na = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8)
        Yr, sr, nr = eng.simulate_batch(X, t, rtol=1e-9, atol=1e-10, method="rosw")
        assert not sa.cpu().numpy().any() and not sr.cpu().numpy().any()
        band = float(((Ya - Yr).abs() / (1e-8 + 1e-6 * Yr.abs())).max())
        assert band <= 0.2, band
        assert (3 * na[:, 0].cpu().numpy() <= nr[:, 0].cpu().numpy()).all()
        eng.close()


def test_fused_objective_edge_shapes():
    """One candidate, the shortest grids: T = 1 (only the initial time: every fold change is 1) and T = 2; an empty batch."""
    import torch
    from phoskintime_amd.global_model import NetworkEngine
    g = np.load(Path(__file__).resolve().parent / "golden" / "network_m0_small.npz")
    eng = NetworkEngine.from_npz(g)
    X = _x(eng, g, 0)[None, :]
    for t in (np.array([0.0]), np.array([0.0, 7.5])):
        lists, ld = eng.make_index_lists(t, t, t[-1:], t)                # rna from its baseline on (the baseline is the grid point nearest t = 4)
        eng.free_loss(lists)
        rng = np.random.default_rng(0)
        for k in ("obs_prot", "obs_rna", "obs_pho"):
            ld[k] = rng.uniform(0.5, 1.5, ld[k].size)
        loss = eng.make_loss(ld, t.size)
        Y, st, ns = eng.simulate_batch(X, t, rtol=1e-8, atol=1e-8)
        s2, F2 = eng.objective_batch(loss, Y, x=X, defaults=X[0], lambdas=(1.0, 1.0, 1.0, 0.5), status=st)
        out = eng.simulate_objective_batch(loss, X, t, rtol=1e-8, atol=1e-8, defaults=X[0], lambdas=(1.0, 1.0, 1.0, 0.5))
        assert out is not None
        np.testing.assert_allclose(out[0].cpu().numpy(), s2.cpu().numpy(), rtol=1e-12)
        np.testing.assert_allclose(out[1].cpu().numpy(), F2.cpu().numpy(), rtol=1e-12)
        if t.size == 1:                                                   # pred = 1 everywhere: the sums are the weighted squared distances of the observations from 1
            want = [float(np.sum(ld["w_" + m] * (ld["obs_" + m] - 1.0) ** 2)) for m in ("prot", "rna", "pho")]
            assert ld["rna_base_idx"] == 0
            np.testing.assert_allclose(out[0].cpu().numpy()[0], want, rtol=1e-12)
        empty = eng.simulate_objective_batch(loss, np.zeros((0, eng.n_var)), t, rtol=1e-8, atol=1e-8)
        assert empty is not None and empty[1].shape == (0, 3)
        eng.free_loss(loss)
    eng.close()
