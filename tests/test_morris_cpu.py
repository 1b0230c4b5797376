"""CPU known-answer tests of the Morris sampler / analyser (SALib is absent: SURVEY.md section 8c-4) and of the host-side
sensitivity helpers against the oracle restatement."""
import numpy as np
import pytest

from phoskintime_amd.sensitivity import morris, compute_bound, define_sensitivity_problem_ds, define_sensitivity_problem_rand, _compute_Y
from oracle import protein_models as pm


def _problem(D, lo=0.0, hi=1.0):
    return {"num_vars": D, "names": [f"x{i}" for i in range(D)], "bounds": [[lo, hi]] * D}


def test_sample_structure():
    D, N, p = 5, 7, 4
    X = morris.sample(_problem(D, -2.0, 3.0), N, num_levels=p, seed=1)
    assert X.shape == (N * (D + 1), D)
    assert X.min() >= -2.0 and X.max() <= 3.0
    U = (X + 2.0) / 5.0
    levels = np.arange(p) / (p - 1)
    assert np.allclose(np.abs(U[:, :, None] - levels[None, None, :]).min(axis=2), 0, atol=1e-12)      # on the level grid
    T = U.reshape(N, D + 1, D)
    d = np.diff(T, axis=1)
    assert ((np.abs(d) > 1e-12).sum(axis=2) == 1).all()                                              # one-at-a-time
    assert np.allclose(np.abs(d).max(axis=2), p / (2.0 * (p - 1)))                                   # jump = delta
    moved = np.argmax(np.abs(d), axis=2)
    assert all(sorted(r) == list(range(D)) for r in moved)                                            # each coordinate once
    assert np.array_equal(X, morris.sample(_problem(D, -2.0, 3.0), N, num_levels=p, seed=1))          # seeded


def test_linear_function_known_answer():
    D = 6
    a = np.array([3.0, -2.0, 0.0, 0.5, 10.0, -1.0])
    prob = _problem(D, 0.0, 2.0)
    X = morris.sample(prob, 40, num_levels=6, seed=3)
    Y = X @ a + 7.0
    Si = morris.analyze(prob, X, Y, num_levels=6, seed=0)
    # inputs are rescaled to the unit cube: EE_i = a_i * (ub - lb) exactly, sigma = 0
    np.testing.assert_allclose(Si["mu"], a * 2.0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Si["mu_star"], np.abs(a) * 2.0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Si["sigma"], 0.0, atol=1e-10)
    np.testing.assert_allclose(Si["mu_star_conf"], 0.0, atol=1e-10)


def test_product_function_has_interaction_sigma():
    prob = _problem(3)
    X = morris.sample(prob, 200, num_levels=4, seed=5)
    Y = X[:, 0] * X[:, 1] + 0.0 * X[:, 2]
    Si = morris.analyze(prob, X, Y, num_levels=4, seed=0)
    assert Si["sigma"][0] > 0.1 and Si["sigma"][1] > 0.1 and Si["sigma"][2] == 0.0 and Si["mu_star"][2] == 0.0
    assert Si["mu_star_conf"][0] > 0.0


def test_hand_made_trajectory():
    prob = _problem(2)
    X = np.array([[0.0, 0.0], [2 / 3, 0.0], [2 / 3, 2 / 3]])
    Y = np.array([1.0, 2.0, 0.0])
    ee = morris.elementary_effects(prob, X, Y)
    np.testing.assert_allclose(ee, [[1.5, -3.0]])
    with pytest.raises(ValueError):
        morris.elementary_effects(prob, X[:2], Y[:2])


def test_scaled_option():
    prob = _problem(2, 0.0, 4.0)
    X = morris.sample(prob, 30, num_levels=4, seed=2)
    Y = 2.0 * X[:, 0] + X[:, 1]
    raw = morris.analyze(prob, X, Y, seed=0)
    sc = morris.analyze(prob, X, Y, scaled=True, seed=0)
    sx = np.std(X / 4.0, axis=0); sy = np.std(Y)
    np.testing.assert_allclose(sc["mu"], raw["mu"] * sx / sy, rtol=1e-12)


def test_bounds_and_problem_definitions_match_oracle():
    for v in (0.0, 1e-7, 2.0, -2.0, 19.0):
        assert compute_bound(v) == pm.compute_bound(v)
    p = define_sensitivity_problem_ds(3, list(np.arange(10.0)))
    assert p["num_vars"] == 10 and p["names"] == ['A', 'B', 'C', 'D', 'S1', 'S2', 'S3', 'D1', 'D2', 'D3']
    assert p["bounds"][0] == [0.0, 0.1] and p["bounds"][4] == [2.0, 6.0]
    pr = define_sensitivity_problem_rand(2, list(np.ones(9)))
    assert pr["num_vars"] == 9 and pr["names"][-3:] == ['D1', 'D2', 'D12']
    with pytest.raises(AssertionError):
        define_sensitivity_problem_ds(3, [1.0] * 9)


def test_compute_Y_matches_oracle():
    rng = np.random.default_rng(0)
    sol = rng.uniform(0, 3, (14, 9))
    for metric in pm.METRICS:
        assert _compute_Y(sol, 4, metric) == pytest.approx(pm.compute_Y(sol, 4, metric), rel=1e-14)


def test_multistart_candidates_follow_the_reference_recipe():
    """normest.py:217-265: base first, n/3 clipped Gaussian jitters, stratified uniform for the rest; seeded by (seed + hash(gene))."""
    from phoskintime_amd.paramest import multistart_candidates
    lb, ub = np.zeros(5), np.full(5, 20.0)
    base = np.array([1.0, 2.0, 30.0, 4.0, 5.0])
    C1 = multistart_candidates("AKT1", base, lb, ub, n_starts=24, seed=42)
    assert C1.shape == (24, 5) and (C1 >= lb).all() and (C1 <= ub).all()
    np.testing.assert_array_equal(C1[0], np.clip(base, lb, ub))
    # restatement with the same RNG calls
    rng = np.random.default_rng(42 + (sum(ord(c) for c in "AKT1") % 1000003))
    jit = [np.clip(C1[0] + 0.1 * 20.0 * rng.normal(0.0, 1.0, 5), lb, ub) for _ in range(8)]
    np.testing.assert_array_equal(C1[1:9], np.stack(jit))
    strat = C1[9:]
    for j in range(5):      # stratified: exactly one sample per bin of width 20 / 15 in every coordinate
        assert sorted((strat[:, j] / (20.0 / 15)).astype(int)) == list(range(15))
    assert not np.array_equal(C1, multistart_candidates("EGFR", base, lb, ub, n_starts=24, seed=42))
    with pytest.raises(ValueError):
        multistart_candidates("x", base, lb, np.full(5, np.inf))


def test_vectorised_draws_make_valid_trajectories():
    """draw() + build(): every trajectory changes each coordinate exactly once by +-delta, stays inside the cube, starts on the grid."""
    from phoskintime_amd.sensitivity import morris
    for p_levels, D, N in ((4, 6, 50), (2, 3, 10), (3, 4, 10), (400, 12, 20), (7, 5, 10)):
        d = morris.draw(D, N, p_levels, seed=5)
        assert d.base.shape == (N, D) and d.rank.dtype == np.int32
        assert all(sorted(r) == list(range(D)) for r in d.rank)
        U = morris.build(d, [[0.0, 1.0]] * D).reshape(N, D + 1, D)
        assert U.min() >= 0.0 and U.max() <= 1.0
        grid = np.arange(p_levels) / (p_levels - 1.0)
        assert np.all(np.min(np.abs(U[:, 0, :, None] - grid[None, None, :]), axis=2) < 1e-12)
        dU = np.diff(U, axis=1)
        assert np.all((np.abs(dU) > 1e-12).sum(axis=2) == 1)                      # one coordinate per step
        assert np.all((np.abs(dU) > 1e-12).sum(axis=1) == 1)                      # each coordinate exactly once
        np.testing.assert_allclose(np.abs(dU).max(axis=2), d.delta, rtol=1e-12)
    # both directions and all admissible levels occur
    d = morris.draw(3, 400, 4, seed=0)
    assert set(np.unique(d.sign)) == {-1.0, 1.0} and len(np.unique(d.base)) == 4


def test_on_disk_artefacts_have_the_reference_layouts(tmp_path):
    """pareto_X/F, sensitivity_indices.csv, fitted_params_picked.json, picked_objectives.json (runner.py:734-743, 912-929;
    global_model/sensitivity.py:266-281) -- file names, columns, ordering."""
    import json
    import pandas as pd
    from phoskintime_amd.global_model import export
    rng = np.random.default_rng(0)
    X = rng.normal(size=(5, 7)); F = rng.uniform(size=(5, 3))
    out = str(tmp_path / "run")
    export.save_pareto(out, X, F)
    np.testing.assert_array_equal(np.load(out + "/pareto_X.npy"), X)
    np.testing.assert_array_equal(np.load(out + "/pareto_F.npy"), F)
    df = pd.read_csv(out + "/pareto_F.csv")
    assert list(df.columns) == ["prot_mse", "rna_mse", "phospho_mse"] and np.allclose(df.values, F)
    problem = {"names": ["a", "b", "c"]}
    Si = {"mu_star": np.array([0.1, 3.0, 1.0]), "sigma": np.array([1.0, 2.0, 3.0]), "mu_star_conf": np.array([0.01, 0.02, 0.03])}
    path = export.save_sensitivity_indices(out, problem, Si)
    ds = pd.read_csv(path)
    assert list(ds.columns) == ["Parameter", "mu_star", "sigma", "mu_star_conf"] and list(ds["Parameter"]) == ["b", "c", "a"]
    params = {"c_k": np.array([1.0, 2.0]), "tf_scale": 0.3}
    dfp = pd.DataFrame({"protein": ["P1"], "time": [0.0], "pred_fc": [1.0]})
    picked = export.save_picked(out, params, F, 2, lambdas=(1.0, 0.5, 2.0), df_prot=dfp)
    assert json.load(open(out + "/fitted_params_picked.json")) == {"c_k": [1.0, 2.0], "tf_scale": 0.3}
    po = json.load(open(out + "/picked_objectives.json"))
    assert po == picked and abs(po["scalar_score"] - (F[2, 0] + 0.5 * F[2, 1] + 2.0 * F[2, 2])) < 1e-15
    assert list(pd.read_csv(out + "/pred_prot_picked.csv").columns) == ["protein", "time", "pred_fc"]
