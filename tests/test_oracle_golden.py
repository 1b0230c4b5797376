"""CPU: the oracle (numpy restatement + SciPy odeint) against the golden vectors produced by running the reference.

The reference's RHS, SciPy-default trajectories and `flat` vectors must be reproduced bit-for-bit (same arithmetic,
same LSODA); the tight trajectories and the closed-form LTI solution pin the oracle's accuracy claims."""
import numpy as np
import pytest

from oracle import protein_models as pm


def _load(f):
    g = np.load(f)
    return g, pm.MODEL_IDS[str(g["model"])], int(g["n_sites"])


def test_golden_inventory(golden_files):
    names = {f.name for f in golden_files}
    for model, ns in (("distmod", (1, 2, 4, 8, 30)), ("succmod", (1, 2, 4, 8, 14)), ("randmod", (1, 2, 3, 4, 5, 6))):
        for n in ns:
            assert f"protein_{model}_n{n}_bounds.npz" in names
            assert f"protein_{model}_n{n}_real.npz" in names
        assert f"protein_{model}_n4_edge.npz" in names


def test_rhs_bit_exact(golden_files):
    for f in golden_files:
        g, model, n = _load(f)
        for k in range(g["theta"].shape[0]):
            np.testing.assert_array_equal(pm.rhs(model, g["y_rand"][k], 0.0, g["theta"][k], n), g["rhs_y_rand"][k])
            np.testing.assert_array_equal(pm.rhs(model, g["y0"][k], 0.0, g["theta"][k], n), g["rhs_y0"][k])


def test_jacobian_and_forcing(golden_files):
    for f in golden_files:
        g, model, n = _load(f)
        for k in range(min(8, g["theta"].shape[0])):
            J = pm.jacobian_analytic(model, g["theta"][k], n)
            scale = max(1.0, np.abs(g["jac"][k]).max())
            assert np.abs(J - g["jac"][k]).max() <= 4e-15 * scale      # reference columns are f(e_j) - f(0): one rounding
            M, b = pm.lti_matrix(model, g["theta"][k], n)
            np.testing.assert_array_equal(b, g["forcing"][k])
            np.testing.assert_array_equal(M, g["jac"][k])


@pytest.mark.parametrize("tag", ["bounds", "real", "edge"])
def test_solve_ode_default_bit_exact(golden_files, tag):
    """Verbatim reference call (SciPy defaults) == oracle, bit for bit, incl. clip and the flat layout."""
    for f in golden_files:
        if not f.name.endswith(f"_{tag}.npz"):
            continue
        g, model, n = _load(f)
        if pm.n_states(model, n) > 20:          # pure-Python RHS at S > 20 is slow; one case is enough
            ks = range(1)
        else:
            ks = range(min(4, g["theta"].shape[0]))
        for k in ks:
            sol, flat = pm.solve_ode(model, g["theta"][k], g["y0"][k], n, g["t"])
            np.testing.assert_array_equal(sol, g["sol_default"][k])
            np.testing.assert_array_equal(flat, g["flat_default"][k])
            assert sol.min() >= 0.0


def test_exact_lti_matches_tight(golden_files):
    """Integrator-free truth (expm) vs the reference RHS under SciPy odeint at 1e-13: pins `sol_tight` itself."""
    for f in golden_files:
        g, model, n = _load(f)
        for k in range(min(3, g["theta"].shape[0])):
            ex = pm.solve_exact_lti(model, g["theta"][k], g["y0"][k], n, g["t"])
            assert pm.band_error(ex, g["sol_tight"][k]) < 2e-3


def test_reference_default_tolerance_is_outside_band_at_32_states(golden_files):
    """Documented fact: the reference's own SciPy-default output misses the 1e-6 / 1e-8 band at S = 32."""
    f = [x for x in golden_files if x.name == "protein_distmod_n30_c3bounds.npz"][0]
    g, model, n = _load(f)
    worst = max(pm.band_error(g["sol_default"][k], np.clip(g["sol_tight"][k], 0, None)) for k in range(g["theta"].shape[0]))
    assert worst > 1.0


def test_score_fit_matches_reference(golden_files):
    for f in golden_files:
        g, model, n = _load(f)
        got = pm.score_fit(g["theta"][0], g["score_target0"], g["flat_default"][0])
        assert got == pytest.approx(float(g["score_fit"][0]), rel=1e-14, abs=0)


def test_flat_layout():
    T, n = 14, 3
    sol = np.arange(T * 5, dtype=float).reshape(T, 5)
    flat = pm.flatten_observables(pm.DIST, sol, n)
    assert flat.shape == ((T - 5) + T + n * T,)
    np.testing.assert_array_equal(flat[:T - 5], sol[5:, 0])
    np.testing.assert_array_equal(flat[T - 5:2 * T - 5], sol[:, 1])
    np.testing.assert_array_equal(flat[2 * T - 5:3 * T - 5], sol[:, 2])
    # randmod keeps only the first n phospho columns
    solr = np.arange(T * 9, dtype=float).reshape(T, 9)
    assert pm.flatten_observables(pm.RAND, solr, 3).shape == ((T - 5) + T + 3 * T,)


def test_compute_Y_metrics_hand_values():
    sol = np.array([[1.0, 2.0, 3.0, 9.0], [2.0, 2.0, 5.0, 9.0]])     # n_sites = 1 -> column 3 is ignored
    assert pm.compute_Y(sol, 1, "total_signal") == 15.0
    assert pm.compute_Y(sol, 1, "mean_activity") == 15.0 / 6
    assert pm.compute_Y(sol, 1, "variance") == pytest.approx(np.var([1, 2, 3, 2, 2, 5]))
    assert pm.compute_Y(sol, 1, "dynamics") == 1.0 + 0.0 + 4.0
    assert pm.compute_Y(sol, 1, "l2_norm") == pytest.approx(np.sqrt(1 + 4 + 9 + 4 + 4 + 25))
    with pytest.raises(ValueError):
        pm.compute_Y(sol, 1, "nope")


def test_compute_bound():
    assert pm.compute_bound(0.0) == [0.0, 0.1]
    assert pm.compute_bound(2.0, 0.5) == [1.0, 3.0]
    assert pm.compute_bound(-2.0, 0.5) == [0.0, -3.0]          # reference quirk: lb clipped, ub not (analysis.py:33-35)


def test_randmod_rate_quirk():
    """randmod.py:201: forward rate into a target mask uses S[lowest set bit of the TARGET]."""
    n = 2
    theta = np.array([0, 0, 0, 0, 3.0, 7.0, 0, 0, 0])          # S = [3, 7], no degradation
    y = np.array([0.0, 0.0, 1.0, 0.0, 0.0])                     # all mass in mask 0b01
    dy = pm.rhs(pm.RAND, y, 0.0, theta, n)
    # 0b01 -> 0b11 carries S[lsb(0b11)] = S[0] = 3 (not S[1] = 7), and 0b01 -> P at unit rate
    assert dy[4] == 3.0 and dy[2] == -(3.0 + 1.0) and dy[1] == 1.0


def test_steady_state_restatement_matches_reference_slsqp():
    """oracle.initial_condition (a linear solve) against the lists the reference's SLSQP formulation returned (tests/golden/steady_init.npz,
    tools/make_golden_steady.py).  SLSQP stops at its own tolerance: 1e-6 relative is what its answers support."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "steady_init.npz")
    assert len(g.files) == 16
    for key in g.files:
        kind, n = key.split("_n")
        y = pm.initial_condition(kind, int(n))
        assert y.shape == g[key].shape
        np.testing.assert_allclose(y, g[key], rtol=2e-6, atol=1e-9, err_msg=key)


def test_wide_fixtures_pin_the_oracle_too(golden_wide_files):
    """Systems beyond 64 states (distmod / succmod n = 64, 100; randmod n = 7, 8): the oracle's RHS against the reference's, bit for bit,
    the analytic Jacobian against the reference's RHS-probed columns, and one verbatim default-tolerance trajectory per model."""
    seen = set()
    for f in golden_wide_files:
        g, model, n = _load(f)
        for k in range(2):
            np.testing.assert_array_equal(pm.rhs(model, g["y_rand"][k], 0.0, g["theta"][k], n), g["rhs_y_rand"][k])
        J = pm.jacobian_analytic(model, g["theta"][0], n)
        assert np.abs(J - g["jac"][0]).max() <= 4e-15 * max(1.0, np.abs(g["jac"][0]).max())
        if model not in seen and pm.n_states(model, n) <= 130:          # LSODA on a pure-Python RHS: one case per model
            seen.add(model)
            sol, flat = pm.solve_ode(model, g["theta"][0], g["y0"][0], n, g["t"])
            np.testing.assert_array_equal(sol, g["sol_default"][0])
            np.testing.assert_array_equal(flat, g["flat_default"][0])
