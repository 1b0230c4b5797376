"""CPU: the network oracle (oracle/network_models.py) against golden vectors produced by running the reference's
global_model classes (tools/make_golden_network.py): RHS and finite-difference Jacobian bit for bit, LSODA trajectories."""
import numpy as np
import pytest

from oracle import network_models as nm
from pathlib import Path

GOLD = sorted((Path(__file__).resolve().parent / "golden").glob("network_m*.npz"))


def test_inventory():
    names = {f.name for f in GOLD}
    for m in (0, 1, 2, 4):
        assert f"network_m{m}_small.npz" in names and f"network_m{m}_medium.npz" in names


@pytest.mark.parametrize("f", GOLD, ids=lambda f: f.stem)
def test_rhs_and_fd_jacobian_bit_exact(f):
    g = np.load(f); net = nm.Network.from_npz(g)
    np.testing.assert_array_equal(nm.default_y0(net), g["y0"])
    ks = range(4) if net.S < 60 else range(2)
    for k in ks:
        p = nm.Params.from_npz(g, k)
        for ti, t in enumerate(g["t_probe"]):
            np.testing.assert_array_equal(nm.rhs(net, p, g["y0"], t), g["rhs_y0"][k, ti])
            np.testing.assert_array_equal(nm.rhs(net, p, g["y_rand"][k], t), g["rhs_rand"][k, ti])
    if net.S < 60:
        p = nm.Params.from_npz(g, 0)
        np.testing.assert_array_equal(nm.fd_jacobian(net, p, g["y_rand"][0], float(g["fd_jac_t"])), g["fd_jac"][0])


@pytest.mark.parametrize("f", [x for x in GOLD if "small" in x.name], ids=lambda f: f.stem)
def test_simulate_odeint_reproduces_reference(f):
    """Same SciPy LSODA, same RHS, same finite-difference Dfun => the same trajectory (bit for bit)."""
    g = np.load(f); net = nm.Network.from_npz(g)
    p = nm.Params.from_npz(g, 1)
    Y = nm.simulate_odeint(net, p, g["t_eval"], 1e-8, 1e-8, 200000)
    np.testing.assert_array_equal(Y, g["Y_lsoda8"][1])


def test_time_bucket_edges():
    grid = np.array([0.0, 0.5, 0.75, 1.0, 2.0])
    assert [nm.time_bucket(t, grid) for t in (-1.0, 0.0, 0.3, 0.5, 0.74, 0.75, 1.9, 2.0, 9.0)] == [0, 0, 0, 1, 1, 2, 3, 4, 4]


def test_synthesis_rate_branches():
    assert nm.calculate_synthesis_rate(2.0, 3.0, 0.0) == 2.0
    u = 0.5 / 1.5
    assert nm.calculate_synthesis_rate(2.0, 3.0, 0.5) == 2.0 * (1.0 + (3.0 * u) / (1.0 + u + 1e-6))
    assert nm.calculate_synthesis_rate(2.0, 3.0, -0.5) == 2.0 / (1.0 + 3.0 * u)


@pytest.mark.parametrize("m", [0, 2])
def test_loss_function_matches_reference_all_modes(m):
    """lossfn.loss_function_noncomb / _comb for every LOSS_MODE (reference outputs: tools/make_golden_loss.py)."""
    g = np.load(Path(__file__).resolve().parent / "golden" / f"network_loss_m{m}.npz")
    ld = {k: g[k] for k in g.files}
    for mode in range(8):
        for k in range(g["Y"].shape[0]):
            got = np.array(nm.loss_function(m, g["Y"][k], ld, mode))
            np.testing.assert_allclose(got, g["loss_sums"][mode, k], rtol=1e-13, atol=0, equal_nan=True)
    assert np.isnan(g["loss_sums"][2]).any()      # LOSS_MODE 2 takes log(diff + eps) of negative residuals in the reference itself


@pytest.mark.parametrize("m", [0, 1, 2, 4])
def test_oracle_rk45_restatement_reproduces_the_reference_integrator(m):
    """oracle.simulate_rk45 (restating solvers.py:293-758) against the reference's own RK45 outputs stored in the fixtures."""
    g = np.load(GOLD[0].parent / f"network_m{m}_small.npz")
    net = nm.Network.from_npz(g)
    for k in range(2):
        Y = nm.simulate_rk45(net, nm.Params.from_npz(g, k), g["t_eval"], 1e-5, 1e-7, y0=g["y0"])
        np.testing.assert_allclose(Y, g["Y_rk45"][k], rtol=1e-13, atol=1e-15)
    Y = nm.simulate_rk45(net, nm.Params.from_npz(g, 0), g["t_eval"], 1e-9, 1e-11, y0=g["y0"])
    np.testing.assert_allclose(Y, g["Y_rk45_tight"][0], rtol=1e-13, atol=1e-15)
    # the explicit method at its default tolerance is ~1e-5 from the truth; at 1e-9 it agrees with LSODA at 1e-12
    assert np.abs(g["Y_rk45_tight"][0] - g["Y_tight"][0]).max() < 1e-7


def test_oracle_frechet_restatement_matches_reference():
    g = np.load(GOLD[0].parent / "frechet.npz")
    for k in range(g["dist"].size):
        assert abs(nm.frechet_distance(g[f"a{k}"], g[f"b{k}"]) - g["dist"][k]) <= 1e-12 * max(1.0, g["dist"][k])


LARGE = sorted((Path(__file__).resolve().parent / "golden").glob("netlarge_m[0-9].npz"))


def test_large_network_inventory():
    """One N = 100 / S ~ 550 network per topology at BASELINE config 4 / 5 size, run through the reference once (VERDICT r1 missing #4)."""
    assert [f.name for f in LARGE] == [f"netlarge_m{m}.npz" for m in (0, 1, 2, 4)]
    for f in LARGE:
        g = np.load(f)
        assert int(g["N"]) == 100 and 500 <= int(g["S"]) <= 600 and g["Y_lsoda8"].shape[0] == 2 and g["Y_tight"].shape[0] == 1


@pytest.mark.parametrize("f", LARGE, ids=lambda f: f.stem)
def test_large_network_rhs_bit_exact(f):
    g = np.load(f); net = nm.Network.from_npz(g)
    np.testing.assert_array_equal(nm.default_y0(net), g["y0"])
    for k in range(2):
        p = nm.Params.from_npz(g, k)
        for ti, t in enumerate(g["t_probe"]):
            np.testing.assert_array_equal(nm.rhs(net, p, g["y_rand"][k], t), g["rhs_rand"][k, ti])
    # the reference's optimiser-tolerance run (1e-8) against its own 1e-12 run: the error the parity band has to live with
    band = np.max(np.abs(g["Y_lsoda8"][0] - g["Y_tight"][0]) / (1e-8 + 1e-6 * np.abs(g["Y_tight"][0])))
    assert band < 5.0


MORE = sorted((Path(__file__).resolve().parent / "golden").glob("netlarge_more_m[0-9].npz"))


def test_large_network_more_reference_runs_inventory():
    """tools/make_golden_network.py large_more: further parameter sets of the SAME four N = 100 networks, each integrated by the reference at
    1e-12 -- the truth the GPU population tests compare with: seven per topology, i.e. eight reference-run 1e-12 trajectories per network
    with the fixture's own."""
    assert [f.name for f in MORE] == [f"netlarge_more_m{m}.npz" for m in (0, 1, 2, 4)]
    for f in MORE:
        q = np.load(f); g = np.load(f.parent / f"netlarge_m{int(q['model'])}.npz")
        K = int(q["done"])
        assert K == 7 and q["Y_tight"].shape == (K, 15, int(g["S"])) and np.isfinite(q["Y_tight"]).all()
        np.testing.assert_array_equal(q["y0"], g["y0"]); np.testing.assert_array_equal(q["t_eval"], g["t_eval"])
        for k in range(K):
            np.testing.assert_array_equal(q["Y_tight"][k, 0], g["y0"])
        # candidate 0 is the fixture's second parameter set (the one that had only a 1e-8 run): same parameters, and its 1e-8 run agrees
        assert int(q["from_netlarge_index"][0]) == 1
        for key in ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i", "tf_scale"):
            np.testing.assert_array_equal(q[key][0], g[key][1])
            assert q[key].shape[0] == K
        band = np.max(np.abs(g["Y_lsoda8"][1] - q["Y_tight"][0]) / (1e-8 + 1e-6 * np.abs(q["Y_tight"][0])))
        assert band < 5.0
        # the oracle's RHS restatement takes these parameter sets (shapes of the same network) and is finite at the start
        net = nm.Network.from_npz(g)
        for k in range(K):
            p = nm.Params.from_npz(q, k)
            f0 = nm.rhs(net, p, q["Y_tight"][k, 0], float(q["t_eval"][0]))
            assert np.isfinite(f0).all()
