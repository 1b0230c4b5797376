"""CPU: the oracle's plain-C restatement (oracle/lrp8_dist.c) -- reference RHS bit-exact, and the LRP8 algorithm the HIP throughput
kernel runs, checked against the closed-form LTI solution and the reference's tight SciPy trajectories on the CPU-only suite."""
import numpy as np
import pytest

from oracle import lrp8_cpu, protein_models as pm


def test_c_rhs_is_the_reference_rhs(golden_files):
    for f in golden_files:
        g = np.load(f)
        if str(g["model"]) != "distmod":
            continue
        n = int(g["n_sites"])
        for k in range(g["theta"].shape[0]):
            np.testing.assert_array_equal(lrp8_cpu.rhs(g["y_rand"][k], g["theta"][k], n), g["rhs_y_rand"][k])


def test_c_lrp8_within_band_of_reference_tight(golden_files):
    worst = 0.0
    for f in golden_files:
        g = np.load(f)
        if str(g["model"]) != "distmod":
            continue
        n = int(g["n_sites"])
        K = min(g["theta"].shape[0], 16)
        if not np.all(g["y0"][:K] == g["y0"][0]):
            K = 1
        for kw, lo, hi in (({}, 10, 60), ({"stages": 8, "rtol": 1e-7, "atol": 1e-9}, 20, 120)):      # LRP12 at the defaults, LRP8 at 1e-7
            sol, st, ns = lrp8_cpu.solve_batch(g["theta"][:K], n, g["y0"][0], g["t"], **kw)
            assert not st.any()
            worst = max(worst, pm.band_error(sol, g["sol_tight"][:K]))
            assert lo <= ns[:, 0].mean() <= hi
    assert worst <= 0.1


def test_c_lrp8_flags_instead_of_crashing():
    th = np.random.default_rng(0).uniform(0.1, 3, (3, 12))
    th[1, 5] = np.nan
    sol, st, ns = lrp8_cpu.solve_batch(th, 4, np.ones(6), pm.TIME_POINTS)
    assert st[0] == 0 and st[2] == 0 and st[1] != 0 and np.isnan(sol[1, -1]).all()
    sol, st, ns = lrp8_cpu.solve_batch(th[[0]], 4, np.ones(6), pm.TIME_POINTS, max_steps=5)
    assert st[0] & 2
