"""GPU parity for per-protein systems beyond one wavefront's lane groups (csrc/pk_wide.hpp): distmod / succmod with more than 64 states
and randmod with n_sites >= 7 -- round 1 refused these (VERDICT r1 "missing" #1; randmod is the reference's default model and has no
size limit, models/randmod.py:9-85).  Golden files made by running the reference (tools/make_golden.py randmod 7, distmod 100, ...)."""
import pathlib

import numpy as np
import pytest

from oracle import protein_models as pm

pytestmark = pytest.mark.gpu
RTOL_GATE, ATOL_GATE = 1e-6, 1e-8


@pytest.fixture(scope="module")
def eng():
    from phoskintime_amd import batch
    batch.get_context()
    return batch


def _load(f):
    g = np.load(f)
    return g, pm.MODEL_IDS[str(g["model"])], int(g["n_sites"])


def _np(x):
    return x.detach().cpu().numpy()


def test_wide_inventory(golden_wide_files):
    names = {f.name for f in golden_wide_files}
    for want in ("protein_randmod_n7_bounds.npz", "protein_randmod_n7_real.npz", "protein_distmod_n64_bounds.npz", "protein_distmod_n100_real.npz",
                 "protein_succmod_n64_bounds.npz", "protein_succmod_n100_real.npz"):
        assert want in names


def test_wide_trajectories_within_band_of_reference_scipy(eng, golden_wide_files):
    """Library defaults (what the drop-in `solve_ode` uses) on every wide golden case: inside the parity band of the reference's RHS under
    SciPy odeint at 1e-13, and no further from the verbatim reference call than the reference is from the truth."""
    for f in golden_wide_files:
        g, model, n = _load(f)
        r = eng.solve_ode_batch(model, g["theta"], g["y0"], n, g["t"], clip_nonneg=False)
        sol = _np(r.sol)
        assert not _np(r.status).any(), f.name
        e = pm.band_error(sol, g["sol_tight"], RTOL_GATE, ATOL_GATE)
        assert e <= 0.1, f"{f.name}: band error {e}"      # every fixture size (randmod n = 7, 8 included) integrates with LRP12 on exact solves
        ref_own = pm.band_error(g["sol_default"], np.clip(g["sol_tight"], 0, None))
        e_def = pm.band_error(np.clip(sol, 0, None), g["sol_default"])
        assert e_def <= ref_own + 1.0, f"{f.name}: {e_def} vs reference's own {ref_own}"
        np.testing.assert_array_equal(sol[:, 0, :], g["y0"])


def test_wide_flat_metric_and_layout(eng, golden_wide_files):
    for f in golden_wide_files:
        g, model, n = _load(f)
        for metric in ("total_signal", "variance", "dynamics", "l2_norm", "mean_activity"):
            r = eng.solve_ode_batch(model, g["theta"][:3], g["y0"][:3], n, g["t"], metric=metric)
            sol, flat, m = _np(r.sol), _np(r.flat), _np(r.metric)
            assert sol.min() >= 0.0
            for k in range(sol.shape[0]):
                np.testing.assert_array_equal(flat[k], pm.flatten_observables(model, sol[k], n))
                assert m[k] == pytest.approx(pm.compute_Y(sol[k], n, metric), rel=1e-10, abs=1e-12), (f.name, metric)
        assert flat.shape[1] == g["flat_default"].shape[1]


def test_wide_rhs_and_jacobian_match_reference(eng, golden_wide_files):
    for f in golden_wide_files:
        g, model, n = _load(f)
        dy = _np(eng.rhs_batch(model, g["theta"], g["y_rand"], n))
        scale = np.abs(g["jac"]).max(axis=(1, 2))[:, None] * np.abs(g["y_rand"]).max() * g["y_rand"].shape[1]
        assert (np.abs(dy - g["rhs_y_rand"]) <= 4e-16 * np.maximum(scale, 1.0) * 4).all(), f.name
        J = _np(eng.jacobian_batch(model, g["theta"], n))
        assert (np.abs(J - g["jac"]) <= 4e-15 * np.maximum(np.abs(g["jac"]).max(), 1.0)).all(), f.name


@pytest.mark.parametrize("model,n", [(pm.DIST, 63), (pm.DIST, 300), (pm.DIST, 1276), (pm.SUCC, 63), (pm.SUCC, 200), (pm.SUCC, 1000), (pm.RAND, 7), (pm.RAND, 8), (pm.RAND, 9),
                                     (pm.RAND, 10)])
def test_wide_sizes_against_closed_form(eng, model, n):
    """Sizes without a reference fixture (both ends of each range) against the oracle's independent closed form (matrix exponential of
    the affine system), ragged batch, batched y0, normalisation, shared-y0 == per-replica-y0 bits."""
    rng = np.random.default_rng(n)
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    B = 5
    th = rng.uniform(0.0, 20.0, (B, P)); th[1] = rng.uniform(0.05, 2.0, P)
    y0 = rng.uniform(0.2, 2.0, (B, S))
    t = np.array([0.0, 0.5, 4.0, 60.0, 960.0])
    r = eng.solve_ode_batch(model, th, y0, n, t, clip_nonneg=False)
    sol = _np(r.sol)
    assert not _np(r.status).any() and sol.shape == (B, t.size, S)
    for b in (0, 1) if S > 400 else range(B):
        ref = pm.solve_exact_lti(model, th[b], y0[b], n, t)
        e = pm.band_error(sol[b], ref)
        assert e <= (0.5 if (model == pm.RAND and n >= 9) else 0.1), (model, n, b, e)      # n >= 9: the n-cube kernel (approximate factorisation)
    rn = eng.solve_ode_batch(model, th, y0, n, t, clip_nonneg=False, normalize=True)
    np.testing.assert_allclose(_np(rn.sol), sol * (1.0 / y0)[:, None, :], rtol=1e-15)
    one = eng.solve_ode_batch(model, th[3:4], y0[3], n, t, clip_nonneg=False)
    np.testing.assert_array_equal(_np(one.sol)[0], sol[3])


def test_wide_failures_are_flagged_not_fatal_and_unsupported_options_raise(eng):
    from phoskintime_amd._capi import ST_MAXSTEPS, ST_NONFINITE, PhoskinError
    rng = np.random.default_rng(1)
    for model, n in ((pm.DIST, 80), (pm.SUCC, 80), (pm.RAND, 7), (pm.RAND, 8), (pm.RAND, 9)):      # n = 7, 8: parity-elimination kernel; n = 9: n-cube kernel
        P, S = pm.n_params(model, n), pm.n_states(model, n)
        th = rng.uniform(0.1, 3.0, (4, P))
        good = _np(eng.solve_ode_batch(model, th, np.ones(S), n, pm.TIME_POINTS).sol)
        bad = th.copy(); bad[2, 5] = np.nan
        r = eng.solve_ode_batch(model, bad, np.ones(S), n, pm.TIME_POINTS, metric="total_signal")
        st, sol = _np(r.status), _np(r.sol)
        assert st[2] & ST_NONFINITE and np.isnan(sol[2, 1:]).all() and np.isnan(_np(r.metric)[2])
        for k in (0, 1, 3):
            assert st[k] == 0
            np.testing.assert_array_equal(sol[k], good[k])
        r = eng.solve_ode_batch(model, th, np.ones(S), n, pm.TIME_POINTS, max_steps=5)
        assert (_np(r.status) & ST_MAXSTEPS).all() and np.isnan(_np(r.sol)[:, -1, :]).all() and np.isfinite(_np(r.sol)[:, 0, :]).all()
        with pytest.raises(PhoskinError):
            eng.solve_ode_batch(model, th, np.ones(S), n, pm.TIME_POINTS, method="bdf2")
    with pytest.raises(PhoskinError):
        eng.solve_ode_batch(pm.DIST, np.ones((1, 4 + 2 * 1277)), np.ones(1279), 1277, pm.TIME_POINTS)        # beyond the LDS budget: refused, not wrong
    with pytest.raises(PhoskinError):
        eng.solve_ode_batch(pm.DIST, np.ones((1, 4 + 2 * 100)), np.ones(102), 100, pm.TIME_POINTS, method="lrp8")


def test_wide_randmod_hbm_scratch_path(eng):
    """n_sites = 12 (4097 states, 4111 parameters per replica): the nine work vectors no longer fit LDS and live in the context's HBM
    scratch arena.  Checked against scipy's sparse expm action on the oracle's LTI matrix over a short horizon."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import expm_multiply
    from phoskintime_amd import batch
    n = 12
    P, S = pm.n_params(pm.RAND, n), pm.n_states(pm.RAND, n)
    rng = np.random.default_rng(12)
    th = rng.uniform(0.05, 3.0, (2, P))
    y0 = rng.uniform(0.2, 2.0, S)
    t = np.array([0.0, 0.25, 1.0])
    r = eng.solve_ode_batch(pm.RAND, th, y0, n, t, clip_nonneg=False)
    assert not _np(r.status).any()
    st = batch.get_context().workspace_stats()
    assert st["scratch_bytes"] >= 2 * 9 * S * 8
    sol = _np(r.sol)
    for b in range(2):
        M = pm.jacobian_analytic(pm.RAND, th[b], n)                       # O(S n) loops (lti_matrix probes the RHS column by column: minutes at S = 4097)
        bvec = pm.rhs(pm.RAND, np.zeros(S), 0.0, th[b], n)
        Aug = sp.bmat([[sp.csr_matrix(M), sp.csr_matrix(bvec[:, None])], [None, sp.csr_matrix((1, 1))]], format="csr")
        z = np.concatenate([y0, [1.0]])
        for k in range(1, t.size):
            z = expm_multiply(Aug * (t[k] - t[k - 1]), z)
            assert pm.band_error(sol[b, k], z[:S]) <= 0.5, (b, k)


def test_wide_dropin_solve_ode_surface(eng, golden_wide_files):
    """phoskintime_amd.models.solve_ode(params, init_cond, num_psites, t) -> (sol, flat) for a 7-site randmod protein: the call round 1
    answered with PK_ERR_UNSUPPORTED."""
    from phoskintime_amd import models
    f = [x for x in golden_wide_files if x.name == "protein_randmod_n7_real.npz"][0]
    g, model, n = _load(f)
    models.set_model("randmod")
    sol, flat = models.solve_ode(g["theta"][0], g["y0"][0], n, g["t"])
    assert sol.shape == g["sol_default"][0].shape and flat.shape == g["flat_default"][0].shape
    assert pm.band_error(sol, np.clip(g["sol_tight"][0], 0, None)) <= 0.5


@pytest.mark.parametrize("model,n", [(pm.DIST, 100), (pm.DIST, 1276), (pm.SUCC, 63), (pm.SUCC, 500), (pm.RAND, 6), (pm.RAND, 7), (pm.RAND, 9)])
def test_wide_steady_states(eng, model, n):
    """pk_steady_state_protein_batch beyond 64 states (round 1: PK_ERR_UNSUPPORTED): against the oracle's dense linear solve of J y* = -b;
    a parameter set without degradation (singular J) is flagged and NaN-filled, its neighbours untouched."""
    from phoskintime_amd._capi import ST_NONFINITE
    rng = np.random.default_rng(100 + n)
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    th = rng.uniform(0.2, 5.0, (4, P))
    bad = th.copy(); bad[2, 1] = 0.0                                 # B = 0: the mRNA never degrades -> no steady state
    yss, st = eng.steady_state_batch(model, bad, n)
    yss, st = _np(yss), _np(st)
    assert st[2] != 0 and np.isnan(yss[2]).all() and not st[[0, 1, 3]].any()
    for b in (0, 1, 3):
        want = pm.steady_state(model, th[b], n)
        np.testing.assert_allclose(yss[b], want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
        # and it IS a steady state of the reference's right-hand side
        assert np.abs(pm.rhs(model, yss[b], 0.0, th[b], n)).max() <= 1e-9 * (1.0 + np.abs(want).max() * np.abs(th[b]).max())


@pytest.mark.parametrize("n", [7, 8])
def test_randmod_exact_kernels_have_no_stragglers(eng, n):
    """n = 7, 8 integrate with LRP12 on EXACT solves (csrc/pk_rand_parity.hpp: odd-popcount states eliminated, even Schur complement inverted
    in registers): the default method's step counts for EVERY draw from the reference's bounds, including mRNA degradation ~ 0 (a solution
    that never comes to rest), which costs the approximate-factorisation kernel 40x the steps of its neighbours."""
    model = pm.RAND
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    rng = np.random.default_rng(20260515)
    th = rng.uniform(0.0, 20.0, (64, P))
    th[:4, 1] = (0.0002, 0.008, 0.016, 0.05)                       # the stragglers of tools/gpu_wide_outlier.py
    t = pm.TIME_POINTS
    r = eng.solve_ode_batch(model, th, np.ones(S), n, t, clip_nonneg=False)
    ns, sol = _np(r.n_steps), _np(r.sol)
    assert not _np(r.status).any()
    assert ns[:, 0].max() <= 80, ns[:, 0].max()
    for b in range(8):
        assert pm.band_error(sol[b], pm.solve_exact_lti(model, th[b], np.ones(S), n, t)) <= 0.1, b


def test_randmod_n7_approximate_factorisation_path_still_in_band(golden_wide_files):
    """PK_WIDE_RAND_DENSE=0 (read once per process: hence a child process) sends n = 7 through the n-cube kernel that n >= 8 uses."""
    import os, subprocess, sys, textwrap
    f = [x for x in golden_wide_files if x.name == "protein_randmod_n7_real.npz"][0]
    code = textwrap.dedent(f"""
        import numpy as np, sys
        sys.path.insert(0, {str(pathlib.Path(__file__).resolve().parents[1])!r})
        from phoskintime_amd import batch
        from oracle import protein_models as pm
        g = np.load({str(f)!r})
        r = batch.solve_ode_batch("randmod", g["theta"][:4], g["y0"][:4], 7, g["t"], clip_nonneg=False)
        ns = r.n_steps.cpu().numpy()
        e = pm.band_error(r.sol.cpu().numpy(), g["sol_tight"][:4])
        assert not r.status.cpu().numpy().any() and e <= 0.6 and ns[:, 0].min() > 150, (e, ns)
        print("ok", e, ns[:, 0])
    """)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "PK_WIDE_RAND_DENSE": "0"}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_ncube_kernel_drift_removal_bounds_the_steps_of_draws_that_never_rest(eng):
    """randmod n = 9 (the n-cube kernel; n = 8 has exact solves since round 3): mRNA degradation B ~ 0 -- down to exactly 0, where the mRNA grows linearly for ever -- used to cost
    10-40x the steps of a benign draw.  With the closed-form response to the mRNA row subtracted the remainder comes to rest: step counts
    stay in the benign range and the trajectories stay inside the band of the closed form."""
    n, model = 9, pm.RAND
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    rng = np.random.default_rng(20260515)
    th = rng.uniform(0.0, 20.0, (12, P))
    th[:5, 1] = (0.0, 1e-7, 0.0002, 0.008, 0.05)
    t = pm.TIME_POINTS
    r = eng.solve_ode_batch(model, th, np.ones(S), n, t, clip_nonneg=False)
    ns, sol = _np(r.n_steps), _np(r.sol)
    assert not _np(r.status).any()
    assert ns[:, 0].max() <= 4000, ns[:, 0]
    for b in range(6):
        assert pm.band_error(sol[b], pm.solve_exact_lti(model, th[b], np.ones(S), n, t)) <= 0.6, b


def test_ncube_kernel_without_drift_removal_still_in_band(golden_wide_files):
    """PK_WIDE_RAND_DRIFT=0 (read once per process: a child process): the plain path on the reference fixture and on one slow-mRNA draw,
    which then needs several times the steps."""
    import os, subprocess, sys, textwrap
    f = [x for x in golden_wide_files if x.name == "protein_randmod_n8_real.npz"]
    f = f[0] if f else [x for x in golden_wide_files if "randmod_n8" in x.name][0]
    code = textwrap.dedent(f"""
        import numpy as np, sys
        sys.path.insert(0, {str(pathlib.Path(__file__).resolve().parents[1])!r})
        from phoskintime_amd import batch
        from oracle import protein_models as pm
        g = np.load({str(f)!r})
        r = batch.solve_ode_batch("randmod", g["theta"][:3], g["y0"][:3], 8, g["t"], clip_nonneg=False)
        e = pm.band_error(r.sol.cpu().numpy(), g["sol_tight"][:3])
        assert not r.status.cpu().numpy().any() and e <= 0.6 and r.n_steps.cpu().numpy()[:, 0].min() > 150, e      # > 150 steps: the n-cube kernel ran
        th = np.random.default_rng(20260515).uniform(0.0, 20.0, (1, pm.n_params(2, 8))); th[0, 1] = 0.008
        q = batch.solve_ode_batch("randmod", th, np.ones(257), 8, pm.TIME_POINTS, clip_nonneg=False)
        print("ok", e, int(q.n_steps.cpu().numpy()[0, 0]))
    """)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "PK_WIDE_RAND_DRIFT": "0", "PK_WIDE_RAND_EXACT": "0"}, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    assert int(out.stdout.split()[-1]) > 2500            # the same draw takes < 2 500 steps with the drift removed (test above)


def test_ncube_kernel_drift_removal_declines_safely(eng):
    """Draws where the two linear solves of the drift removal cannot be trusted -- mRNA degradation faster than the cube's slowest rate makes
    M_c + B I indefinite (the defect correction stalls), or B is not small against the smallest loss rate -- are integrated as they stand:
    still inside the band, never flagged."""
    n, model = 9, pm.RAND
    P, S = pm.n_params(model, n), pm.n_states(model, n)
    rng = np.random.default_rng(4)
    th = rng.uniform(0.5, 3.0, (3, P))
    th[0, 3] = 1e-4; th[0, 4 + n:] = 1e-4; th[0, 1] = 0.02          # protein degradation ~ 1e-4 everywhere, mRNA degradation 0.02: indefinite
    th[1, 1] = 15.0                                                  # B far above 0.25 x the smallest loss: not attempted
    th[2, 1] = 0.0; th[2, 0] = 0.0                                   # no mRNA synthesis and no degradation: R constant, u = 0
    t = pm.TIME_POINTS
    r = eng.solve_ode_batch(model, th, np.ones(S), n, t, clip_nonneg=False)
    assert not _np(r.status).any()
    sol = _np(r.sol)
    for b in range(3):
        assert pm.band_error(sol[b], pm.solve_exact_lti(model, th[b], np.ones(S), n, t)) <= 0.6, b


_EXACT_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from phoskintime_amd import batch
from oracle import protein_models as pm
out = {}
for n in (6, 7, 8):
    rng = np.random.default_rng(40 + n)
    P, S = pm.n_params(2, n), pm.n_states(2, n)
    th = np.concatenate([rng.uniform(0.0, 20.0, (5, P)), np.exp(rng.uniform(np.log(1e-3), np.log(1e2), (5, P)))])
    th[0, 1] = 0.0; th[1, 1] = 0.008                       # mRNA that never comes to rest
    y0 = rng.uniform(0.3, 1.5, (10, S))
    r = batch.solve_ode_batch("randmod", th, y0, n, pm.TIME_POINTS, clip_nonneg=False, metric="variance")
    out[f"sol{n}"] = r.sol.cpu().numpy(); out[f"st{n}"] = r.status.cpu().numpy(); out[f"ns{n}"] = r.n_steps.cpu().numpy(); out[f"m{n}"] = r.metric.cpu().numpy()
    out[f"th{n}"] = th; out[f"y0{n}"] = y0
np.savez(sys.argv[2], **out)
"""


def test_exact_randmod_kernels_agree_with_each_other_and_the_closed_form(tmp_path):
    """Three independent exact implementations per size, selected per process (PK_WIDE_RAND_EXACT / PK_RAND_LEVEL6 are read once):
      n = 8: parity elimination (default) vs block elimination over the popcount levels (csrc/pk_rand_level.hpp)
      n = 7: parity elimination (default) vs the full 128 x 128 inverse in registers (csrc/pk_rand_dense.hpp, round 2)
      n = 6: parity elimination in one wave per replica ([r3], default) vs the popcount-level kernel vs the 64 x 64 in-register inverse of
             round 1 (PK_RAND_PARITY56=0)
    Same LRP12 steps on the same matrices, different elimination orders: equal step counts, trajectories far inside the band of each other
    and of the oracle's closed form -- on draws from the reference's bounds, log-uniform draws over five decades and two stragglers."""
    import os, subprocess, sys
    root = str(pathlib.Path(__file__).resolve().parents[1])
    res = {}
    for tag, env in (("default", {}), ("twin", {"PK_WIDE_RAND_EXACT": "2", "PK_RAND_LEVEL6": "1"}), ("third", {"PK_RAND_PARITY56": "0"})):
        f = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", _EXACT_SCRIPT, root, str(f)], check=True, env={**os.environ, **env}, timeout=900)
        res[tag] = np.load(f)
    a, b, c3 = res["default"], res["twin"], res["third"]
    assert not c3["st6"].any() and np.abs(a["ns6"][:, 0] - c3["ns6"][:, 0]).max() <= 2
    assert pm.band_error(a["sol6"], c3["sol6"]) <= 0.05
    np.testing.assert_allclose(a["m6"], c3["m6"], rtol=1e-6)
    for n in (6, 7, 8):
        assert not a[f"st{n}"].any() and not b[f"st{n}"].any()
        assert np.abs(a[f"ns{n}"][:, 0] - b[f"ns{n}"][:, 0]).max() <= 2 and a[f"ns{n}"][:, 0].max() <= 80, (n, a[f"ns{n}"][:, 0], b[f"ns{n}"][:, 0])
        assert pm.band_error(a[f"sol{n}"], b[f"sol{n}"]) <= 0.05, n
        np.testing.assert_allclose(a[f"m{n}"], b[f"m{n}"], rtol=1e-6)
        for k in (0, 1, 4, 5, 9):
            exact = pm.solve_exact_lti(2, a[f"th{n}"][k], a[f"y0{n}"][k], n, pm.TIME_POINTS)
            assert pm.band_error(a[f"sol{n}"][k], exact) <= 0.2, (n, k)      # log-uniform draws: the oracle's matrix exponentials carry part of this
            assert pm.band_error(b[f"sol{n}"][k], exact) <= 0.2, (n, k)


def test_randmod_n8_population_step_counts(eng):
    """1 024 draws from U(0, 20) and 1 024 log-uniform draws at n = 8 -- at most 80 steps for every one of them
    (the n-cube kernel took up to 30 000 and left the parity band on the log-uniform set: profiles/r03_c_rand8_ab.txt)."""
    n = 8
    P, S = pm.n_params(2, n), pm.n_states(2, n)
    rng = np.random.default_rng(8)
    for th in (rng.uniform(0.0, 20.0, (1024, P)), np.exp(rng.uniform(np.log(1e-3), np.log(1e2), (1024, P)))):
        r = eng.solve_ode_batch(2, th, np.ones(S), n, pm.TIME_POINTS, want_flat=False)
        ns = _np(r.n_steps)
        assert not _np(r.status).any() and ns[:, 0].max() <= 80 and ns[:, 1].max() <= 10, (ns[:, 0].max(), ns[:, 1].max())
        assert np.isfinite(_np(r.sol)).all() and _np(r.sol).min() >= 0.0
