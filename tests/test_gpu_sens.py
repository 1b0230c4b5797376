"""GPU: forward parameter sensitivities (csrc/pk_sens.hpp, pk_solve_protein_sens_batch) against central differences of the oracle's
integrator-free solution (oracle.protein_models.solve_exact_lti: matrix exponentials), and the Levenberg-Marquardt driver on them.
Reference behaviour replaced: scipy.optimize.curve_fit's '2-point' differencing of models.solve_ode (paramest/normest.py:167-326)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import protein_models as pm

pytestmark = pytest.mark.gpu

SENS_RTOL = 1e-7          # d flat / d theta against the oracle's central differences: |diff| <= SENS_RTOL * (1 + |d|)


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from phoskintime_amd import batch
    batch.get_context()
    return batch


def _oracle_flat(model, th, y0, n, t):
    sol = pm.solve_exact_lti(model, th, y0, n, t)
    return pm.flatten_observables(model, np.clip(sol, 0.0, None), n)


def _oracle_jac(model, th, y0, n, t):
    P = th.size
    cols = []
    for c in range(P):
        h = 1e-5 * max(1.0, abs(th[c]))
        tp, tm = th.copy(), th.copy()
        tp[c] += h; tm[c] -= h
        cols.append((_oracle_flat(model, tp, y0, n, t) - _oracle_flat(model, tm, y0, n, t)) / (2 * h))
    return np.stack(cols, axis=1)                      # [F, P]


CASES = [("distmod", 1), ("distmod", 3), ("distmod", 4), ("distmod", 8), ("distmod", 13), ("distmod", 14),
         ("succmod", 1), ("succmod", 2), ("succmod", 5), ("succmod", 9), ("succmod", 14),
         # rows-per-lane kernel (csrc/pk_sens_rows.hpp): 32-lane groups up to 30 sites (BASELINE config 3's size), 64-lane groups to 62
         ("distmod", 15), ("distmod", 30), ("distmod", 31), ("distmod", 62), ("succmod", 15), ("succmod", 30), ("succmod", 47), ("succmod", 62),
         ("randmod", 1), ("randmod", 2), ("randmod", 3), ("randmod", 4), ("randmod", 5),
         # parity-eliminated inverse serving eight columns per workgroup (csrc/pk_rand_sens.hpp): 74 / 139 columns of 65 / 129 rows
         ("randmod", 6), ("randmod", 7)]


@pytest.mark.parametrize("model,n", CASES)
def test_sensitivities_match_central_differences_of_the_oracle(eng, model, n):
    mid = pm.MODEL_IDS[model]
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(100 * mid + n)
    B = 5 if S <= 64 else 2                             # not a multiple of the replicas per wave: the last wave carries shadow groups
    th = rng.uniform(0.2, 2.0, size=(B, P))
    y0 = rng.uniform(0.3, 1.5, size=S)
    t = pm.TIME_POINTS
    res = eng.solve_ode_sens_batch(model, th, y0, n, t, rtol=1e-9, atol=1e-11)
    assert int(res.status.cpu().numpy().max()) == 0
    flat = res.flat.cpu().numpy(); dflat = res.dflat.cpu().numpy()
    plain = eng.solve_ode_batch(model, th, y0, n, t, want_sol=False, rtol=1e-9, atol=1e-11).flat.cpu().numpy()
    assert np.max(np.abs(flat - plain) / (1e-8 + 1e-6 * np.abs(plain))) < 0.1
    for b in (0, B - 1):
        ref_f = _oracle_flat(mid, th[b], y0, n, t)
        assert pm.band_error(flat[b], ref_f) < 0.1
        ref_J = _oracle_jac(mid, th[b], y0, n, t)
        err = np.abs(dflat[b] - ref_J) / (1.0 + np.abs(ref_J))
        assert err.max() < SENS_RTOL, (model, n, b, err.max())
        assert np.isfinite(dflat[b]).all()


def test_clipped_entries_have_zero_derivative_rows(eng):
    """flat is np.clip(sol, 0, None) flattened (distmod.py:112-134): where the clip is active the output is the constant 0, so its
    parameter derivative is 0 -- the rule include/phoskin.h states for dflat.  Negative initial site states make the clip bite at the
    early time points."""
    n = 3
    rng = np.random.default_rng(11)
    th = rng.uniform(0.3, 1.5, size=(4, pm.n_params(0, n)))
    y0 = np.array([1.0, 1.0, -0.5, 0.7, -0.2])
    t = pm.TIME_POINTS
    res = eng.solve_ode_sens_batch("distmod", th, y0, n, t)
    raw = eng.solve_ode_sens_batch("distmod", th, y0, n, t, clip_nonneg=False)
    flat, dflat = res.flat.cpu().numpy(), res.dflat.cpu().numpy()
    fraw, draw = raw.flat.cpu().numpy(), raw.dflat.cpu().numpy()
    clipped = fraw < 0.0
    assert clipped.any() and (~clipped).any()
    assert np.all(flat[clipped] == 0.0) and np.all(dflat[clipped] == 0.0)               # whole P-rows of zeros
    np.testing.assert_array_equal(flat[~clipped], fraw[~clipped])
    np.testing.assert_array_equal(dflat[~clipped], draw[~clipped])
    assert np.abs(draw[clipped]).max() > 0.0                                            # the unclipped derivative there is not zero


def test_sensitivities_follow_flat_postprocessing_and_batched_y0(eng):
    rng = np.random.default_rng(7)
    n, mid = 3, 0
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    B = 9
    th = rng.uniform(0.2, 2.0, size=(B, P)); y0 = rng.uniform(0.5, 2.0, size=(B, S))
    t = pm.TIME_POINTS
    a = eng.solve_ode_sens_batch("distmod", th, y0, n, t, normalize=False)
    b = eng.solve_ode_sens_batch("distmod", th, y0, n, t, normalize=True)
    T = t.size
    # flat = [R(t5..), P(t0..), sites site-major]: the scale of every entry is 1 / y0 of its state
    scale = np.concatenate([np.repeat(y0[:, :1], T - 5, axis=1), np.repeat(y0[:, 1:2], T, axis=1)] + [np.repeat(y0[:, 2 + j:3 + j], T, axis=1) for j in range(n)], axis=1)
    np.testing.assert_allclose(b.flat.cpu().numpy() * scale, a.flat.cpu().numpy(), rtol=1e-12, atol=0)
    np.testing.assert_allclose(b.dflat.cpu().numpy() * scale[:, :, None], a.dflat.cpu().numpy(), rtol=1e-10, atol=1e-14)
    # the entries at t0 are data: zero derivative
    d = a.dflat.cpu().numpy()
    assert np.all(d[:, T - 5, :] == 0.0)                # P(t0)
    # replica independence: a row computed alone gives the same bits
    one = eng.solve_ode_sens_batch("distmod", th[4:5], y0[4:5], n, t)
    assert np.array_equal(one.dflat.cpu().numpy()[0], d[4])


def test_sizes_without_a_sensitivity_kernel_are_refused(eng):
    from phoskintime_amd._capi import PhoskinError
    assert eng.sens_available("distmod", 62) and not eng.sens_available("distmod", 63) and eng.sens_available("succmod", 62) and not eng.sens_available("succmod", 63)
    assert eng.sens_available("randmod", 7) and not eng.sens_available("randmod", 8)
    with pytest.raises(PhoskinError):
        eng.solve_ode_sens_batch("randmod", np.ones((1, pm.n_params(2, 8))), np.ones(pm.n_states(2, 8)), 8, pm.TIME_POINTS)
    with pytest.raises(PhoskinError):
        eng.solve_ode_sens_batch("distmod", np.ones((1, pm.n_params(0, 63))), np.ones(65), 63, pm.TIME_POINTS)
    # failed replicas: flagged, NaN rows, the rest of the batch unaffected
    th = np.ones((3, 10)); th[1, 1] = np.nan
    r = eng.solve_ode_sens_batch("distmod", th, np.ones(5), 3, pm.TIME_POINTS)
    st = r.status.cpu().numpy()
    assert st[0] == 0 and st[2] == 0 and st[1] != 0
    assert np.isnan(r.dflat.cpu().numpy()[1, -1]).all() and np.isfinite(r.dflat.cpu().numpy()[[0, 2]]).all()


@pytest.mark.parametrize("model,n", [("distmod", 4), ("succmod", 3), ("randmod", 2)])
def test_levenberg_marquardt_on_sensitivities_matches_the_differenced_fit(eng, model, n):
    from phoskintime_amd.paramest import multistart as ms
    mid = pm.MODEL_IDS[model]
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(11)
    truth = rng.uniform(0.5, 1.5, size=P)
    y0 = np.ones(S)
    t = pm.TIME_POINTS
    target = eng.solve_ode_batch(model, truth[None], y0, n, t, want_sol=False).flat.cpu().numpy()[0]
    R = 12
    log = model == "randmod"
    P0 = truth * rng.uniform(0.7, 1.4, size=(R, P))
    lb, ub = np.full(P, 1e-3), np.full(P, 10.0)
    if log:
        P0, lb, ub = np.log(P0), np.log(lb), np.log(ub)
    fits = {}
    for jac in ("sens", "fd"):
        fits[jac] = ms.fit_rows_batch(model, n, t, P0, y0, target, bounds=(lb, ub), jacobian=jac, max_iter=200)
    s, f = fits["sens"], fits["fd"]
    # one Jacobian launch per iteration carries n_active integrations instead of n_active * P
    assert s.n_solves < f.n_solves / 2
    pred = lambda fit: eng.solve_ode_batch(model, np.exp(fit.p) if log else fit.p, y0, n, t, want_sol=False).flat.cpu().numpy()
    # both reach the data (the parameters themselves are only weakly identified: compare in data space)
    assert np.median(np.abs(pred(s) - target).max(axis=1)) < 1e-5
    assert np.median(s.cost) <= 10.0 * np.median(f.cost) + 1e-12


def test_batched_damping_levels_take_the_same_steps_as_one_try_per_launch(eng):
    """`trial_levels` = 3 evaluates mu, 4 mu, 16 mu in one launch and takes the first acceptable one: the accepted steps, hence the
    iterates, are those of the sequential rule (up to the rounding of a solve that ran in a batch of another size)."""
    from phoskintime_amd.paramest import multistart as ms
    model, n = "distmod", 3
    mid = pm.MODEL_IDS[model]
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(5)
    truth = rng.uniform(0.5, 1.5, size=P)
    y0 = np.ones(S); t = pm.TIME_POINTS
    target = eng.solve_ode_batch(model, truth[None], y0, n, t, want_sol=False).flat.cpu().numpy()[0]
    target = target * (1.0 + 0.01 * rng.standard_normal(target.size))
    P0 = truth * rng.uniform(0.2, 4.0, size=(16, P))                 # far starts: some first tries are rejected
    lb, ub = np.full(P, 1e-3), np.full(P, 10.0)
    kw = dict(bounds=(lb, ub), max_iter=8, kernel="group")
    a = ms.fit_rows_batch(model, n, t, P0, y0, target, trial_levels=1, **kw)
    b = ms.fit_rows_batch(model, n, t, P0, y0, target, trial_levels=3, **kw)
    np.testing.assert_allclose(a.p, b.p, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(a.cost, b.cost, rtol=1e-6, atol=1e-14)
    assert b.n_launches <= a.n_launches and b.n_solves >= a.n_solves


def test_lm_algebra_on_the_device_takes_the_steps_of_the_host_algebra(eng):
    """`lm_algebra` = "device" (damped normal equations of all levels by the vendor's batched LU, projection and predicted reductions in HBM)
    against "host" (batched numpy, rounds 1-2): with the sensitivity Jacobian the iterates agree to the rounding of two different LU codes
    -- including rows that sit on a bound (free-set masking) and rounds whose first tries are rejected."""
    from phoskintime_amd.paramest import multistart as ms
    model, n = "distmod", 20
    mid = pm.MODEL_IDS[model]
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(7)
    truth = rng.uniform(0.3, 1.5, size=P)
    y0 = np.ones(S); t = pm.TIME_POINTS
    target = eng.solve_ode_batch(model, truth[None], y0, n, t, want_sol=False).flat.cpu().numpy()[0]
    target = np.abs(target * (1.0 + 0.02 * rng.standard_normal(target.size)))
    lb, ub = np.zeros(P), np.full(P, 2.0)                            # a tight box: several variables end on it
    P0 = np.clip(truth * rng.uniform(0.2, 4.0, size=(40, P)), lb, ub)
    kw = dict(bounds=(lb, ub), max_iter=6, jacobian="sens", device_algebra=True)
    a = ms.fit_rows_batch(model, n, t, P0, y0, target, lm_algebra="host", **kw)
    b = ms.fit_rows_batch(model, n, t, P0, y0, target, lm_algebra="device", **kw)
    assert ((a.p == lb) | (a.p == ub)).any()
    np.testing.assert_allclose(a.p, b.p, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a.cost, b.cost, rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(a.JTJ, b.JTJ, rtol=1e-5, atol=1e-9 * np.abs(a.JTJ).max())
    assert a.n_solves == b.n_solves
    with pytest.raises(ValueError):
        ms.fit_rows_batch(model, n, t, P0, y0, target, lm_algebra="gpu", **kw)


def test_sensitivity_edge_shapes(eng):
    """T = 1 (only the initial time: flat is data, zero Jacobian), T <= 5 (no R block in flat), B = 0, one replica, batched y0 at randmod n = 4."""
    import torch
    n, mid = 4, 2
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(3)
    th = rng.uniform(0.3, 2.0, size=(3, P)); y0 = rng.uniform(0.5, 1.5, size=(3, S))
    r1 = eng.solve_ode_sens_batch("randmod", th, y0, n, [0.0])
    assert r1.flat.shape == (3, pm.flatten_observables(mid, np.zeros((1, S)), n).size) and r1.dflat.shape == (3, r1.flat.shape[1], P)
    assert np.all(r1.dflat.cpu().numpy() == 0.0) and int(r1.status.abs().sum()) == 0
    t3 = np.array([0.0, 0.7, 3.0])
    r3 = eng.solve_ode_sens_batch("randmod", th, y0, n, t3, rtol=1e-9, atol=1e-11)
    assert r3.flat.shape[1] == 3 + 3 * n                    # T <= 5: P(t) and the n site series only
    for b in range(3):
        ref = pm.flatten_observables(mid, np.clip(pm.solve_exact_lti(mid, th[b], y0[b], n, t3), 0, None), n)
        assert pm.band_error(r3.flat[b].cpu().numpy(), ref) < 0.1
    one = eng.solve_ode_sens_batch("randmod", th[1], y0[1], n, t3, rtol=1e-9, atol=1e-11)      # 1-D theta: one replica
    assert np.array_equal(one.dflat.cpu().numpy()[0], r3.dflat.cpu().numpy()[1])
    r0 = eng.solve_ode_sens_batch("randmod", np.zeros((0, P)), np.ones(S), n, t3)
    assert r0.flat.shape == (0, 3 + 3 * n) and r0.dflat.shape == (0, 3 + 3 * n, P)
    with pytest.raises(ValueError):
        eng.solve_ode_sens_batch("randmod", np.ones((2, P + 1)), np.ones(S), n, t3)


def test_dropin_jacobian_callable_for_scipy_curve_fit(eng):
    """models.solve_ode_jac through the host-pointer entry point: the `jac=` callable a maintainer hands to scipy.optimize.curve_fit around
    models.solve_ode (paramest/normest.py:167-326).  curve_fit with it reaches the synthetic truth; the callable equals the device-pointer
    entry point bit for bit."""
    from scipy.optimize import curve_fit
    from phoskintime_amd import models
    models.set_model("distmod")
    n = 2
    truth = np.array([1.2, 0.4, 0.9, 0.15, 0.8, 0.3, 0.5, 0.25])
    y0 = np.ones(4); t = pm.TIME_POINTS
    _, target = models.solve_ode(truth, y0, n, t)
    flat, J = models.solve_ode_jac(truth, y0, n, t)
    dev = eng.solve_ode_sens_batch("distmod", truth[None], y0, n, t)
    assert np.array_equal(J, dev.dflat.cpu().numpy()[0]) and np.array_equal(flat, dev.flat.cpu().numpy()[0])
    f = lambda tt, *p: models.solve_ode(np.array(p), y0, n, t)[1]
    jac = lambda tt, *p: models.solve_ode_jac(np.array(p), y0, n, t)[1]
    p0 = truth * 1.3
    popt, _ = curve_fit(f, t, target, p0=p0, jac=jac, bounds=(0.0, 20.0), maxfev=2000)
    assert np.max(np.abs(f(t, *popt) - target)) < 1e-6
    models.set_model("randmod")
    with pytest.raises(Exception):
        models.solve_ode_jac(np.ones(pm.n_params(2, 8)), np.ones(pm.n_states(2, 8)), 8, t)      # no kernel at n = 8: loud, not silent
    fl6, J6 = models.solve_ode_jac(np.full(pm.n_params(2, 6), 0.7), np.ones(pm.n_states(2, 6)), 6, t)       # n = 6, 7: the chunked-column kernel of round 3
    assert fl6.shape == (9 + 14 + 6 * 14,) and J6.shape == (fl6.size, pm.n_params(2, 6)) and np.isfinite(J6).all()
    models.set_model("distmod")


def test_clipped_entries_have_zero_derivative(eng):
    """flat clips negative values to 0 (reference np.clip(sol, 0, None)); the Jacobian follows: zero where the value was clipped, the
    plain derivative elsewhere -- compared with differences of the oracle's CLIPPED output."""
    n, mid = 3, 0
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    rng = np.random.default_rng(9)
    th = rng.uniform(0.3, 2.0, size=(2, P))
    y0 = np.ones(S); y0[1] = -0.8; y0[3] = -0.4                    # protein and one site start negative: clipped for a while
    t = pm.TIME_POINTS
    r = eng.solve_ode_sens_batch("distmod", th, y0, n, t, rtol=1e-9, atol=1e-11)
    flat, d = r.flat.cpu().numpy(), r.dflat.cpu().numpy()
    assert (flat == 0.0).any() and (flat >= 0.0).all()
    for b in range(2):
        ref = _oracle_jac(mid, th[b], y0, n, t)
        clipped = flat[b] == 0.0
        assert np.all(d[b][clipped] == 0.0)
        ref_f = _oracle_flat(mid, th[b], y0, n, t)
        interior = (~clipped) & (ref_f > 1e-6)                      # away from the kink, where the difference quotient is a derivative
        assert np.max(np.abs(d[b][interior] - ref[interior]) / (1.0 + np.abs(ref[interior]))) < SENS_RTOL


_AB_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from phoskintime_amd import batch
from oracle import protein_models as pm
out = {}
for model, n in (("distmod", 4), ("distmod", 14), ("succmod", 3), ("succmod", 14)):
    mid = pm.MODEL_IDS[model]
    rng = np.random.default_rng(31 * mid + n)
    th = rng.uniform(0.2, 2.0, size=(7, pm.n_params(mid, n))); y0 = rng.uniform(0.3, 1.5, size=(7, pm.n_states(mid, n)))
    r = batch.solve_ode_sens_batch(model, th, y0, n, pm.TIME_POINTS, rtol=1e-10, atol=1e-12)
    out[f"{model}_{n}_flat"] = r.flat.cpu().numpy(); out[f"{model}_{n}_dflat"] = r.dflat.cpu().numpy(); out[f"{model}_{n}_status"] = r.status.cpu().numpy()
np.savez(sys.argv[2], **out)
"""


def test_rows_kernel_agrees_with_the_column_kernel(tmp_path):
    """The two sensitivity kernels are independent implementations (columns across lanes with O(S) in-lane solves vs rows across lanes
    with cross-lane arrow / cyclic-reduction solves, chunked columns): at the sizes both cover (PK_SENS_ROWS=1 forces the rows kernel, =2 the column kernel;
    read once per process, hence child processes) they must agree far inside the tolerance both integrate to."""
    import os, subprocess, sys
    root = str(Path(__file__).resolve().parents[1])
    res = {}
    for tag, env in (("cols", "2"), ("rows", "1")):       # 2: the column kernel wherever it exists (n <= 14); 1: the rows kernel everywhere
        f = tmp_path / f"{tag}.npz"
        e = dict(os.environ, PK_SENS_ROWS=env)
        subprocess.run([sys.executable, "-c", _AB_SCRIPT, root, str(f)], check=True, env=e, timeout=600)
        res[tag] = np.load(f)
    for key in res["cols"].files:
        a, b = res["cols"][key], res["rows"][key]
        if key.endswith("status"):
            assert not a.any() and not b.any()
        else:
            assert np.max(np.abs(a - b) / (1.0 + np.abs(a))) < 2e-8, key


def test_rows_kernel_postprocessing_chunks_and_failures(eng):
    """n = 30 (P = 64: ten chunks of seven columns, the last one holding a single column): normalisation, clipping, batched y0, replica
    independence and a failed replica, as test_sensitivities_follow_flat_postprocessing_and_batched_y0 checks them for the column kernel."""
    rng = np.random.default_rng(5)
    n, mid = 30, 0
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    B = 5
    th = rng.uniform(0.2, 2.0, size=(B, P)); y0 = rng.uniform(0.5, 2.0, size=(B, S)); y0[:, 7] = -5.0
    t = pm.TIME_POINTS
    T = t.size
    a = eng.solve_ode_sens_batch("distmod", th, y0, n, t)
    b = eng.solve_ode_sens_batch("distmod", th, y0, n, t, normalize=True)
    raw = eng.solve_ode_sens_batch("distmod", th, y0, n, t, clip_nonneg=False)
    assert not a.status.cpu().numpy().any()
    scale = np.concatenate([np.repeat(y0[:, :1], T - 5, axis=1), np.repeat(y0[:, 1:2], T, axis=1)] + [np.repeat(y0[:, 2 + j:3 + j], T, axis=1) for j in range(n)], axis=1)
    fa, da = a.flat.cpu().numpy(), a.dflat.cpu().numpy()
    np.testing.assert_allclose(b.flat.cpu().numpy() * scale, fa, rtol=1e-12, atol=0)
    np.testing.assert_allclose(b.dflat.cpu().numpy() * scale[:, :, None], da, rtol=1e-10, atol=1e-14)
    fr, dr = raw.flat.cpu().numpy(), raw.dflat.cpu().numpy()
    clipped = fr < 0.0
    assert clipped.any() and np.all(fa[clipped] == 0.0) and np.all(da[clipped] == 0.0) and np.abs(dr[clipped]).max() > 0.0
    np.testing.assert_array_equal(da[~clipped], dr[~clipped])
    assert np.all(da[:, T - 5, :] == 0.0)                                       # P(t0) is data
    plain = eng.solve_ode_batch("distmod", th, y0, n, t, want_sol=False).flat.cpu().numpy()
    assert np.max(np.abs(fa - plain) / (1e-8 + 1e-6 * np.abs(plain))) < 0.1
    one = eng.solve_ode_sens_batch("distmod", th[3:4], y0[3:4], n, t)
    assert np.array_equal(one.dflat.cpu().numpy()[0], da[3]) and np.array_equal(one.flat.cpu().numpy()[0], fa[3])
    th2 = th.copy(); th2[1, 40] = np.nan                                        # a parameter of chunk 5 only: every chunk of the replica must fail
    r = eng.solve_ode_sens_batch("distmod", th2, y0, n, t)
    st = r.status.cpu().numpy()
    assert st[1] != 0 and not st[[0, 2, 3, 4]].any()
    d2 = r.dflat.cpu().numpy()
    at_t0 = [T - 5] + [T - 5 + T + j * T for j in range(n)]                      # rows written before the first step: values at t0 (data)
    assert np.isnan(np.delete(d2[1], at_t0, axis=0)).all() and np.all(d2[1][at_t0] == 0.0)
    assert np.array_equal(d2[0], da[0]) and np.array_equal(d2[4], da[4])


def test_randmod_n6_sensitivities_postprocessing_chunks_and_failures(eng):
    """randmod n = 6 (P = 73: eleven chunks of seven columns, the last with three): normalisation, clipping, batched y0, replica
    independence, a failed replica flagged by every chunk, and the log-space chain rule the fits use."""
    rng = np.random.default_rng(6)
    n, mid = 6, 2
    S, P = pm.n_states(mid, n), pm.n_params(mid, n)
    B = 3
    th = rng.uniform(0.2, 2.0, size=(B, P)); y0 = rng.uniform(0.5, 2.0, size=(B, S)); y0[:, 2] = -50.0
    t = pm.TIME_POINTS
    T = t.size
    a = eng.solve_ode_sens_batch("randmod", th, y0, n, t)
    b = eng.solve_ode_sens_batch("randmod", th, y0, n, t, normalize=True)
    raw = eng.solve_ode_sens_batch("randmod", th, y0, n, t, clip_nonneg=False)
    assert not a.status.cpu().numpy().any()
    scale = np.concatenate([np.repeat(y0[:, :1], T - 5, axis=1), np.repeat(y0[:, 1:2], T, axis=1)] + [np.repeat(y0[:, 2 + j:3 + j], T, axis=1) for j in range(n)], axis=1)
    fa, da = a.flat.cpu().numpy(), a.dflat.cpu().numpy()
    np.testing.assert_allclose(b.flat.cpu().numpy() * scale, fa, rtol=1e-12, atol=0)
    np.testing.assert_allclose(b.dflat.cpu().numpy() * scale[:, :, None], da, rtol=1e-10, atol=1e-14)
    fr, dr = raw.flat.cpu().numpy(), raw.dflat.cpu().numpy()
    clipped = fr < 0.0
    assert clipped.any() and np.all(fa[clipped] == 0.0) and np.all(da[clipped] == 0.0) and np.abs(dr[clipped]).max() > 0.0
    np.testing.assert_array_equal(da[~clipped], dr[~clipped])
    plain = eng.solve_ode_batch("randmod", th, y0, n, t, want_sol=False).flat.cpu().numpy()
    assert np.max(np.abs(fa - plain) / (1e-8 + 1e-6 * np.abs(plain))) < 0.1
    one = eng.solve_ode_sens_batch("randmod", th[1:2], y0[1:2], n, t)
    assert np.array_equal(one.dflat.cpu().numpy()[0], da[1]) and np.array_equal(one.flat.cpu().numpy()[0], fa[1])
    th2 = th.copy(); th2[2, 40] = np.nan
    r = eng.solve_ode_sens_batch("randmod", th2, y0, n, t)
    st = r.status.cpu().numpy()
    assert st[2] != 0 and not st[:2].any() and np.array_equal(r.dflat.cpu().numpy()[0], da[0])
    # the LM driver picks the kernel up: a log-space fit of a 6-site protein converges on exact Jacobians
    from phoskintime_amd.paramest import fit_rows_batch
    truth = rng.uniform(0.3, 1.5, P)
    target = eng.solve_ode_batch("randmod", truth[None], np.ones(S), n, t, want_sol=False).flat.cpu().numpy()[0]
    P0 = np.log(truth * np.exp(0.2 * rng.standard_normal((4, P))))
    fit = fit_rows_batch("randmod", n, t, P0, np.ones(S), target, bounds=(np.full(P, np.log(1e-8)), np.full(P, np.log(20.0))), max_iter=40, jacobian="sens")
    assert (fit.cost < 1e-5).all(), fit.cost            # 73 weakly identified parameters, 40 iterations: five decades below the start (cost ~ 1)
