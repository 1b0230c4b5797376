"""Distributive phosphorylation model -- drop-in for the reference's ``models/distmod.py`` on the MI355X engine.

Same names and signatures (``ode_core`` distmod.py:7, ``unpack_params`` :68, ``solve_ode`` :93); arithmetic runs in
libphoskin_hip.so (model id 0)."""
import numpy as np

from ._common import pack_params, rhs_host, solve_host, solve_jac_host

MODEL_ID = 0


def ode_core(y, t, A, B, C, D, S_rates, D_rates):
    """dy/dt of the distributive model (reference distmod.py:7-65), evaluated on the GPU."""
    n = np.asarray(S_rates).shape[0]
    return rhs_host(MODEL_ID, pack_params(A, B, C, D, S_rates, D_rates), y, n)


def unpack_params(params, num_psites):
    """[A, B, C, D, S_1..S_n, D_1..D_n] -> (A, B, C, D, S_rates, D_rates)   (reference distmod.py:68-91)."""
    params = np.asarray(params)
    return (params[0], params[1], params[2], params[3],
            params[4:4 + num_psites], params[4 + num_psites:4 + 2 * num_psites])


def solve_ode(params, init_cond, num_psites, t):
    """(sol[T, S] clipped >= 0 (and / y0 if NORMALIZE_MODEL_OUTPUT), flat = [R(t5..), P(t0..), sites site-major]).
    Reference distmod.py:93-134 (odeint at SciPy defaults); here the engine's default integrator: adaptive LRP12 (order-11 L-stable resolvent method, include/phoskin.h) at rtol 1e-6 / atol 1e-8."""
    return solve_host(MODEL_ID, params, init_cond, num_psites, t)


def solve_ode_jac(params, init_cond, num_psites, t):
    """(flat, d flat / d params [F, P]) from ONE integration (forward sensitivities) -- an addition to the reference's surface: the ``jac=``
    callable for scipy.optimize.curve_fit around ``solve_ode`` (paramest/normest.py:167-326 lets curve_fit difference it)."""
    return solve_jac_host(MODEL_ID, params, init_cond, num_psites, t)
