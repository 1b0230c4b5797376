"""Successive (ordered) phosphorylation model -- drop-in for the reference's ``models/succmod.py`` (model id 1).

``ode_core`` succmod.py:9, ``unpack_params`` :94, ``solve_ode`` :114."""
import numpy as np

from ._common import pack_params, rhs_host, solve_host, solve_jac_host

MODEL_ID = 1


def ode_core(y, t, A, B, C, D, S_rates, D_rates):
    """dy/dt of the successive model (reference succmod.py:9-90, incl. the one-site branch :59-63), on the GPU."""
    n = np.asarray(S_rates).shape[0]
    return rhs_host(MODEL_ID, pack_params(A, B, C, D, S_rates, D_rates), y, n)


def unpack_params(params, num_psites):
    """Reference succmod.py:94-112."""
    params = np.asarray(params)
    return (params[0], params[1], params[2], params[3],
            params[4:4 + num_psites], params[4 + num_psites:4 + 2 * num_psites])


def solve_ode(params, init_cond, num_psites, t):
    """Reference succmod.py:114-152 contract: (sol, flat)."""
    return solve_host(MODEL_ID, params, init_cond, num_psites, t)


def solve_ode_jac(params, init_cond, num_psites, t):
    """(flat, d flat / d params [F, P]) from ONE integration (forward sensitivities) -- an addition to the reference's surface: the ``jac=``
    callable for scipy.optimize.curve_fit around ``solve_ode`` (paramest/normest.py:167-326 lets curve_fit difference it)."""
    return solve_jac_host(MODEL_ID, params, init_cond, num_psites, t)
