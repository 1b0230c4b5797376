"""Random (2^n - 1 phospho-state) model -- drop-in for the reference's ``models/randmod.py`` (model id 2).

``_precompute_indices`` randmod.py:9, ``unpack_params`` :88, ``ode_system`` :122, ``solve_ode`` :249.  The engine
derives the bit-transition structure on the fly (one lane per state, XOR neighbours), so the int64 tables are only
kept for signature compatibility."""
from functools import lru_cache

import numpy as np

from ._common import pack_params, rhs_host, solve_host, solve_jac_host

MODEL_ID = 2


@lru_cache(maxsize=None)
def _precompute_indices(num_sites):
    """Transition tables with the reference's layout (randmod.py:9-85): mono_idx[n], forward/drop [m, n] (1-based
    masks, -1 padded), fcounts/dcounts [m]; m = 2^n - 1.  Host-side bookkeeping only."""
    n = num_sites
    m = (1 << n) - 1
    states = np.arange(1, m + 1, dtype=np.int64)[:, None]
    bits = (np.int64(1) << np.arange(n, dtype=np.int64))[None, :]
    has = (states & bits) != 0
    forward = -np.ones((m, n), dtype=np.int64)
    drop = -np.ones((m, n), dtype=np.int64)
    for s in range(m):
        f = (states[s] | bits[0])[~has[s]]
        d = (states[s] & ~bits[0])[has[s]]
        forward[s, :f.size] = f
        drop[s, :d.size] = d
    mono_idx = (np.int64(1) << np.arange(n, dtype=np.int64)) - 1
    return mono_idx, forward, drop, (~has).sum(1).astype(np.int64), has.sum(1).astype(np.int64)


def unpack_params(params, num_sites):
    """[A, B, C, D, S_1..S_n, Ddeg_1..Ddeg_m] -> (A, B, C, D, S, Ddeg)   (reference randmod.py:88-119)."""
    params = np.asarray(params)
    n = num_sites
    m = (1 << n) - 1
    return params[0], params[1], params[2], params[3], np.array(params[4:4 + n], dtype=float), np.array(params[4 + n:4 + n + m], dtype=float)


def ode_system(y, t, A, B, C, D, num_sites, S, Ddeg, mono_idx=None, forward=None, drop=None, fcounts=None, dcounts=None):
    """dy/dt of the random model (reference randmod.py:122-247, incl. the lowest-set-bit rate rule at :201), on the GPU.
    The table arguments are accepted and ignored."""
    return rhs_host(MODEL_ID, pack_params(A, B, C, D, S, Ddeg), y, int(num_sites))


def solve_ode(popt, y0, num_sites, t):
    """Reference randmod.py:249-305 contract: (sol, [R(t5..), P, first num_sites phospho columns site-major])."""
    return solve_host(MODEL_ID, popt, y0, num_sites, t)


def solve_ode_jac(popt, y0, num_sites, t):
    """(flat, d flat / d params [F, P]) from ONE integration (forward sensitivities) -- an addition to the reference's surface: the ``jac=``
    callable for scipy.optimize.curve_fit around ``solve_ode`` (paramest/normest.py:167-326 lets curve_fit difference it)."""
    return solve_jac_host(MODEL_ID, popt, y0, num_sites, t)
