"""Reference: steady/initsucc.py:9-55 (successive model, all rates 1).

Note the reference's residual function is the DISTRIBUTIVE one (initsucc.py:39-41 repeats initdist.py's equations: every site is fed
from P, not from its predecessor), so the initial condition of a successive-model fit is the distributive steady state.  Reproduced."""
from ._common import unit_rate_steady_state


def initial_condition(num_psites: int) -> list:
    return unit_rate_steady_state("distmod", num_psites).tolist()
