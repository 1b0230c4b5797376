"""Reference: steady/initdist.py:9-50 (distributive model, all rates 1)."""
from ._common import unit_rate_steady_state


def initial_condition(num_psites: int) -> list:
    """[R, P, P_1..P_n] with dR = dP = dP_i = 0 at A = B = C = D = S_i = D_i = 1."""
    return unit_rate_steady_state("distmod", num_psites).tolist()
