"""``steady`` -- initial conditions of the per-protein fits (reference steady/__init__.py:1-14): ``initial_condition(num_psites)`` bound
to the module of the configured ``ODE_MODEL``, as ``paramest/core.py:83`` calls it.

The reference poses the steady state of the model with every rate fixed to 1 as an SLSQP feasibility problem.  All three models are
affine (dy/dt = J y + b), so the steady state is the linear solve J y* = -b; it runs on the GPU through
``pk_steady_state_protein_batch`` (arrow / tridiagonal elimination, dense inverse for the random model), which also serves arbitrary
per-replica parameters (``phoskintime_amd.batch.steady_state_batch``)."""
from .. import config
from . import initdist, initsucc, initrand

_IMPL = {"distmod": initdist.initial_condition, "succmod": initsucc.initial_condition, "randmod": initrand.initial_condition}


def initial_condition(num_psites: int) -> list:
    """Steady state [R, P, P_sites...] for the currently configured model (``config.ODE_MODEL``; the reference binds at import)."""
    try:
        return _IMPL[config.ODE_MODEL](num_psites)
    except KeyError:
        raise ValueError(f"Unsupported ODE_MODEL: {config.ODE_MODEL}") from None
