import numpy as np

from .. import batch


def unit_rate_steady_state(model: str, num_psites: int) -> np.ndarray:
    """y* for theta = (A, B, C, D, S..., D...) = ones.  Raises ValueError like the reference when no steady state exists."""
    n = int(num_psites)
    if n < 1:
        raise ValueError("num_psites must be >= 1")
    P = batch.n_params(model, n)
    y, status = batch.steady_state_batch(model, np.ones((1, P)), n)
    if int(status[0]) != 0:
        raise ValueError("Failed to find steady-state conditions")
    return y[0].cpu().numpy()
