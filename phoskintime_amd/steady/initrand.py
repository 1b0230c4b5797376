"""Reference: steady/initrand.py:9-77 (random model, all rates 1).

The reference orders the phospho states by subset SIZE, then lexicographically (itertools.combinations, initrand.py:24-28), whereas
models/randmod.py orders them by bit mask; the list is nevertheless used directly as y0 of the bit-mask ODE (paramest/core.py:83-111).
With unit rates the steady state depends only on the number of phosphorylated sites, and the drop-in returns the reference's order."""
from itertools import combinations

import numpy as np

from ._common import unit_rate_steady_state


def initial_condition(num_psites: int) -> list:
    n = int(num_psites)
    y = unit_rate_steady_state("randmod", n)            # [R, P, X_mask=1 .. X_mask=2^n-1]
    out = [float(y[0]), float(y[1])]
    for k in range(1, n + 1):
        for comb in combinations(range(1, n + 1), k):
            mask = sum(1 << (s - 1) for s in comb)
            out.append(float(y[1 + mask]))
    return out
