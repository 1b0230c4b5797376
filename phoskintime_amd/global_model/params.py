"""Drop-in names of the reference's ``global_model/params.py``: the optimiser's decision vector.

  softplus / inv_softplus      global_model/utils.py:229-253     (the batched softplus runs on the GPU: ``NetworkEngine.unpack_batch``)
  init_raw_params              global_model/params.py:24-103     defaults -> (theta0, slices, xl, xu) in raw (inverse-softplus) space
  unpack_params                global_model/params.py:106-132    one raw vector -> dict of physical parameters (host, single vector)

The slice layout ``[c_k | A_i | B_i | C_i | D_i | Dp_i | E_i | tf_scale]`` is the engine's candidate row layout, so ``X_raw`` rows go to
``NetworkEngine.simulate_batch(..., raw=True)`` unchanged."""
from __future__ import annotations

import numpy as np

from . import config

_KEYS = ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i")


def softplus(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(over="ignore"):
        return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def inv_softplus(y):
    y = np.maximum(np.asarray(y, dtype=np.float64), 1e-12)
    return np.log(np.expm1(y))


def init_raw_params(defaults, custom_bounds=None):
    custom_bounds = custom_bounds or {}
    vecs, slices, lo, hi, curr = [], {}, [], [], 0
    for k in _KEYS + ("tf_scale",):
        raw = inv_softplus(np.atleast_1d(np.asarray(defaults[k], dtype=np.float64)))
        vecs.append(raw)
        slices[k] = slice(curr, curr + raw.size)
        curr += raw.size
        pmin, pmax = custom_bounds[k] if k in custom_bounds else config.BOUNDS_CONFIG[k]
        lo += [float(inv_softplus(np.array([pmin]))[0])] * raw.size
        hi += [float(inv_softplus(np.array([pmax]))[0])] * raw.size
    return np.concatenate(vecs), slices, np.array(lo, dtype=float), np.array(hi, dtype=float)


def unpack_params(theta, slices):
    theta = np.asarray(theta, dtype=np.float64)
    out = {k: softplus(theta[slices[k]]) for k in _KEYS}
    out["tf_scale"] = softplus(theta[slices["tf_scale"]])[0]
    return out
