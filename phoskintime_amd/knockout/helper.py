"""In-silico knock-outs (reference knockout/helper.py:5-62): parameter zeroing, host side, plus the batched re-solve that
replaces the serial loop of 4 * (n + 2) ``solve_ode`` calls in paramest/core.py:148-154."""
import itertools

import numpy as np


def knockout_mask(knockout_targets: dict, num_psites: int, n_params: int) -> np.ndarray:
    """Boolean mask [n_params] of the entries a knock-out zeroes: A (index 0) for transcription, C (index 2) for translation, the S-rate
    block 4 .. 4 + n for phosphorylation (``True``: every site; a list / tuple: those sites, out-of-range indices ignored)."""
    pos = np.arange(int(n_params))
    k = knockout_targets.get('phosphorylation', False)
    sites = np.arange(num_psites) if (isinstance(k, bool) and k) else np.asarray(k if isinstance(k, (list, tuple)) else [], dtype=np.int64).reshape(-1)
    sites = sites[(sites >= 0) & (sites < num_psites)]
    return ((pos == 0) & bool(knockout_targets.get('transcription', False))) | ((pos == 2) & bool(knockout_targets.get('translation', False))) \
        | np.isin(pos, 4 + sites)


def _apply_knockout(base_params: np.ndarray, knockout_targets: dict, num_psites: int) -> np.ndarray:
    """Copy of ``base_params`` with the masked rates set to zero (reference knockout/helper.py:5-36: same result, one mask operation)."""
    params = np.array(base_params, dtype=float, copy=True)
    return np.where(knockout_mask(knockout_targets, num_psites, params.size), 0.0, params)


def _generate_knockout_combinations(num_psites: int):
    """{transcription} x {translation} x {none, all sites, each single site}: 4 * (n + 2) dictionaries."""
    phospho = [False, True] + [[i] for i in range(num_psites)]
    return [{'transcription': a, 'translation': b, 'phosphorylation': c}
            for a, b, c in itertools.product([False, True], [False, True], phospho)]


def knockout_batch(final_params, init_cond, num_psites, time_points, model=None, **solver_kw):
    """All knock-out variants of one fitted parameter vector in one launch -> (combinations, sol [K, T, S], flat [K, F])."""
    from .. import batch, config
    combos = _generate_knockout_combinations(num_psites)
    base = np.asarray(final_params, dtype=float)
    masks = np.stack([knockout_mask(c, num_psites, base.size) for c in combos])
    thetas = np.where(masks, 0.0, base[None, :])
    res = batch.solve_ode_batch(config.ODE_MODEL if model is None else model, thetas, init_cond, num_psites, time_points, **solver_kw)
    return combos, res.sol.cpu().numpy(), res.flat.cpu().numpy()
