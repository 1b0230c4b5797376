"""In-silico knock-outs (reference knockout/helper.py:5-62): parameter zeroing, host side, plus the batched re-solve that
replaces the serial loop of 4 * (n + 2) ``solve_ode`` calls in paramest/core.py:148-154."""
import itertools

import numpy as np


def _apply_knockout(base_params: np.ndarray, knockout_targets: dict, num_psites: int) -> np.ndarray:
    """Copy of ``base_params`` with A (transcription), C (translation) and/or S-rates (phosphorylation) set to zero."""
    params = np.array(base_params, dtype=float, copy=True)
    if knockout_targets.get('transcription', False):
        params[0] = 0.0
    if knockout_targets.get('translation', False):
        params[2] = 0.0
    if 'phosphorylation' in knockout_targets:
        k = knockout_targets['phosphorylation']
        if isinstance(k, bool) and k:
            params[4:4 + num_psites] = 0.0
        elif isinstance(k, (list, tuple)):
            for idx in k:
                if 0 <= idx < num_psites:
                    params[4 + idx] = 0.0
    return params


def _generate_knockout_combinations(num_psites: int):
    """{transcription} x {translation} x {none, all sites, each single site}: 4 * (n + 2) dictionaries."""
    phospho = [False, True] + [[i] for i in range(num_psites)]
    return [{'transcription': a, 'translation': b, 'phosphorylation': c}
            for a, b, c in itertools.product([False, True], [False, True], phospho)]


def knockout_batch(final_params, init_cond, num_psites, time_points, model=None, **solver_kw):
    """All knock-out variants of one fitted parameter vector in one launch -> (combinations, sol [K, T, S], flat [K, F])."""
    from .. import batch, config
    combos = _generate_knockout_combinations(num_psites)
    thetas = np.stack([_apply_knockout(final_params, c, num_psites) for c in combos])
    res = batch.solve_ode_batch(config.ODE_MODEL if model is None else model, thetas, init_cond, num_psites, time_points, **solver_kw)
    return combos, res.sol.cpu().numpy(), res.flat.cpu().numpy()
