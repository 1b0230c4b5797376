"""Residual weightings of the per-protein fits -- drop-in for the reference's ``models/weights.py`` (host-side data preparation: a few
dozen numbers per gene, once per fit; nothing here is worth a kernel).

  early_emphasis(pr_data, p_data, time_points, num_psites)      models/weights.py:10-76    (vectorised; the reference loops under numba)
  get_protein_weights(gene, input1_path, input2_path)           models/weights.py:79-145   (pandas merge of the two measurement tables)
  full_weight(p_data_weight, use_regularization, reg_len)       models/weights.py:148-163
  get_weight_options(target, t_target, num_psites, ...)         models/weights.py:166-240  (17 schemes; only "uncertainties_from_data"
                                                                                            unless config.USE_CUSTOM_WEIGHTS)

The two table paths default to ``config.INPUT1_WSTD_PATH`` / ``config.INPUT2_PATH`` (the reference hard-wires paths relative to its own
source tree, ``processing/input1_wstd.csv`` and ``data/input2.csv``).  Pinned by tests/golden/pins_normest_*.npz (reference-run)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .. import config


def early_emphasis(pr_data, p_data, time_points, num_psites):
    """Weights [T + n T]: 1 / (|value| + 1e-5), times 1 / (dt + 1e-5) for the first eight time points after t0 (protein row first, then
    the phospho rows site-major)."""
    p = np.asarray(p_data, dtype=float)
    pr = np.asarray(pr_data, dtype=float)
    p = p.reshape(1, -1) if p.ndim == 1 else p
    pr = pr.reshape(1, -1) if pr.ndim == 1 else pr
    t = np.asarray(time_points, dtype=float)
    T = t.size
    time_w = np.ones(T)
    time_w[1:] = 1.0 / (np.diff(t) + 1e-5)
    time_w[8:] = 1.0                                   # j >= 8: no time factor
    rows = np.concatenate([pr[:1, :T], p[:num_psites, :T]], axis=0)
    return ((1.0 / (np.abs(rows) + 1e-5)) * time_w[None, :]).reshape(-1)


def get_protein_weights(gene, input1_path=None, input2_path=None):
    """Measurement standard deviations x1_std .. x14_std of one gene, protein row first and then its phospho rows in the order of
    ``input2`` -- flattened [14 (1 + n)].  Raises ValueError for an unknown gene or a (GeneID, Psite) pair without uncertainties."""
    import pandas as pd
    input1 = pd.read_csv(Path(config.INPUT1_WSTD_PATH if input1_path is None else input1_path))
    input2 = pd.read_csv(Path(config.INPUT2_PATH if input2_path is None else input2_path))
    input1.columns = input1.columns.str.strip()
    input2.columns = input2.columns.str.strip()
    std_columns = [f'x{i}_std' for i in range(1, 15)]
    rows2 = input2[input2['GeneID'] == gene]
    if rows2.empty:
        raise ValueError(f"No entries for GeneID {gene} found in input2.csv")
    in1 = input1[input1['GeneID'] == gene].copy()
    in1['Psite'] = in1['Psite'].replace('', pd.NA)
    sites = rows2['Psite'].replace('', pd.NA)
    by_site = in1[in1['Psite'].notna()].drop_duplicates('Psite').set_index('Psite')
    missing = [s for s in sites if pd.isna(s) or s not in by_site.index]
    if missing:
        raise ValueError(f"Missing (GeneID, Psite) pairs for {gene} in input1_wstd.csv:\n{missing}")
    blocks = []
    tf = in1[in1['Psite'].isna()]
    if tf.shape[0] == 1:                                # the protein-level row, when the table has exactly one
        blocks.append(tf[std_columns].to_numpy(dtype=float))
    blocks.append(by_site.loc[list(sites), std_columns].to_numpy(dtype=float))
    out = np.concatenate(blocks, axis=0)
    if np.isnan(out).any():
        raise ValueError(f"Missing (GeneID, Psite) pairs for {gene} in input1_wstd.csv:\nNaN uncertainties")
    return out.flatten()


def full_weight(p_data_weight, use_regularization, reg_len):
    """[ones(9) (the RNA block) | weights of the protein + phospho block | ones(reg_len) when the ridge rows are fitted]."""
    parts = [np.ones(9), np.asarray(p_data_weight, dtype=float)]
    if use_regularization:
        parts.append(np.ones(reg_len))
    return np.concatenate(parts)


def get_weight_options(target, t_target, num_psites, use_regularization, reg_len, early_weights, ms_gauss_weights):
    """name -> sigma vector for ``curve_fit``.  With ``config.USE_CUSTOM_WEIGHTS`` false (the default, config.toml:207) only the
    measurement uncertainties are offered."""
    from scipy.ndimage import uniform_filter1d
    fw = lambda w: full_weight(w, use_regularization, reg_len)
    if not config.USE_CUSTOM_WEIGHTS:
        return {"uncertainties_from_data": fw(ms_gauss_weights)}
    target = np.asarray(target, dtype=float)
    tail = target[9:]
    ti = np.tile(np.arange(1, len(t_target) + 1), num_psites)
    floor = lambda a: np.maximum(a, 1e-5)
    sqrt_signal = np.sqrt(floor(np.abs(target)))
    flat_pen = 1 / floor(np.abs(np.gradient(target))) if len(target) >= 2 else 1 / floor(np.abs(target))
    L = len(ti)
    return {
        "inverse": fw(1 / floor(np.abs(tail))),
        "exponential_decay": fw(np.exp(-0.5 * tail)),
        "inverse_log_scale": fw(1 / floor(np.log1p(np.abs(target))[9:])),
        "inverse_time_diff": fw(1 / floor(np.abs(np.diff(tail, prepend=tail[0])))),
        "inverse_moving_avg": fw(1 / floor(np.abs(tail - uniform_filter1d(tail, 3)))),
        "sigmoid_decay": fw(1 / (1 + np.exp((ti - 5)))),
        "exponential_early_decay": fw(np.exp(-0.5 * ti)),
        "polynomial_time_decay": fw(1 / (1 + 0.5 * ti)),
        "signal_noise": fw(1 / sqrt_signal[9:]),
        "inverse_variance": fw(1 / (floor(np.abs(tail)) ** 0.7)),
        "flat_penalty": fw(flat_pen[9:]) if flat_pen.shape[0] == target.shape[0] else flat_pen,
        "steady_decay": fw(np.exp(-0.1 * ti)),
        "inverse_square_root_data": fw(1 / sqrt_signal[9:]),
        "early_moderate_decay": fw(np.linspace(1.0, 0.3, L)),
        "early_steep_decay": fw(np.concatenate([np.full(min(8, L), 0.05), np.full(min(2, max(L - 8, 0)), 0.2), np.ones(max(L - 10, 0))])),
        "early_emphasis": fw(early_weights),
        "uncertainties_from_data": fw(ms_gauss_weights),
    }
