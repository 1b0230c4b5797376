"""``process_gene`` -- the per-gene driver of the reference (paramest/core.py:17-257) on the batched engine.

Numerical steps kept, in the reference's order: data slices of the three input frames -> ``steady.initial_condition`` -> ``estimate_parameters``
(the lockstep fits of ``paramest.normest``) -> MSE / MAE of the fit -> full solve at the estimate -> EVERY knock-out variant in ONE launch
(the reference loops over 4 (n + 2) ``solve_ode`` calls, core.py:141-154) -> parameter table -> Morris screening (``sensitivity_analysis``,
one launch for its N (D + 1) solves).  Dropped: everything that only draws (PCA / t-SNE / parallel-coordinate / fit / knock-out plots, the
graphviz diagram); their slots in the result dict are ``None``.  ``<gene>_parameters.xlsx`` is written when ``config.OUT_DIR`` is set and
an Excel writer is installed."""
from __future__ import annotations

import logging
import os

import numpy as np

from .. import config
from ..knockout import knockout_batch
from ..models import model_module_for
from ..steady import initial_condition
from .normest import get_param_names
from .toggle import estimate_parameters

logger = logging.getLogger(__name__)


def generate_labels(num_psites: int) -> list:
    """State labels, config/helpers/__init__.py:40-68."""
    if config.ODE_MODEL == 'randmod':
        from itertools import combinations
        return ["R", "P"] + ["P" + "".join(map(str, c)) for k in range(1, num_psites + 1) for c in combinations(range(1, num_psites + 1), k)]
    return ["R", "P"] + [f"P{i}" for i in range(1, num_psites + 1)]


def _knockout_name(setting: dict, psite_values) -> str:
    name = []
    if setting['transcription']:
        name.append("Transcription KO")
    if setting['translation']:
        name.append("Translation KO")
    phospho = setting['phosphorylation']
    if phospho is True:
        name.append("Phospho KO")
    elif isinstance(phospho, list) and phospho:
        name.append(f"PhosphoSite KO {','.join(psite_values[p] for p in phospho)}")
    return "_".join(name or ["WT"])


def process_gene(gene, protein_data, kinase_data, mrna_data, time_points, bounds, bootstraps=0, out_dir=None):
    """One gene end to end -> the reference's result dict (same keys)."""
    import pandas as pd
    out_dir = config.OUT_DIR if out_dir is None else out_dir
    protein_rows = protein_data[protein_data['Psite'].isna() & (protein_data['GeneID'] == gene)]
    gene_data = kinase_data[kinase_data['Gene'] == gene]
    rna_rows = mrna_data[mrna_data['mRNA'] == gene]
    num_psites = gene_data.shape[0]
    psite_values = gene_data['Psite'].values
    Pr_data, P_data, R_data = protein_rows.iloc[:, 2:].values, gene_data.iloc[:, 2:].values, rna_rows.iloc[:, 1:].values
    init_cond = initial_condition(num_psites)

    model_fits, estimated_params, seq_model_fit, errors, regularization_val = estimate_parameters(
        gene, Pr_data, P_data, R_data, init_cond, num_psites, time_points, bounds, bootstraps)
    observed = np.concatenate((R_data.flatten(), Pr_data.flatten(), P_data.flatten()))
    resid = observed - np.asarray(seq_model_fit).flatten()
    mse, mae = float(np.mean(resid ** 2)), float(np.mean(np.abs(resid)))
    final_params = estimated_params[-1]
    names = get_param_names(num_psites)
    gene_psite = {'Protein': gene, **{name: [final_params[i]] for i, name in enumerate(names)}}
    sol_full, _ = model_module_for(config.ODE_MODEL).solve_ode(final_params, init_cond, num_psites, time_points)
    labels = generate_labels(num_psites)

    combos, sol_ko, flat_ko = knockout_batch(final_params, init_cond, num_psites, time_points, normalize=config.NORMALIZE_MODEL_OUTPUT)
    knockout_results = {_knockout_name(c, psite_values): {"knockout_setting": c, "sol_ko": sol_ko[k], "p_fit_ko": flat_ko[k]} for k, c in enumerate(combos)}

    df_params = pd.DataFrame(estimated_params, columns=names)
    df_params.insert(0, "Time", np.asarray(time_points)[:len(estimated_params)])
    df_params['Regularization'] = regularization_val
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        try:
            df_params.to_excel(os.path.join(out_dir, f"{gene}_parameters.xlsx"), index=False)
        except ImportError:                                  # no Excel writer in this image: the table still travels in the result dict
            df_params.to_csv(os.path.join(out_dir, f"{gene}_parameters.csv"), index=False)

    perturbation_analysis = trajectories_w_params = None
    if config.SENSITIVITY_ANALYSIS:
        from ..sensitivity import sensitivity_analysis
        perturbation_analysis, trajectories_w_params = sensitivity_analysis(Pr_data, P_data, R_data, final_params, time_points, num_psites, psite_values,
                                                                            labels, init_cond, gene)
    T = len(config.TIME_POINTS)
    return {"gene": gene, "labels": labels, "psite_labels": psite_values, "estimated_params": estimated_params, "model_fits": sol_full,
            "seq_model_fit": np.asarray(seq_model_fit)[23:].reshape(num_psites, T), "observed_data": P_data, "errors": errors, "final_params": final_params,
            "param_df": df_params, "gene_psite_data": gene_psite, "mse": mse, "mae": mae, "pca_result": None, "ev": None, "tsne_result": None,
            "perturbation_analysis": perturbation_analysis, "perturbation_curves_params": trajectories_w_params, "knockout_results": knockout_results,
            "regularization": regularization_val}


def process_gene_wrapper(gene, protein_data, kinase_data, mrna_data, time_points, bounds, bootstraps, out_dir=None):
    return process_gene(gene=gene, protein_data=protein_data, kinase_data=kinase_data, mrna_data=mrna_data, time_points=time_points, bounds=bounds,
                        bootstraps=bootstraps, out_dir=out_dir)
