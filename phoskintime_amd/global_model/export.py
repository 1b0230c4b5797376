"""On-disk artefacts the callers downstream of the hot path read, in the reference's file names and layouts (host-side, tiny):

  pareto_X.npy / pareto_F.npy / pareto_F.csv                     global_model/runner.py:734-743
  sensitivity_indices.csv                                        global_model/sensitivity.py:266-281
  pred_{prot,rna,phospho}_picked.csv, fitted_params_picked.json, picked_objectives.json      global_model/runner.py:912-929
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Sequence

import numpy as np


def save_pareto(output_dir: str, X, F) -> None:
    import pandas as pd
    os.makedirs(output_dir, exist_ok=True)
    X = np.asarray(X); F = np.asarray(F, float)
    np.save(os.path.join(output_dir, "pareto_X.npy"), X)
    np.save(os.path.join(output_dir, "pareto_F.npy"), F)
    pd.DataFrame(F, columns=["prot_mse", "rna_mse", "phospho_mse"]).to_csv(os.path.join(output_dir, "pareto_F.csv"), index=False)


def save_sensitivity_indices(output_dir: str, problem: Dict, Si: Dict) -> str:
    """Parameter, mu_star, sigma, mu_star_conf -- sorted by mu_star, most influential first."""
    import pandas as pd
    os.makedirs(output_dir, exist_ok=True)
    df = pd.DataFrame({"Parameter": list(problem["names"]), "mu_star": np.asarray(Si["mu_star"]), "sigma": np.asarray(Si["sigma"]),
                       "mu_star_conf": np.asarray(Si["mu_star_conf"])}).sort_values("mu_star", ascending=False)
    path = os.path.join(output_dir, "sensitivity_indices.csv")
    df.to_csv(path, index=False)
    return path


def save_picked(output_dir: str, params: Dict, F, picked_index: int, lambdas: Sequence[float] = (1.0, 1.0, 1.0), df_prot=None, df_rna=None,
                df_pho=None) -> Dict:
    """The picked Pareto solution: parameter dict (arrays -> lists, scalars -> float), its three objectives and their weighted sum,
    and the prediction tables of ``simulate_and_measure`` if given."""
    os.makedirs(output_dir, exist_ok=True)
    for df, name in ((df_prot, "pred_prot_picked.csv"), (df_rna, "pred_rna_picked.csv"), (df_pho, "pred_phospho_picked.csv")):
        if df is not None:
            df.to_csv(os.path.join(output_dir, name), index=False)
    p_out = {k: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else float(v)) for k, v in params.items()}
    with open(os.path.join(output_dir, "fitted_params_picked.json"), "w") as f:
        json.dump(p_out, f, indent=2)
    F = np.asarray(F, float)
    i = int(picked_index)
    picked = {"prot_mse": float(F[i, 0]), "rna_mse": float(F[i, 1]), "phospho_mse": float(F[i, 2]),
              "scalar_score": float(lambdas[0] * F[i, 0] + lambdas[1] * F[i, 1] + lambdas[2] * F[i, 2])}
    with open(os.path.join(output_dir, "picked_objectives.json"), "w") as f:
        json.dump(picked, f, indent=2)
    return picked
