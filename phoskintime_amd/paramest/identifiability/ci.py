"""Wald confidence intervals of a fit -- drop-in for the reference's ``paramest/identifiability/ci.py:10-84`` (host arithmetic on P
numbers; pinned by tests/golden/pins_normest_*.npz)."""
import logging

import numpy as np
import scipy.stats as stats

from ... import config

logger = logging.getLogger(__name__)


def confidence_intervals(gene, popt, pcov, target, model, alpha_val=0.05):
    """dict(beta_hat, se_lin, df_lin, t_stat, pval, qt_lin, lwr_ci, upr_ci) or None without a covariance.  Residuals are scaled by the
    number of observations before the mean squared error is formed, and that MSE rescales ``pcov`` unless ``config.USE_CUSTOM_WEIGHTS``
    (the reference's convention, kept)."""
    if pcov is None:
        logger.info("No covariance matrix available; cannot compute confidence intervals using linearization.")
        return None
    beta_hat = popt
    target = np.asarray(target, dtype=float)
    df_lin = max(target.size - np.size(beta_hat), 1)
    mse = np.sum(((target - model) / target.size) ** 2) / df_lin
    var = np.diag(pcov) if config.USE_CUSTOM_WEIGHTS else np.diag(pcov * mse)
    se_lin = np.sqrt(var)
    t_stat = beta_hat / se_lin
    qt_lin = stats.t.ppf(1 - alpha_val / 2, df_lin)
    return {'beta_hat': beta_hat, 'se_lin': se_lin, 'df_lin': df_lin, 't_stat': t_stat, 'pval': stats.t.sf(np.abs(t_stat), df_lin) * 2,
            'qt_lin': qt_lin, 'lwr_ci': np.maximum(beta_hat - qt_lin * se_lin, 0), 'upr_ci': beta_hat + qt_lin * se_lin}
