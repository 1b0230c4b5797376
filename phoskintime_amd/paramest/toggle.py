"""``estimate_parameters`` -- the mode switch of the reference (paramest/toggle.py:4-37): one mode, ``normest``."""
from .normest import normest


def estimate_parameters(gene, pr_data, p_data, r_data, init_cond, num_psites, time_points, bounds, bootstraps):
    """-> (model_fits, estimated_params, seq_model_fit, errors, reg_term); ``seq_model_fit`` is the flat model output at the estimate."""
    estimated_params, model_fits, errors, reg_term = normest(gene, pr_data, p_data, r_data, init_cond, num_psites, time_points, bounds, bootstraps)
    return model_fits, estimated_params, model_fits[0][1], errors, reg_term
