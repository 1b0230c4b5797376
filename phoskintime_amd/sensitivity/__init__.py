"""Per-protein Morris sensitivity (reference: sensitivity/analysis.py), batched: the N*(D+1) solves of one screening run
are ONE kernel launch with the scalar model output fused into the solve."""
from .analysis import (compute_bound, define_sensitivity_problem_ds, define_sensitivity_problem_rand, _compute_Y,
                       sensitivity_analysis_batch)
from . import morris
