"""Per-protein Morris sensitivity (reference: sensitivity/analysis.py), batched: the N*(D+1) solves of one screening run
are ONE kernel launch with the scalar model output fused into the solve."""
from .analysis import (compute_bound, define_sensitivity_problem_ds, define_sensitivity_problem_rand, _compute_Y,
                       sensitivity_analysis_batch, _perturb_solve, _sensitivity_analysis)
from . import morris

sensitivity_analysis = _sensitivity_analysis          # the name paramest/core.py imports (sensitivity/__init__.py:3)
