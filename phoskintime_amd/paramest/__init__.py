"""Per-gene parameter estimation on the batched engine (reference: paramest/normest.py)."""
from .multistart import (multistart_candidates, curve_fit_multistart_batch, fit_rows_batch, fit_rows_sharded, find_best_lambda_batch, bootstrap_fit_batch,
                         build_free_bounds, normest_core, FitResult, RowsFit)
