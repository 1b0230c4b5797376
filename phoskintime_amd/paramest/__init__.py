"""Per-gene parameter estimation on the batched engine (reference: paramest/).  ``normest`` / ``toggle`` / ``core`` carry the reference's
names and argument lists; ``multistart`` holds the batched numerical cores they run on."""
from .multistart import (multistart_candidates, curve_fit_multistart_batch, fit_rows_batch, fit_rows_sharded, find_best_lambda_batch, bootstrap_fit_batch,
                         build_free_bounds, normest_core, FitResult, RowsFit)
from .normest import normest, find_best_lambda, worker_find_lambda, _curve_fit_multistart
from .toggle import estimate_parameters
