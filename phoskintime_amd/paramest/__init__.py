"""Per-gene parameter estimation on the batched engine (reference: paramest/).  ``normest`` / ``toggle`` / ``core`` carry the reference's
names and argument lists; ``multistart`` holds the batched numerical cores they run on."""
from .multistart import (multistart_candidates, curve_fit_multistart_batch, fit_rows_batch, fit_rows_sharded, find_best_lambda_batch, bootstrap_fit_batch,
                         build_free_bounds, normest_core, FitResult, RowsFit)
from . import normest, toggle          # modules, as in the reference (``from paramest.normest import normest``); not re-exported as functions
