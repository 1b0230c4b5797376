"""Per-gene parameter estimation on the batched engine (reference: paramest/normest.py)."""
from .multistart import multistart_candidates, curve_fit_multistart_batch, FitResult
