from .helper import _apply_knockout, _generate_knockout_combinations, knockout_batch

# names the reference's callers import (knockout/__init__.py:3-4; paramest/core.py:5)
apply_knockout = _apply_knockout
generate_knockout_combinations = _generate_knockout_combinations
