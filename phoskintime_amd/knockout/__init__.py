from .helper import _apply_knockout, _generate_knockout_combinations, knockout_batch
