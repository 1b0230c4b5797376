from .ci import confidence_intervals
