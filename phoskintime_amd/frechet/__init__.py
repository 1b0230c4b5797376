"""``frechet`` -- discrete Frechet distance (reference frechet/distance.py:9-56) on the batched engine."""
from .distance import frechet_distance, frechet_batch
