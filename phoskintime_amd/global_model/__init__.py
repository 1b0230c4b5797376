"""Network (global_model) path of the reference on the MI355X engine -- round 1: batched right-hand side, analytic Jacobian
and softplus unpack.  The batched implicit integrator (a17) is the next row."""
from .engine import NetworkEngine
from . import config
