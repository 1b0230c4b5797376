"""Network (global_model) path of the reference on the MI355X engine: batched right-hand side, analytic Jacobian, softplus unpack,
batched integration, observables, losses / objectives and the Morris driver."""
from .engine import NetworkEngine
from . import config
