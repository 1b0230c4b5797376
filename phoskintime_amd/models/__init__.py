"""``models`` -- the per-protein simulation boundary of the reference (models/__init__.py:6-12), MI355X engine behind it.

``solve_ode`` is bound to ``models.<ODE_MODEL>.solve_ode`` exactly like the reference does at import time; unlike the
reference the binding can be changed afterwards with ``set_model``."""
import importlib

from .. import config

_VALID = ("distmod", "succmod", "randmod")


def _bind(name: str):
    if name not in _VALID:
        raise ImportError(f"Cannot import model module 'models.{name}'")
    return importlib.import_module(f"{__name__}.{name}")


def model_module_for(name: str):
    """The drop-in module of one model by name, whatever ``solve_ode`` is currently bound to."""
    return _bind(name)


model_module = _bind(config.ODE_MODEL)
solve_ode = model_module.solve_ode
solve_ode_jac = model_module.solve_ode_jac          # not in the reference: flat and its parameter Jacobian from one integration


def set_model(name: str):
    """Re-bind ``models.solve_ode`` (the reference needs a config.toml edit and a fresh interpreter for this)."""
    global model_module, solve_ode, solve_ode_jac
    model_module = _bind(name)
    solve_ode = model_module.solve_ode
    solve_ode_jac = model_module.solve_ode_jac
    config.ODE_MODEL = name
    return model_module
