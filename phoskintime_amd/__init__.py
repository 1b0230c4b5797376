"""phoskintime_amd -- MI355X-native batched stiff-ODE engine behind PhosKinTime's per-protein call surface.

Public surface (mirrors the reference's names):

  phoskintime_amd.models.solve_ode / models.{distmod,succmod,randmod}.{solve_ode,ode_core|ode_system,unpack_params}
  phoskintime_amd.batch.solve_ode_batch / rhs_batch / jacobian_batch      (batched siblings, torch tensors on HBM)
"""
__version__ = "0.1.0"
