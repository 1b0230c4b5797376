"""Population evaluation for the network calibration -- the batched form of ``GlobalODE_MOO._evaluate`` (global_model/optproblem.py:87-160).

The reference subclasses pymoo's ``ElementwiseProblem`` and lets ``StarmapParallelization`` fan single candidates out over a process
pool (runner.py:643-645).  Here a whole population ``X [B, n_var]`` (raw decision vectors) is ONE simulate launch + ONE loss launch:

    F[b] = (loss_sum_m(b) * norm_m) * lambda_m + prior(b)     for the three modalities m,  or  fail_value where the simulation failed.

``evaluate`` has no pymoo dependency; ``as_pymoo_problem`` wraps it in a vectorised ``pymoo.core.problem.Problem`` when pymoo is installed."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import config
from .engine import NetworkEngine


class GlobalODEBatch:
    def __init__(self, eng: NetworkEngine, slices: Optional[Dict], loss_data: Dict, defaults: Dict, lambdas: Dict, time_grid, xl=None, xu=None,
                 fail_value: float = 1e12, loss_mode: int = 0, y0=None, rtol: float = config.ODE_REL_TOL, atol: float = config.ODE_ABS_TOL,
                 max_steps: int = config.ODE_MAX_STEPS, err_norm: str = "max"):
        """``defaults``: physical default parameters (dict with the System.update keys) -- the prior centre of optproblem.py:105-114;
        ``lambdas``: {"protein", "rna", "phospho", "prior"}; ``slices`` is accepted for signature compatibility (the engine's
        candidate layout IS the slice layout of params.init_raw_params)."""
        self.eng = eng
        self.time_grid = np.asarray(time_grid, dtype=np.float64)
        self.loss = eng.make_loss(loss_data, self.time_grid.size)
        self.defaults = eng.pack_params(*(defaults[k] for k in ("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i", "tf_scale")))
        self.lam = (float(lambdas["protein"]), float(lambdas["rna"]), float(lambdas["phospho"]), float(lambdas["prior"]))
        self.fail_value = float(fail_value)
        self.loss_mode = int(loss_mode)
        self.y0 = y0
        self.rtol, self.atol, self.max_steps = rtol, atol, max_steps
        # "max": every component inside rtol / atol (parity-safe on every fixture and on the random full-size populations);
        # "rms": ODEPACK's norm (what the reference's LSODA controls): 1.4-1.7x fewer steps, but up to 2 parity band-widths off on
        # combinatorial populations (tests/test_gpu_network.py) -- opt-in only
        self.err_norm = err_norm
        self.fused = None          # None: not asked yet; True / False: whether the library runs simulate + objective as one launch here
        self.xl, self.xu = xl, xu
        self.n_var, self.n_obj = eng.n_var, 3

    def evaluate_device(self, X) -> torch.Tensor:
        """X [b, n_var] raw (softplus space, params.py:106-132; numpy or a GPU tensor) -> F [b, 3] as a GPU tensor, nothing returns to the
        host.  ONE launch where the integrator can score the observations itself (arrow topologies on the default integrator: no
        trajectory in HBM), else one simulate launch + one loss launch."""
        if self.fused is not False:
            out = self.eng.simulate_objective_batch(self.loss, X, self.time_grid, y0=self.y0, raw=True, rtol=self.rtol, atol=self.atol,
                                                    max_steps=self.max_steps * self.time_grid.size, err_norm=self.err_norm, loss_mode=self.loss_mode,
                                                    defaults=self.defaults, lambdas=self.lam, fail_value=self.fail_value)
            self.fused = out is not None                        # the library's answer for this network / loss data: asked once
            if out is not None:
                return out[1]
        Y, status, _ = self.eng.simulate_batch(X, self.time_grid, y0=self.y0, raw=True, rtol=self.rtol, atol=self.atol,
                                               max_steps=self.max_steps * self.time_grid.size, err_norm=self.err_norm)
        _, F = self.eng.objective_batch(self.loss, Y, loss_mode=self.loss_mode, x=X, raw=True, defaults=self.defaults, lambdas=self.lam,
                                        fail_value=self.fail_value, status=status)
        return F

    def evaluate(self, X) -> np.ndarray:
        """X [B, n_var] raw -> F [B, 3] (host).  Under an initialised ``torch.distributed`` group the candidates are dealt round-robin over
        the ranks (one process per GPU) in decreasing order of a stiffness proxy -- the largest physical rate of the candidate, which is what
        sets its step count -- so that every rank integrates the same mix; every rank evaluates its rows, and ONE all-gather of the 24-byte
        objective rows (RCCL over xGMI) gives every rank the whole F in population order -- what pymoo's non-dominated sorting needs;
        candidates themselves never move."""
        from ..distributed import interleaved_rows, all_gather_interleaved, cost_order, _world
        rank, world = _world()
        if world == 1:
            return self.evaluate_device(X).cpu().numpy()
        total = len(X)
        dev = torch.device("cuda", self.eng.ctx.device)
        Xd = torch.as_tensor(np.asarray(X), device=dev) if not isinstance(X, torch.Tensor) else X.to(dev)
        order = cost_order(Xd.max(dim=1).values)            # softplus is monotone: the largest raw entry is the largest rate
        rows = interleaved_rows(total, rank, world, order)
        Floc = self.evaluate_device(Xd[rows]) if rows.numel() else torch.empty((0, 3), dtype=torch.float64, device=dev)
        return all_gather_interleaved(Floc, total, order).cpu().numpy()

    def close(self):
        if self.loss is not None:
            self.eng.free_loss(self.loss)
            self.loss = None

    def as_pymoo_problem(self):
        from pymoo.core.problem import Problem          # optional dependency of the reference (pymoo 0.6.1.3)

        outer = self

        class _P(Problem):
            def __init__(self):
                super().__init__(n_var=outer.n_var, n_obj=3, xl=outer.xl, xu=outer.xu)

            def _evaluate(self, X, out, *args, **kwargs):
                out["F"] = outer.evaluate(X)

        return _P()
