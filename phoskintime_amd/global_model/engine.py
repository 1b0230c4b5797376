"""``NetworkEngine``: the static network of a reference ``System`` (global_model/network.py:199-526) resident in HBM, plus
batched evaluation of candidates through the C ABI (``pk_network_*``, include/phoskin.h)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _capi
from ..batch import get_context, _dev_f64, _ptr


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class NetworkEngine:
    """Device-resident topology + kinase input of one network.  Candidates are rows ``x`` of length ``n_var`` =
    ``n_K + 5 N + total_sites + 1`` = ``[c_k | A_i | B_i | C_i | D_i | Dp_i | E_i | tf_scale]`` (params.py:60-96)."""

    def __init__(self, model: int, offset_y, offset_s, n_sites, W_indptr, W_indices, W_data, TF_indptr, TF_indices, TF_data,
                 tf_deg, driver_map, kin_grid, kin_Kmat, device: Optional[int] = None):
        self._device = get_context(device).device
        self.model = int(model)
        self._keep = [_i32(offset_y), _i32(offset_s), _i32(n_sites), _i32(W_indptr), _i32(W_indices), _f64(W_data),
                      _i32(TF_indptr), _i32(TF_indices), _f64(TF_data), _f64(tf_deg), _i32(driver_map), _f64(kin_grid), _f64(kin_Kmat)]
        (oy, os_, ns, wp, wi, wd, tp, ti, td, deg, drv, grid, kmat) = self._keep
        self.N = int(oy.size); self.n_K = int(kmat.shape[0]); self.total_sites = int(ns.sum()); self.n_grid = int(grid.size)
        if kmat.shape != (self.n_K, self.n_grid):
            raise ValueError("kin_Kmat must be [n_K, n_grid]")
        d = _capi.NetworkDesc(self.model, self.N, self.n_K, self.total_sites, self.n_grid,
                              *(a.ctypes.data for a in (oy, os_, ns, wp, wi, wd, tp, ti, td, deg, drv, grid, kmat)))
        self._h = self.ctx.lib.pk_network_create(self.ctx.handle, C.byref(d))
        if not self._h:
            raise _capi.PhoskinError("pk_network_create failed: " + (self.ctx.lib.pk_last_error(self.ctx.handle) or b"").decode())
        self.S = self.ctx.lib.pk_network_n_states(self._h)
        self.n_var = self.ctx.lib.pk_network_n_var(self._h)

    @property
    def ctx(self):
        """The CALLING thread's context on this engine's GPU (``batch.get_context``): the network handle is read-only device data and may
        be used from any thread, the context (stream, staging buffers, last error) may not be shared."""
        return get_context(self._device)

    # ------------------------------------------------------------------ constructors from reference objects
    @staticmethod
    def desc_from_system(sys) -> dict:
        """Host-only packing of a reference ``global_model.network.System`` (duck-typed: only its array attributes and the index maps
        are read) into the constructor's keyword arrays.  ``driver_map`` follows network.py:454-469: a kinase that is also a protein
        drives that protein's block; an orphan TF is driven by its proxy kinase (``idx.p2i`` already redirects the orphan)."""
        idx = sys.idx
        drv = np.full(idx.N, -1, dtype=np.int32)
        for k_name in idx.kinases:
            if k_name in idx.p2i:
                drv[idx.p2i[k_name]] = idx.k2i[k_name]
        for orphan, proxy in getattr(idx, "proxy_map", {}).items():
            if orphan in idx.p2i:
                drv[idx.p2i[orphan]] = idx.k2i[proxy]
        return dict(offset_y=_i32(idx.offset_y), offset_s=_i32(idx.offset_s), n_sites=_i32(idx.n_sites), W_indptr=_i32(sys.W_indptr),
                    W_indices=_i32(sys.W_indices), W_data=_f64(sys.W_data), TF_indptr=_i32(sys.TF_indptr), TF_indices=_i32(sys.TF_indices),
                    TF_data=_f64(sys.TF_data), tf_deg=_f64(sys.tf_deg), driver_map=drv, kin_grid=_f64(sys.kin_grid), kin_Kmat=_f64(sys.kin_Kmat))

    @classmethod
    def from_system(cls, sys, model: int, device: Optional[int] = None):
        """From a reference ``global_model.network.System`` (``desc_from_system`` + upload)."""
        return cls(model, device=device, **cls.desc_from_system(sys))

    @classmethod
    def from_odeint_args(cls, args: Sequence, model: int, device: Optional[int] = None):
        """From the 23-tuple ``System.odeint_args()`` of models 0 / 1 / 4 (network.py:508-526)."""
        if model == 2:
            raise ValueError("model 2's odeint_args carry S_cache instead of W / kin_Kmat: use from_system")
        (c_k, A, B, Cc, D, Dp, E, tfs, grid, kmat, wp, wi, wd, nW, tp, ti, td, nT, oy, os_, ns, deg, drv) = args
        return cls(model, oy, os_, ns, wp, wi, wd, tp, ti, td, deg, drv, grid, kmat, device)

    @classmethod
    def from_npz(cls, g, device: Optional[int] = None):
        return cls(int(g["model"]), g["offset_y"], g["offset_s"], g["n_sites"], g["W_indptr"], g["W_indices"], g["W_data"],
                   g["TF_indptr"], g["TF_indices"], g["TF_data"], g["tf_deg"], g["driver_map"], g["kin_grid"], g["kin_Kmat"], device)

    def close(self):
        if getattr(self, "_h", None):
            _capi.load().pk_network_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ candidates
    def pack_params(self, c_k, A_i, B_i, C_i, D_i, Dp_i, E_i, tf_scale) -> np.ndarray:
        """Physical parameters in ``System.update`` order (network.py:293-302) -> one candidate row."""
        x = np.concatenate([np.ravel(c_k), np.ravel(A_i), np.ravel(B_i), np.ravel(C_i), np.ravel(D_i), np.ravel(Dp_i), np.ravel(E_i),
                            [float(tf_scale)]]).astype(np.float64)
        if x.size != self.n_var:
            raise ValueError(f"expected {self.n_var} parameters, got {x.size}")
        return x

    def _prep(self, x, y, t):
        dev = torch.device("cuda", self.ctx.device)
        xd = _dev_f64(x, dev)
        if xd.dim() == 1:
            xd = xd.unsqueeze(0)
        if xd.shape[1] != self.n_var:
            raise ValueError(f"x must be [B, {self.n_var}]")
        B = xd.shape[0]
        yd = _dev_f64(y, dev)
        if yd.shape == (self.S,):
            yb = 0
        elif yd.shape == (B, self.S):
            yb = 1
        else:
            raise ValueError(f"y must be [{self.S}] or [{B}, {self.S}]")
        td = _dev_f64(np.atleast_1d(t) if not isinstance(t, torch.Tensor) else t, dev).reshape(-1)
        if td.numel() not in (1, B):
            raise ValueError("t must be a scalar or hold one value per candidate")
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        return dev, xd, yd, yb, td, int(td.numel() == B and B > 1), B

    def rhs_batch(self, x, y, t, raw: bool = False) -> torch.Tensor:
        """dy/dt [B, S] of B candidates: reference ``rhs_odeint(y, t, *args)`` (jacspeedup.py:392-394)."""
        dev, xd, yd, yb, td, tb, B = self._prep(x, y, t)
        out = torch.empty((B, self.S), dtype=torch.float64, device=dev)
        self.ctx.check(self.ctx.lib.pk_network_rhs_batch(self.ctx.handle, self._h, B, _ptr(xd), int(raw), _ptr(yd), yb, _ptr(td), tb, _ptr(out)))
        out._keepalive = (xd, yd, td)  # type: ignore[attr-defined]
        return out

    def jacobian_batch(self, x, y, t, raw: bool = False) -> torch.Tensor:
        """Analytic Jacobian [B, S, S], row-major (the reference's Dfun is the finite-difference ``fd_jacobian_odeint``)."""
        dev, xd, yd, yb, td, tb, B = self._prep(x, y, t)
        out = torch.empty((B, self.S, self.S), dtype=torch.float64, device=dev)
        self.ctx.check(self.ctx.lib.pk_network_jacobian_batch(self.ctx.handle, self._h, B, _ptr(xd), int(raw), _ptr(yd), yb, _ptr(td), tb, _ptr(out)))
        out._keepalive = (xd, yd, td)  # type: ignore[attr-defined]
        return out

    def simulate_batch(self, x, t_eval, y0=None, raw: bool = False, rtol: Optional[float] = None, atol: Optional[float] = None, max_steps: int = 1000000,
                       h0: float = 0.0, kernel: str = "auto", method: str = "auto", err_norm: str = "max"):
        """Y [B, T, S] for B candidates: reference ``simulate_odeint(sys, t_eval, rtol, atol, mxstep)`` (simulate.py:34-80) batched.
        Returns (Y, status [B], n_steps [B, 2]) as GPU tensors; flagged candidates have NaN rows (callers test np.isfinite,
        optproblem.py:125-133).  method = "auto": the order-4 additive method ARK436 where its kernel applies (topologies 0 / 1 / 4, <= 8
        sites per protein, N <= 256), else the order-3 Rosenbrock-W method; "ark" / "rosw" request one of them; "dp5": the reference's
        explicit RK45 (solvers.py:293-758) step for step.  err_norm = "max" (default: every component inside its tolerance) or "rms" (ODEPACK's weighted
        root-mean-square norm, i.e. what the reference's LSODA controls with the same rtol / atol: 1.4-1.7x fewer steps).  Measured
        (tools/gpu_norm_scan.py): on the reference-run fixtures the RMS run at 1e-8 / 1e-8 lands 0.04-0.27 band widths from LSODA at 1e-12
        (the reference's own LSODA run at those settings: 0.05-0.43), but on random full-size combinatorial populations it reaches 2 band
        widths and at 1e-5 / 1e-7 it is 6x less accurate than LSODA -- the order-3 method's error constant is larger.  Hence opt-in."""
        if method not in ("auto", "ark", "rosw", "dp5"):
            raise ValueError("method must be 'auto', 'ark', 'rosw' or 'dp5'")
        if rtol is None or atol is None:
            # parity-grade defaults (worst band error over every reference-run fixture <= 0.3, tools/gpu_norm_scan.py): the order-4 method
            # at the reference optimiser's own tolerances (config.toml:403-404), the order-3 method at 1e-7 / 1e-9.  Which of the two
            # will run is the library's decision (pk_network_resolve_method), not restated here
            ark = self.resolved_method(method, kernel) == "ark"
            rtol = (1e-8 if ark else 1e-7) if rtol is None else rtol
            atol = (1e-8 if ark else 1e-9) if atol is None else atol
        dev = torch.device("cuda", self.ctx.device)
        xd = _dev_f64(x, dev)
        if xd.dim() == 1:
            xd = xd.unsqueeze(0)
        if xd.shape[1] != self.n_var:
            raise ValueError(f"x must be [B, {self.n_var}]")
        B = xd.shape[0]
        yd = _dev_f64(self.default_y0() if y0 is None else y0, dev)
        if yd.shape == (self.S,):
            yb = 0
        elif yd.shape == (B, self.S):
            yb = 1
        else:
            raise ValueError(f"y0 must be [{self.S}] or [{B}, {self.S}]")
        th = np.ascontiguousarray(np.atleast_1d(np.asarray(t_eval, dtype=np.float64)))
        T = th.size
        Y = torch.empty((B, T, self.S), dtype=torch.float64, device=dev)
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
        nsteps = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        # kernel = "auto": register-resident one-thread-per-protein kernel when eligible; "lds": the general LDS kernel
        opts = _capi.default_opts(rtol=rtol, atol=atol, max_steps=max_steps, h0=h0, linsolve=("structured" if kernel == "lds" else "auto"),
                                  method={"dp5": "dp5", "ark": "ark436", "rosw": "ros34pw2"}.get(method), err_norm=err_norm)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.ctx.check(self.ctx.lib.pk_network_simulate_batch(self.ctx.handle, self._h, B, _ptr(xd), int(raw), _ptr(yd), yb, th.ctypes.data, T,
                                                             C.byref(opts), _ptr(Y), _ptr(status), _ptr(nsteps)))
        Y._keepalive = (xd, yd)  # type: ignore[attr-defined]
        return Y, status, nsteps

    def simulate_objective_batch(self, loss, x, t_eval, y0=None, raw: bool = False, rtol: float = 1e-8, atol: float = 1e-8, max_steps: int = 1000000,
                                 err_norm: str = "max", loss_mode: int = 0, defaults=None, lambdas=(1.0, 1.0, 1.0, 0.0), fail_value: float = 1e12,
                                 want_Y: bool = False):
        """simulate_odeint -> LOSS_FN -> objectives (optproblem.py:99-160) for B candidates in ONE launch: the integrator scores the
        observations at its output times, the trajectory stays in registers (``want_Y``: also written).  Returns
        (loss_sums [B, 3], F [B, 3], status [B], n_steps [B, 2], Y or None), or ``None`` when this network / loss data do not take the fused
        path (``pk_network_simulate_objective_batch`` answers PK_ERR_UNSUPPORTED: call ``simulate_batch`` + ``objective_batch``)."""
        dev = torch.device("cuda", self.ctx.device)
        xd = _dev_f64(x, dev)
        if xd.dim() == 1:
            xd = xd.unsqueeze(0)
        if xd.shape[1] != self.n_var:
            raise ValueError(f"x must be [B, {self.n_var}]")
        B = xd.shape[0]
        yd = _dev_f64(self.default_y0() if y0 is None else y0, dev)
        if yd.shape == (self.S,):
            yb = 0
        elif yd.shape == (B, self.S):
            yb = 1
        else:
            raise ValueError(f"y0 must be [{self.S}] or [{B}, {self.S}]")
        th = np.ascontiguousarray(np.atleast_1d(np.asarray(t_eval, dtype=np.float64)))
        T = th.size
        dd = _dev_f64(defaults, dev) if defaults is not None else None
        Y = torch.empty((B, T, self.S), dtype=torch.float64, device=dev) if want_Y else None
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
        nsteps = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        sums = torch.empty((B, 3), dtype=torch.float64, device=dev)
        F = torch.empty((B, 3), dtype=torch.float64, device=dev)
        lam = (C.c_double * 4)(*[float(v) for v in lambdas])
        opts = _capi.default_opts(rtol=rtol, atol=atol, max_steps=max_steps, err_norm=err_norm)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        rc = self.ctx.lib.pk_network_simulate_objective_batch(self.ctx.handle, self._h, loss, B, _ptr(xd), int(raw), _ptr(yd), yb, th.ctypes.data, T,
                                                              C.byref(opts), int(loss_mode), _ptr(dd), C.cast(lam, C.c_void_p), float(fail_value),
                                                              _ptr(Y), _ptr(status), _ptr(nsteps), _ptr(sums), _ptr(F))
        if rc == _capi.PK_ERR_UNSUPPORTED:
            return None
        self.ctx.check(rc)
        F._keepalive = (xd, yd, dd)  # type: ignore[attr-defined]
        return sums, F, status, nsteps, Y

    def resolved_method(self, method: str = "auto", kernel: str = "auto") -> str:
        """The integrator ``simulate_batch(method=, kernel=)`` will run on this network -- "ark", "rosw" or "dp5" -- as decided by the
        library itself (``pk_network_resolve_method``: network size, sites per protein AND the LDS footprint of the order-4 kernel).
        Raises PhoskinError when "ark" was requested and cannot run."""
        opts = _capi.default_opts(linsolve=("structured" if kernel == "lds" else "auto"), method={"dp5": "dp5", "ark": "ark436", "rosw": "ros34pw2"}.get(method))
        m = _capi.load().pk_network_resolve_method(self._h, C.byref(opts))
        if m < 0:
            raise _capi.PhoskinError("method='ark': the order-4 kernel does not fit this network (N <= 256, <= 8 sites per protein, 160 KB of LDS)")
        return {_capi.METHOD_DP5: "dp5", _capi.METHOD_ARK436: "ark"}.get(m, "rosw")

    def ark_eligible(self) -> bool:
        """Whether pk_network_simulate_batch runs the order-4 additive integrator on this network BY DEFAULT."""
        return self.resolved_method() == "ark"

    # ------------------------------------------------------------------ loss / objectives
    def make_loss(self, loss_data: dict, T: int):
        """Upload the arrays of ``cache.prepare_fast_loss_data`` (cache.py:19) for a solver grid of T points -> opaque handle."""
        keep = {}
        for k in ("p_prot", "t_prot", "p_rna", "t_rna", "p_pho", "s_pho", "t_pho"):
            keep[k] = _i32(loss_data[k])
        for k in ("obs_prot", "w_prot", "obs_rna", "w_rna", "obs_pho", "w_pho"):
            keep[k] = _f64(loss_data[k])
        d = _capi.LossData(keep["p_prot"].size, keep["p_rna"].size, keep["p_pho"].size,
                           keep["p_prot"].ctypes.data, keep["t_prot"].ctypes.data, keep["obs_prot"].ctypes.data, keep["w_prot"].ctypes.data,
                           keep["p_rna"].ctypes.data, keep["t_rna"].ctypes.data, keep["obs_rna"].ctypes.data, keep["w_rna"].ctypes.data,
                           keep["p_pho"].ctypes.data, keep["s_pho"].ctypes.data, keep["t_pho"].ctypes.data, keep["obs_pho"].ctypes.data,
                           keep["w_pho"].ctypes.data, int(loss_data["prot_base_idx"]), int(loss_data["rna_base_idx"]), int(loss_data["pho_base_idx"]))
        h = self.ctx.lib.pk_network_loss_create(self.ctx.handle, self._h, C.byref(d), int(T))
        if not h:
            raise _capi.PhoskinError("pk_network_loss_create failed: " + (self.ctx.lib.pk_last_error(self.ctx.handle) or b"").decode())
        return h

    def free_loss(self, h):
        self.ctx.lib.pk_network_loss_destroy(h)

    def objective_batch(self, loss, Y: torch.Tensor, loss_mode: int = 0, x=None, raw: bool = False, defaults=None,
                        lambdas=(1.0, 1.0, 1.0, 0.0), fail_value: float = 1e12, status: Optional[torch.Tensor] = None):
        """(loss_sums [B, 3], F [B, 3]): LOSS_FN (lossfn.py:386) and the objectives of GlobalODE_MOO._evaluate (optproblem.py:137-160)."""
        dev = torch.device("cuda", self.ctx.device)
        Y = Y.contiguous()
        B, T, S = Y.shape
        if S != self.S:
            raise ValueError("Y must be [B, T, S]")
        xd = _dev_f64(x, dev) if x is not None else None
        dd = _dev_f64(defaults, dev) if defaults is not None else None
        sums = torch.empty((B, 3), dtype=torch.float64, device=dev)
        F = torch.empty((B, 3), dtype=torch.float64, device=dev)
        lam = (C.c_double * 4)(*[float(v) for v in lambdas])
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.ctx.check(self.ctx.lib.pk_network_objective_batch(self.ctx.handle, self._h, loss, B, _ptr(Y), T, int(loss_mode), _ptr(xd), int(raw),
                                                              _ptr(dd), C.cast(lam, C.c_void_p), float(fail_value), _ptr(status), _ptr(sums), _ptr(F)))
        sums._keepalive = (Y, xd, dd, status)  # type: ignore[attr-defined]
        return sums, F

    def make_index_lists(self, times, t_prot, t_rna, t_pho):
        """Index lists for EVERY protein / site at the requested times of each modality (what simulate_and_measure tabulates,
        simulate.py:119-202) on the solver grid ``times`` -> (handle, layout dict); baselines: t = 0 (protein, phospho), t = 4 (RNA)."""
        times = np.asarray(times, float)
        bidx = lambda t0: int(np.argmin(np.abs(times - float(t0))))
        sel = lambda tp: np.where(np.isin(times, np.asarray(tp, float)))[0].astype(np.int32)
        ip, ir, iph = sel(t_prot), sel(t_rna), sel(t_pho)
        ns = self._keep[2]
        N = self.N
        p_prot = np.repeat(np.arange(N, dtype=np.int32), ip.size); t_prot_i = np.tile(ip, N)
        p_rna = np.repeat(np.arange(N, dtype=np.int32), ir.size); t_rna_i = np.tile(ir, N)
        pp, ss = [], []
        for i in range(N):
            for j in range(int(ns[i])):
                pp.append(i); ss.append(j)
        pp = np.asarray(pp, np.int32); ss = np.asarray(ss, np.int32)
        p_pho = np.repeat(pp, iph.size); s_pho = np.repeat(ss, iph.size); t_pho_i = np.tile(iph, pp.size)
        one = lambda a: np.ones(a.size)
        ld = dict(p_prot=p_prot, t_prot=t_prot_i, obs_prot=one(p_prot), w_prot=one(p_prot), p_rna=p_rna, t_rna=t_rna_i, obs_rna=one(p_rna),
                  w_rna=one(p_rna), p_pho=p_pho, s_pho=s_pho, t_pho=t_pho_i, obs_pho=one(p_pho), w_pho=one(p_pho),
                  prot_base_idx=bidx(0.0), rna_base_idx=bidx(4.0), pho_base_idx=bidx(0.0))
        return self.make_loss(ld, times.size), ld

    def observables_batch(self, lists, Y: torch.Tensor, n_obs: int, eps: float = 1e-12) -> torch.Tensor:
        """pred_fc [B, n_obs] in (protein | rna | phospho) order for index lists made by ``make_index_lists`` / ``make_loss``."""
        dev = torch.device("cuda", self.ctx.device)
        Y = Y.contiguous()
        B, T, _ = Y.shape
        out = torch.empty((B, n_obs), dtype=torch.float64, device=dev)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.ctx.check(self.ctx.lib.pk_network_observables_batch(self.ctx.handle, self._h, lists, B, _ptr(Y), T, float(eps), _ptr(out)))
        out._keepalive = (Y,)  # type: ignore[attr-defined]
        return out

    def unpack_batch(self, x_raw) -> torch.Tensor:
        """softplus of raw decision vectors (params.unpack_params, params.py:106-132) -> physical [B, n_var]."""
        dev = torch.device("cuda", self.ctx.device)
        xd = _dev_f64(x_raw, dev)
        if xd.dim() == 1:
            xd = xd.unsqueeze(0)
        out = torch.empty_like(xd)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.ctx.check(self.ctx.lib.pk_network_unpack_batch(self.ctx.handle, self._h, xd.shape[0], _ptr(xd), _ptr(out)))
        out._keepalive = (xd,)  # type: ignore[attr-defined]
        return out

    def default_y0(self) -> np.ndarray:
        """``System.y0`` default (network.py:421-441): R = 1, P (state_0) = 1, phospho states 0.01."""
        oy, ns = self._keep[0], self._keep[2]
        y = np.zeros(self.S)
        for i in range(self.N):
            st = int(oy[i])
            y[st] = 1.0; y[st + 1] = 1.0
            cnt = ((1 << int(ns[i])) - 1) if self.model == 2 else int(ns[i])
            y[st + 2: st + 2 + cnt] = 0.01
        return y
