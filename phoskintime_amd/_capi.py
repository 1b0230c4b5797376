"""ctypes binding of libphoskin_hip.so (include/phoskin.h).  Thin: argument marshalling and error mapping only.

The library is the product's only compute path.  If it is missing or cannot be loaded this module raises
``PhoskinLibraryError`` -- there is deliberately NO CPU fallback (the CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here)."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libphoskin_hip.so"

DIST, SUCC, RAND = 0, 1, 2
MODEL_IDS = {"distmod": DIST, "succmod": SUCC, "randmod": RAND}
MODEL_NAMES = {v: k for k, v in MODEL_IDS.items()}
METHOD_RODAS4, METHOD_BDF2, METHOD_RK4, METHOD_LRP8, METHOD_DP5, METHOD_LRP12, METHOD_ARK436, METHOD_ROS34PW2 = 0, 1, 2, 3, 4, 5, 6, 7
METHODS = {"rodas4": METHOD_RODAS4, "bdf2": METHOD_BDF2, "rk4": METHOD_RK4, "lrp8": METHOD_LRP8, "dp5": METHOD_DP5, "lrp12": METHOD_LRP12, "ark436": METHOD_ARK436, "ros34pw2": METHOD_ROS34PW2}
LINSOLVE_AUTO, LINSOLVE_DENSE, LINSOLVE_STRUCTURED = 0, 1, 2
LINSOLVES = {"auto": LINSOLVE_AUTO, "dense": LINSOLVE_DENSE, "structured": LINSOLVE_STRUCTURED}
KERNEL_AUTO, KERNEL_GROUP, KERNEL_TPR = 0, 1, 2
KERNELS = {"auto": KERNEL_AUTO, "group": KERNEL_GROUP, "tpr": KERNEL_TPR}
NORMS = {"default": 0, "max": 1, "rms": 2}
METRICS = {"total_signal": 0, "mean_activity": 1, "variance": 2, "dynamics": 3, "l2_norm": 4}
ST_NONFINITE, ST_MAXSTEPS, ST_HMIN = 1, 2, 4


class PhoskinLibraryError(RuntimeError):
    pass


class PhoskinError(RuntimeError):
    pass


class NetworkDesc(C.Structure):
    """Mirror of ``pk_network_desc`` (include/phoskin.h)."""
    _fields_ = [("model", C.c_int32), ("N", C.c_int32), ("n_K", C.c_int32), ("total_sites", C.c_int32), ("n_grid", C.c_int32),
                ("offset_y", C.c_void_p), ("offset_s", C.c_void_p), ("n_sites", C.c_void_p),
                ("W_indptr", C.c_void_p), ("W_indices", C.c_void_p), ("W_data", C.c_void_p),
                ("TF_indptr", C.c_void_p), ("TF_indices", C.c_void_p), ("TF_data", C.c_void_p),
                ("tf_deg", C.c_void_p), ("driver_map", C.c_void_p), ("kin_grid", C.c_void_p), ("kin_Kmat", C.c_void_p)]


class LossData(C.Structure):
    """Mirror of ``pk_loss_data`` (include/phoskin.h)."""
    _fields_ = [("n_prot", C.c_int32), ("n_rna", C.c_int32), ("n_pho", C.c_int32),
                ("p_prot", C.c_void_p), ("t_prot", C.c_void_p), ("obs_prot", C.c_void_p), ("w_prot", C.c_void_p),
                ("p_rna", C.c_void_p), ("t_rna", C.c_void_p), ("obs_rna", C.c_void_p), ("w_rna", C.c_void_p),
                ("p_pho", C.c_void_p), ("s_pho", C.c_void_p), ("t_pho", C.c_void_p), ("obs_pho", C.c_void_p), ("w_pho", C.c_void_p),
                ("prot_base_idx", C.c_int32), ("rna_base_idx", C.c_int32), ("pho_base_idx", C.c_int32)]


class SolverOpts(C.Structure):
    """Mirror of ``pk_solver_opts`` (include/phoskin.h)."""
    _fields_ = [("method", C.c_int32), ("linsolve", C.c_int32), ("rtol", C.c_double), ("atol", C.c_double),
                ("h0", C.c_double), ("rk4_h", C.c_double), ("max_steps", C.c_int32), ("clip_nonneg", C.c_int32),
                ("normalize", C.c_int32), ("stage_form", C.c_int32), ("kernel", C.c_int32), ("err_norm", C.c_int32)]


#: every symbol include/phoskin.h declares (tests/test_capi_symbols.py checks the header against this list)
SYMBOLS = (
    "pk_version", "pk_create", "pk_create_error", "pk_destroy", "pk_last_error", "pk_set_stream", "pk_use_own_stream", "pk_synchronize", "pk_default_opts", "pk_workspace_stats",
    "pk_protein_n_states", "pk_protein_n_params", "pk_protein_flat_len",
    "pk_solve_protein_batch", "pk_solve_protein_sens_batch", "pk_protein_sens_available", "pk_rhs_protein_batch", "pk_jacobian_protein_batch", "pk_steady_state_protein_batch", "pk_morris_build_batch", "pk_morris_effects_batch", "pk_score_fit_batch",
    "pk_solve_protein_batch_host", "pk_solve_protein_sens_batch_host", "pk_rhs_protein_batch_host", "pk_jacobian_protein_batch_host",
    "pk_time_solve_protein_batch", "pk_measure_hbm_gbs", "pk_measure_hbm_stream_gbs", "pk_measure_fp64_fma_tflops",
    "pk_network_create", "pk_network_destroy", "pk_network_n_states", "pk_network_n_var",
    "pk_network_rhs_batch", "pk_network_jacobian_batch", "pk_network_unpack_batch", "pk_network_simulate_batch",
    "pk_network_loss_create", "pk_network_loss_destroy", "pk_network_objective_batch", "pk_network_simulate_objective_batch", "pk_network_observables_batch", "pk_frechet_batch", "pk_loss_fn_batch_host", "pk_network_resolve_method",
    "pk_comm_unique_id", "pk_comm_init", "pk_comm_rank", "pk_comm_world", "pk_allgather_f64", "pk_comm_destroy",
)

PK_OK, PK_ERR_ARG, PK_ERR_UNSUPPORTED, PK_ERR_HIP, PK_ERR_NOMEM = 0, -1, -2, -3, -4      # include/phoskin.h

_lib = None


def load():
    """Load libphoskin_hip.so (once) and declare the signatures.  Raises PhoskinLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters: PyTorch-ROCm ships its own libamdhip64; if libphoskin_hip.so were dlopen'ed first it would bind the
    # system HIP runtime and the process would end up with two runtimes (observed: pk_create fails after torch initialises).
    # Importing torch first makes both use the one runtime torch has already loaded.
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise PhoskinLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    except OSError as e:  # pragma: no cover
        raise PhoskinLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    optp = C.POINTER(SolverOpts)
    lib.pk_version.restype = i32
    lib.pk_create.restype = vp; lib.pk_create.argtypes = [i32]
    lib.pk_create_error.restype = C.c_char_p; lib.pk_create_error.argtypes = []
    lib.pk_destroy.restype = None; lib.pk_destroy.argtypes = [vp]
    lib.pk_last_error.restype = C.c_char_p; lib.pk_last_error.argtypes = [vp]
    lib.pk_set_stream.restype = i32; lib.pk_set_stream.argtypes = [vp, vp]
    lib.pk_use_own_stream.restype = i32; lib.pk_use_own_stream.argtypes = [vp]
    lib.pk_synchronize.restype = i32; lib.pk_synchronize.argtypes = [vp]
    lib.pk_default_opts.restype = None; lib.pk_default_opts.argtypes = [optp]
    lib.pk_workspace_stats.restype = i32; lib.pk_workspace_stats.argtypes = [vp, C.POINTER(C.c_int64 * 6)]
    for f in ("pk_protein_n_states", "pk_protein_n_params"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = [i32, i32]
    lib.pk_protein_flat_len.restype = i32; lib.pk_protein_flat_len.argtypes = [i32, i32, i32]
    lib.pk_protein_sens_available.restype = i32; lib.pk_protein_sens_available.argtypes = [i32, i32]
    for f in ("pk_solve_protein_sens_batch", "pk_solve_protein_sens_batch_host"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = [vp, i32, i32, i64, vp, vp, i32, vp, i32, optp, vp, vp, vp, vp]
    solve_args = [vp, i32, i32, i64, vp, vp, i32, vp, i32, optp, vp, vp, vp, i32, vp, vp]
    for f in ("pk_solve_protein_batch", "pk_solve_protein_batch_host"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = solve_args
    for f in ("pk_rhs_protein_batch", "pk_rhs_protein_batch_host"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = [vp, i32, i32, i64, vp, vp, vp]
    for f in ("pk_jacobian_protein_batch", "pk_jacobian_protein_batch_host"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = [vp, i32, i32, i64, vp, vp]
    lib.pk_steady_state_protein_batch.restype = i32; lib.pk_steady_state_protein_batch.argtypes = [vp, i32, i32, i64, vp, vp, vp]
    lib.pk_morris_build_batch.restype = i32; lib.pk_morris_build_batch.argtypes = [vp, i64, i32, dbl, vp, vp, vp, vp, vp, vp]
    lib.pk_morris_effects_batch.restype = i32; lib.pk_morris_effects_batch.argtypes = [vp, i64, i32, dbl, vp, vp, vp, vp]
    lib.pk_frechet_batch.restype = i32; lib.pk_frechet_batch.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.pk_network_create.restype = vp; lib.pk_network_create.argtypes = [vp, C.POINTER(NetworkDesc)]
    lib.pk_network_destroy.restype = None; lib.pk_network_destroy.argtypes = [vp]
    lib.pk_network_n_states.restype = i32; lib.pk_network_n_states.argtypes = [vp]
    lib.pk_network_n_var.restype = i32; lib.pk_network_n_var.argtypes = [vp]
    for f in ("pk_network_rhs_batch", "pk_network_jacobian_batch"):
        getattr(lib, f).restype = i32; getattr(lib, f).argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, i32, vp]
    lib.pk_network_simulate_batch.restype = i32
    lib.pk_network_simulate_batch.argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, i32, optp, vp, vp, vp]
    lib.pk_network_loss_create.restype = vp; lib.pk_network_loss_create.argtypes = [vp, vp, C.POINTER(LossData), i32]
    lib.pk_network_loss_destroy.restype = None; lib.pk_network_loss_destroy.argtypes = [vp]
    lib.pk_network_objective_batch.restype = i32
    lib.pk_network_objective_batch.argtypes = [vp, vp, vp, i64, vp, i32, i32, vp, i32, vp, vp, dbl, vp, vp, vp]
    lib.pk_network_simulate_objective_batch.restype = i32
    lib.pk_network_simulate_objective_batch.argtypes = [vp, vp, vp, i64, vp, i32, vp, i32, vp, i32, optp, i32, vp, vp, dbl, vp, vp, vp, vp, vp]
    lib.pk_network_observables_batch.restype = i32
    lib.pk_network_observables_batch.argtypes = [vp, vp, vp, i64, vp, i32, dbl, vp]
    lib.pk_network_resolve_method.restype = i32; lib.pk_network_resolve_method.argtypes = [vp, optp]
    lib.pk_loss_fn_batch_host.restype = i32
    lib.pk_loss_fn_batch_host.argtypes = [vp, i32, i32, i64, vp, i32, i32, C.POINTER(LossData), vp, i32, vp]
    lib.pk_network_unpack_batch.restype = i32; lib.pk_network_unpack_batch.argtypes = [vp, vp, i64, vp, vp]
    lib.pk_comm_unique_id.restype = i32; lib.pk_comm_unique_id.argtypes = [vp, C.c_char_p]
    lib.pk_comm_init.restype = i32; lib.pk_comm_init.argtypes = [vp, C.c_char_p, i32, i32]
    lib.pk_comm_rank.restype = i32; lib.pk_comm_rank.argtypes = [vp]
    lib.pk_comm_world.restype = i32; lib.pk_comm_world.argtypes = [vp]
    lib.pk_allgather_f64.restype = i32; lib.pk_allgather_f64.argtypes = [vp, vp, i64, vp]
    lib.pk_comm_destroy.restype = i32; lib.pk_comm_destroy.argtypes = [vp]
    lib.pk_score_fit_batch.restype = i32; lib.pk_score_fit_batch.argtypes = [vp, i64, vp, i32, vp, vp, i32, vp, vp]
    lib.pk_measure_hbm_gbs.restype = dbl; lib.pk_measure_hbm_gbs.argtypes = [vp, i64, i32]
    lib.pk_measure_hbm_stream_gbs.restype = dbl; lib.pk_measure_hbm_stream_gbs.argtypes = [vp, i64, i32, i32]
    lib.pk_measure_fp64_fma_tflops.restype = dbl; lib.pk_measure_fp64_fma_tflops.argtypes = [vp, i32]
    lib.pk_time_solve_protein_batch.restype = dbl
    lib.pk_time_solve_protein_batch.argtypes = [vp, i32] + solve_args[1:]
    _lib = lib
    return lib


def default_opts(**kw) -> SolverOpts:
    o = SolverOpts()
    load().pk_default_opts(C.byref(o))
    for k, v in kw.items():
        if v is None:
            continue
        if k == "method" and isinstance(v, str):
            v = METHODS[v]
        if k == "linsolve" and isinstance(v, str):
            v = LINSOLVES[v]
        if k == "kernel" and isinstance(v, str):
            v = KERNELS[v]
        if k == "err_norm" and isinstance(v, str):
            v = NORMS[v]
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


class Context:
    """One ``pk_ctx`` (one GPU).  Not thread-safe; create one per thread / per rank."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.device = int(device)
        self._h = self.lib.pk_create(self.device)
        if not self._h:
            why = self.lib.pk_create_error()
            raise PhoskinError(f"pk_create({device}) failed: {why.decode() if why else 'no usable HIP device'}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pk_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            msg = self.lib.pk_last_error(self._h)
            raise PhoskinError(f"libphoskin_hip error {rc}: {msg.decode() if msg else ''}")

    def set_stream(self, stream_ptr):
        self.check(self.lib.pk_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def synchronize(self):
        self.check(self.lib.pk_synchronize(self._h))

    def workspace_stats(self) -> dict:
        """Allocation counters / sizes of the context's persistent buffers (pk_workspace_stats)."""
        out = (C.c_int64 * 6)()
        self.check(self.lib.pk_workspace_stats(self._h, C.byref(out)))
        return dict(zip(("stage_allocs", "stage_bytes", "scratch_allocs", "scratch_bytes", "pinned_allocs", "pinned_bytes"), (int(v) for v in out)))

    @property
    def handle(self):
        return self._h
