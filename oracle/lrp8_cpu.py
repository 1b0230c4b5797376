"""ctypes wrapper of oracle/lrp8_dist.c (TEST INFRASTRUCTURE ONLY): the HIP throughput kernel's algorithm, scalar C, for the distributive
model.  ``build()`` compiles it with gcc into oracle/_build/ (also called from __graft_entry__.build())."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
SRC = _HERE / "lrp8_dist.c"
LIB = _HERE / "_build" / "liboracle_lrp8.so"
_lib = None


def build(force: bool = False) -> Path:
    LIB.parent.mkdir(exist_ok=True)
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", str(SRC), "-o", str(LIB), "-lm"], check=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.oracle_dist_rhs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.oracle_lrp8_dist_batch.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                                C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def rhs(y, theta, n):
    lib = _load()
    y = np.ascontiguousarray(y, float); th = np.ascontiguousarray(theta, float); out = np.empty_like(y)
    lib.oracle_dist_rhs(y.ctypes.data, th.ctypes.data, int(n), out.ctypes.data)
    return out


def solve_batch(theta, n, y0, t, rtol=1e-6, atol=1e-8, max_steps=100000, lo=0, hi=None, stages=12):
    """(sol [B, T, S] raw, status [B], n_steps [B, 2]) for replicas [lo, hi) of theta (others left untouched / zero).
    Defaults = the library's defaults (pk_default_opts): LRP12 at rtol 1e-6 / atol 1e-8; stages=8 with 1e-7 / 1e-9 is the LRP8 setting."""
    lib = _load()
    th = np.ascontiguousarray(theta, float); y0 = np.ascontiguousarray(y0, float); t = np.ascontiguousarray(t, float)
    B = th.shape[0]; hi = B if hi is None else hi
    sol = np.zeros((B, t.size, n + 2)); st = np.zeros(B, np.int32); ns = np.zeros((B, 2), np.int32)
    lib.oracle_lrp8_dist_batch(th.ctypes.data, lo, hi, int(n), y0.ctypes.data, t.ctypes.data, t.size, float(rtol), float(atol), int(max_steps),
                               int(stages), sol.ctypes.data, st.ctypes.data, ns.ctypes.data)
    return sol, st, ns


def _worker(args):
    import time
    theta, n, y0, t, kw = args
    t0 = time.perf_counter()
    solve_batch(theta, n, y0, t, **kw)
    return theta.shape[0], time.perf_counter() - t0


def cpu_rate(theta, n, y0, t, workers, **kw):
    """Replicas/s of the same algorithm on `workers` host processes (one chunk each); slowest worker's compute time."""
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    build()
    chunks = [c for c in np.array_split(theta, workers) if len(c)]
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("fork")) as ex:
        res = list(ex.map(_worker, [(c, n, y0, t, kw) for c in chunks]))
    return sum(r[0] for r in res) / max(r[1] for r in res)
