#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (pk_solve_protein_batch_host) on BASELINE config 3: pageable numpy buffers in,
results back to pageable numpy buffers, per call hipMalloc + H2D + kernel + D2H + hipFree.  Reported in DESIGN.md next to the
HBM-resident figure of bench.py (which is the headline `value`; this one never is)."""
import ctypes as C
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import _capi, batch  # noqa: E402

T_GRID = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])


def main():
    B, n = 65536, 30
    S, P = n + 2, 2 * n + 4
    th = np.random.default_rng(20260517).uniform(0.0, 20.0, (B, P))
    y0 = np.ones(S); t = T_GRID
    ctx = batch.get_context()
    opts = _capi.default_opts()
    F = (t.size - 5) + t.size + n * t.size
    for what in ("metric", "flat", "sol"):
        sol = np.empty((B, t.size, S)) if what == "sol" else None
        flat = np.empty((B, F)) if what == "flat" else None
        met = np.empty(B) if what == "metric" else None
        st = np.zeros(B, np.int32)
        ptr = lambda a: a.ctypes.data if a is not None else None
        times = []
        for it in range(6):
            t0 = time.perf_counter()
            rc = ctx.lib.pk_solve_protein_batch_host(ctx.handle, 0, n, B, th.ctypes.data, y0.ctypes.data, 0, t.ctypes.data, t.size, C.byref(opts),
                                                     ptr(sol), ptr(flat), ptr(met), 0, st.ctypes.data, None)
            times.append(time.perf_counter() - t0)
            assert rc == 0 and not st.any()
        best = min(times[1:])
        out_bytes = {"metric": 8 * B, "flat": 8 * B * F, "sol": 8 * B * t.size * S}[what]
        print(f"host entry point, output={what:6s}: {best * 1e3:8.2f} ms/call  {B / best:12.0f} replicas/s  "
              f"(H2D {th.nbytes / 1e6:.1f} MB, D2H {out_bytes / 1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
