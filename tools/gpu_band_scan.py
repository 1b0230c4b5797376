#!/usr/bin/env python3
"""Dev tool: worst band error (vs the reference RHS under SciPy odeint at 1e-13, `sol_tight`) over every golden fixture for a list of
(method, rtol, atol) settings -- the table behind the choice of the default integrator / tolerance."""
import pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch
from oracle import protein_models as pm

files = sorted((pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden").glob("protein_*.npz"))
settings = [("lrp8", 1e-7, 1e-9), ("lrp12", 1e-7, 1e-9), ("lrp12", 1e-6, 1e-8), ("lrp12", 3e-6, 1e-8), ("lrp8", 1e-6, 1e-8), ("rodas4", 1e-7, 1e-9)]
for meth, rt, at in settings:
    worst = 0.0; wname = ""; per_model = {}
    for f in files:
        g = np.load(f); model = pm.MODEL_IDS[str(g["model"])]; n = int(g["n_sites"])
        r = batch.solve_ode_batch(model, g["theta"], g["y0"], n, g["t"], method=meth, rtol=rt, atol=at, clip_nonneg=False)
        assert not r.status.cpu().numpy().any(), f.name
        e = pm.band_error(r.sol.cpu().numpy(), g["sol_tight"], 1e-6, 1e-8)
        per_model[str(g["model"])] = max(per_model.get(str(g["model"]), 0.0), e)
        if e > worst: worst, wname = e, f.name
    print("%-7s rtol %.0e atol %.0e: worst band %.4f (%s)  per model %s" % (meth, rt, at, worst, wname, {k: round(v, 4) for k, v in per_model.items()}), flush=True)
