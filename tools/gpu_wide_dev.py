"""Dev: band error / step counts of the wide kernels per replica at several tolerances (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, ".")
from pathlib import Path
from oracle import protein_models as pm
from phoskintime_amd import batch
files = sorted(Path("tests/golden").glob(sys.argv[1] if len(sys.argv) > 1 else "protein_randmod_n7_*.npz"))
for f in files:
    g = np.load(f); model = pm.MODEL_IDS[str(g["model"])]; n = int(g["n_sites"])
    for rtol in (1e-6, 3e-7, 1e-7):
        r = batch.solve_ode_batch(model, g["theta"], g["y0"], n, g["t"], clip_nonneg=False, rtol=rtol, atol=rtol * 1e-2)
        sol = r.sol.cpu().numpy(); ns = r.n_steps.cpu().numpy()
        bands = [pm.band_error(sol[k], g["sol_tight"][k]) for k in range(sol.shape[0])]
        print(f.name, "rtol", rtol, "bands", np.round(bands, 3), "steps", ns[:, 0], "rej", ns[:, 1], "status", r.status.cpu().numpy(), flush=True)
