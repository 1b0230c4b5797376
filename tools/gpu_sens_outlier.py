import sys, zlib, numpy as np
import pathlib; R = pathlib.Path(__file__).resolve().parents[1]; sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tools'))
from oracle import protein_models as pm
import gpu_sens_audit as A
from phoskintime_amd import batch
model, n, dname = "succmod", 1, "logU(1e-2,20)"
mid = pm.MODEL_IDS[model]; S, P = pm.n_states(mid, n), pm.n_params(mid, n)
rng = np.random.default_rng(zlib.crc32(('%s %d %s' % (model, n, dname)).encode()))
th = 10.0 ** rng.uniform(-2.0, np.log10(20.0), (16, P)); y0 = rng.uniform(0.3, 2.0, S)
for rt, at in ((1e-9, 1e-11), (1e-11, 1e-13)):
    r = batch.solve_ode_sens_batch(model, th, y0, n, A.T, rtol=rt, atol=at)
    d = r.dflat.cpu().numpy(); fl = r.flat.cpu().numpy()
    worst = (0, None)
    for b in range(16):
        ref = A._jac((mid, th[b], y0, n))
        e = np.abs(d[b] - ref) / (1.0 + np.abs(ref))
        if e.max() > worst[0]:
            f, c = np.unravel_index(np.argmax(e), e.shape); worst = (e.max(), (b, f, c, d[b, f, c], ref[f, c], fl[b, f], th[b]))
    print("rtol %g: worst %.3e replica %d flat-entry %d param %d: kernel %.10e oracle %.10e  flat value %.6e\n theta %s steps %s" % (rt, worst[0], *worst[1][:6], np.round(worst[1][6], 4), r.n_steps.cpu().numpy()[worst[1][0]]))
