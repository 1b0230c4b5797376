"""Dev tool: run the HIP path against every golden file and print band errors (needs a GPU)."""
import glob, sys, time
import numpy as np, torch
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch
from oracle import protein_models as pm

def main():
    lins = sys.argv[1:] or ['auto', 'dense']
    for f in sorted(glob.glob('tests/golden/protein_*.npz')):
        g = np.load(f)
        name = str(g['model']); n = int(g['n_sites'])
        for method in ('rodas4', 'bdf2'):
            for lin in lins:
                t0 = time.time()
                r = batch.solve_ode_batch(name, g['theta'], g['y0'], n, g['t'], method=method, linsolve=lin, clip_nonneg=False, metric='total_signal')
                torch.cuda.synchronize()
                sol = r.sol.cpu().numpy(); st = r.status.cpu().numpy(); ns = r.n_steps.cpu().numpy()
                e_t = pm.band_error(sol, g['sol_tight'])
                e_d = pm.band_error(np.clip(sol, 0, None), g['sol_default'])
                print('%-36s %-6s %-6s band vs tight %9.3e vs default %6.3f status %s steps med %d max %d rej %d  %.2fs' % (
                    f.split('/')[-1], method, lin, e_t, e_d, np.unique(st), np.median(ns[:, 0]), ns[:, 0].max(), ns[:, 1].max(), time.time() - t0), flush=True)
        # rhs / jac
        rr = batch.rhs_batch(name, g['theta'], g['y_rand'], n).cpu().numpy()
        jj = batch.jacobian_batch(name, g['theta'], n).cpu().numpy()
        print('   rhs err %.2e jac err %.2e' % (np.abs(rr - g['rhs_y_rand']).max(), np.abs(jj - g['jac']).max()), flush=True)

if __name__ == '__main__':
    main()
