"""Dev tool: time pk_network_simulate_batch on the synthetic config-5-shaped network; compare a few candidates with the CPU oracle."""
import sys, time
import numpy as np, torch
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd.global_model import NetworkEngine
from phoskintime_amd.global_model import synthetic

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    for model in (0, 1, 4, 2):
        net = synthetic.make_network(model=model)
        eng = NetworkEngine(**net)
        X = synthetic.random_candidates(net, B, seed=1)
        t_eval = np.unique(np.concatenate([net['kin_grid'], [15.0]]))
        Xd = torch.as_tensor(X, device='cuda')
        for rtol, atol in ((1e-5, 1e-7), (1e-7, 1e-9), (1e-8, 1e-8)):
          for method in ("rosw", "auto"):
            eng.simulate_batch(Xd[:64], t_eval, rtol=rtol, atol=atol, method=method); torch.cuda.synchronize()
            t0 = time.perf_counter()
            Y, st, ns = eng.simulate_batch(Xd, t_eval, rtol=rtol, atol=atol, method=method)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nsc = ns.cpu().numpy()
            print('model %d S %d n_var %d B %d rtol %.0e atol %.0e %s: %.1f ms  %.0f candidates/s  steps mean %.0f max %d rej mean %.0f  flagged %d' % (
                model, eng.S, eng.n_var, B, rtol, atol, method, dt * 1e3, B / dt, nsc[:, 0].mean(), nsc[:, 0].max(), nsc[:, 1].mean(), int((st != 0).sum())), flush=True)
        if model == 0 and '--oracle' in sys.argv:
            from oracle import network_models as nm
            onet = nm.Network(model=model, N=eng.N, n_K=eng.n_K, total_sites=eng.total_sites, S=eng.S, **{k: net[k] for k in (
                'offset_y', 'offset_s', 'n_sites', 'W_indptr', 'W_indices', 'W_data', 'TF_indptr', 'TF_indices', 'TF_data', 'tf_deg', 'driver_map', 'kin_grid', 'kin_Kmat')})
            x = X[0]; s = lambda a, b: x[a:b]
            nK, N, sites = eng.n_K, eng.N, eng.total_sites
            p = nm.Params(x[:nK], x[nK:nK+N], x[nK+N:nK+2*N], x[nK+2*N:nK+3*N], x[nK+3*N:nK+4*N], x[nK+4*N:nK+4*N+sites], x[nK+4*N+sites:nK+5*N+sites], float(x[-1]))
            t0 = time.perf_counter(); ref = nm.simulate_odeint(onet, p, t_eval, 1e-10, 1e-10, 500000, use_fd_jac=False); dt = time.perf_counter() - t0
            Yk = Y[0].cpu().numpy()
            print('   oracle (LSODA 1e-10, internal FD Jacobian) %.1f s; band error of the GPU result %.4f' % (dt, np.max(np.abs(Yk - ref) / (1e-8 + 1e-6 * np.abs(ref)))))
        eng.close()

if __name__ == '__main__':
    main()
