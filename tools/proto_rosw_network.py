"""Numpy model of the network integrator: ROS34PW2 (Rosenbrock-W, order 3) with W = the per-protein BLOCK-DIAGONAL part of the
Jacobian only (TF coupling left to the W-method), steps landing on every output time and every kinase-bucket edge (dev tool)."""
import numpy as np, sys, glob
sys.path.insert(0, '.')
from oracle import network_models as nm

GAM = 0.435866521508459
A_ = {2: [2.0000000000000000229], 3: [1.41921731745576465, -0.25923221167296971378], 4: [4.1847604823191607312, -0.2851920173554959137, 2.2942803602790417167]}
C_ = {2: [-4.5885607205580834861], 3: [-4.1847604823191607312, 0.2851920173554959137], 4: [-6.3681792001283577635, -6.7956209444668361844, 2.8700986043310560892]}
M_ = [4.1847604823191601453, -0.28519201735549587377, 2.2942803602790413955, 1.0]
E_ = [0.27774994764796811038, -1.4032398951759990242, 1.7726301276675507452, 0.5]

def block_mask(net):
    mask = np.zeros((net.S, net.S), bool)
    for i in range(net.N):
        st = int(net.offset_y[i]); cnt = (1 + int(net.n_states[i])) if net.model == 2 else 2 + int(net.n_sites[i])
        mask[st:st+cnt, st:st+cnt] = True
    return mask

def jac_cd(net, p, y, t, h=1e-6):
    J = np.empty((net.S, net.S))
    for c in range(net.S):
        yp = y.copy(); ym = y.copy(); yp[c] += h; ym[c] -= h
        J[:, c] = (nm.rhs(net, p, yp, t) - nm.rhs(net, p, ym, t)) / (2*h)
    return J

def solve(net, p, y0, t_eval, rtol, atol, full_jac=False):
    mask = block_mask(net)
    stops = np.unique(np.concatenate([t_eval[1:], net.kin_grid[(net.kin_grid > t_eval[0]) & (net.kin_grid < t_eval[-1])]]))
    y = y0.copy(); out = np.empty((len(t_eval), net.S)); out[0] = y
    tc = t_eval[0]; I = np.eye(net.S); nst = nrej = 0
    f = lambda yy, tt: nm.rhs(net, p, yy, tt)
    h = 1e-3
    for te in stops:
        while True:
            last = tc + 1.0001*h >= te
            hs = te - tc if last else (0.5*(te-tc) if tc + 2*h > te else h)
            tb = tc                                  # bucket of the step start: the forcing is frozen over the step
            J = jac_cd(net, p, y, tb)
            if not full_jac: J = J * mask
            Winv = np.linalg.inv(I/(GAM*hs) - J)
            U = []
            U.append(Winv @ f(y, tb))
            for i in (2, 3, 4):
                Y = y + sum(a*u for a, u in zip(A_[i], U))
                U.append(Winv @ (f(Y, tb) + sum(c*u for c, u in zip(C_[i], U))/hs))
            yn = y + sum(m*u for m, u in zip(M_, U))
            e = sum(c*u for c, u in zip(E_, U))
            err = np.max(np.abs(e)/(atol + rtol*np.maximum(np.abs(y), np.abs(yn))))
            fac = max(1/6, min(5, err**(1/3)/0.9)); hnew = hs/fac; nst += 1
            if err <= 1:
                y = yn; tc += hs
                if last:
                    tc = te; h = max(hnew, h) if hs < h else hnew; break
                h = hnew
            else:
                nrej += 1; h = hnew
        idx = np.where(t_eval == te)[0]
        if idx.size: out[idx[0]] = y
    return out, nst, nrej

def band(y, ref, rtol=1e-6, atol=1e-8): return np.max(np.abs(y-ref)/(atol+rtol*np.abs(ref)))

if __name__ == '__main__':
    for fn in sorted(glob.glob('tests/golden/network_m*_small.npz')) + ['tests/golden/network_m0_medium.npz']:
        g = np.load(fn); net = nm.Network.from_npz(g)
        for k in (0, 1):
            p = nm.Params.from_npz(g, k)
            for rtol, atol in ((1e-6, 1e-8), (1e-7, 1e-9)):
                for full in (False, True):
                    if full and rtol != 1e-7: continue
                    Y, nst, nrej = solve(net, p, g['y0'], g['t_eval'], rtol, atol, full)
                    print('%-24s set %d rtol %.0e %s: band vs tight %.4f  (ref lsoda8 vs tight %.3f) steps %d rej %d' % (
                        fn.split('/')[-1], k, rtol, 'fullJ ' if full else 'blockJ', band(Y, g['Y_tight'][k]), band(g['Y_lsoda8'][k], g['Y_tight'][k]), nst, nrej), flush=True)
